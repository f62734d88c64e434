"""Diagnostic: repeat the small L-BFGS fit of tests/test_training.py many times in one process and
check that every repeat follows the same trajectory bit for bit (the matrix-core objective sums in a
fixed order, so any difference is a race).  Run on the GPU box:
    python tools/train_repeat_probe.py [repeats]"""
import hashlib
import sys

import numpy as np

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
from test_training import training_problem  # noqa: E402

from gp_dla_detection_amd import training  # noqa: E402


def main():
    repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    shapes = ((80, 96, 4, 9), (37, 203, 20, 3), (130, 70, 7, 5))
    for (nq, G, k, seed) in shapes:
        x, F, L1, NV = training_problem(nq=nq, G=G, k=k, seed=seed)
        rng = np.random.default_rng(2)
        x0 = x.copy()
        x0[: G * k] += 0.05 * rng.standard_normal(G * k)
        first = None
        bad = 0
        for rep in range(repeats):
            t = training.TrainingSet(F, L1, NV)
            trace = hashlib.sha256()
            evals = [0]

            def fun(xx):
                f, g = t.objective(xx)
                evals[0] += 1
                trace.update(np.float64(f).tobytes())
                trace.update(g.tobytes())
                if not (np.isfinite(f) and np.isfinite(g).all()):
                    print(f"  non-finite output at repeat {rep} evaluation {evals[0]}", flush=True)
                return f, g

            from scipy.optimize import minimize
            res = minimize(fun, x0, jac=True, method="L-BFGS-B", options=dict(maxiter=30, maxfun=60))
            t.close()
            digest = trace.hexdigest()
            if first is None:
                first = digest
                print(f"nq={nq} G={G} k={k}: f={res.fun:.12g} after {evals[0]} evaluations", flush=True)
            elif digest != first:
                bad += 1
                print(f"  repeat {rep}: trajectory differs (f={res.fun:.17g}, {evals[0]} evaluations)", flush=True)
        print(f"nq={nq} G={G} k={k}: {repeats} repeats, {bad} differing trajectories", flush=True)


if __name__ == "__main__":
    main()
