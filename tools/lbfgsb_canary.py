#!/usr/bin/env python3
"""CPU-only desk check for the round-2 host fault: does SciPy 1.15.3's C port of
L-BFGS-B write outside the workspaces `_lbfgsb_py._minimize_lbfgsb` hands it?

`_lbfgsb.setulb` is driven exactly as `_lbfgsb_py.py:418-470` drives it (same array sizes and dtypes,
n = 483, m = 10: the problem of tests/test_training.py::test_gpu_fit_decreases_objective, with the
CPU oracle's objective in place of the GPU's), but every array is a view into the middle of a larger
buffer filled with a canary pattern; after every call the canaries are compared.  Not for the GPU
box: nothing here touches a GPU."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.optimize import _lbfgsb  # noqa: E402

from oracle import oracle  # noqa: E402
from tests.test_training import training_problem  # noqa: E402

PAD = 4096  # bytes of canary on either side


class Guarded:
    def __init__(self, count, dtype, fill=0):
        dt = np.dtype(dtype)
        self.raw = np.full(2 * PAD + count * dt.itemsize, 0xA5, dtype=np.uint8)
        self.view = self.raw[PAD:PAD + count * dt.itemsize].view(dt)
        self.view[...] = fill
        self.nbytes = count * dt.itemsize

    def intact(self):
        return bool((self.raw[:PAD] == 0xA5).all() and (self.raw[PAD + self.nbytes:] == 0xA5).all())


def main():
    x0, F, L1, NV = training_problem(nq=80, G=96, k=4, seed=9)
    rng = np.random.default_rng(2)
    x0 = x0.copy()
    x0[: 96 * 4] += 0.05 * rng.standard_normal(96 * 4)
    n, m = x0.size, 10
    assert n == 483
    arrays = dict(
        x=Guarded(n, np.float64), low=Guarded(n, np.float64), up=Guarded(n, np.float64),
        nbd=Guarded(n, np.int32), g=Guarded(n, np.float64),
        wa=Guarded(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64), iwa=Guarded(3 * n, np.int32),
        task=Guarded(2, np.int32), ln_task=Guarded(2, np.int32), lsave=Guarded(4, np.int32),
        isave=Guarded(44, np.int32), dsave=Guarded(29, np.float64))
    a = {k: v.view for k, v in arrays.items()}
    a["x"][:] = x0
    f = 0.0
    calls = evals = iters = 0
    factr, pgtol, maxls = 1e7, 1e-5, 20
    while True:
        _lbfgsb.setulb(m, a["x"], a["low"], a["up"], a["nbd"], f, a["g"], factr, pgtol, a["wa"],
                       a["iwa"], a["task"], a["lsave"], a["isave"], a["dsave"], maxls, a["ln_task"])
        calls += 1
        bad = [k for k, v in arrays.items() if not v.intact()]
        if bad:
            print(f"CANARY OVERWRITTEN around {bad} after call {calls}")
            return 1
        if a["task"][0] == 3:
            f, g = oracle.objective(a["x"].copy(), F, L1, NV)
            a["g"][:] = g
            evals += 1
        elif a["task"][0] == 1:
            iters += 1
            if iters >= 200 or evals > 400:
                break
        else:
            break
    print(f"n = {n}, m = {m}: {calls} setulb calls, {evals} evaluations, {iters} iterations, f = {f:.6f}, "
          f"task = {a['task'].tolist()}; all canaries intact ({PAD} B either side of 12 arrays)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
