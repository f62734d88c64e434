"""Randomised parity check of the multi-DLA driver against the CPU oracle (run on the GPU box):
random max_dlas, rank (half of the cases in the 20 < k <= 40 class), length, sample count and mask
fraction; the base-sample indices are drawn on the GPU and replayed by the oracle, as in
tests/test_gpu_multi.py, whose comparison (1e-8 absolute, NaN patterns equal, MAP columns to 1e-12)
this reuses.
    python tools/fuzz_multi.py [trials [seed]]"""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np

import gp_dla_detection_amd as gp
from gp_dla_detection_amd import synthetic
from gp_dla_detection_amd.parameters import MultiParameters
from oracle import oracle
import test_gpu_multi as T

args = [int(a) for a in sys.argv[1:]]
trials, seed = (args + [48, 5][len(args):])[:2]
rng = np.random.default_rng(seed)
for trial in range(trials):
    md = int(rng.integers(1, 5))
    k = int(rng.integers(21, 41)) if trial % 2 else int(rng.integers(1, 21))
    n = int(rng.integers(60, 700))
    S = int(rng.integers(16, 150))
    p = MultiParameters(max_dlas=md, rng_seed=500 + trial)
    model = synthetic.make_model(k)
    samples = synthetic.make_samples(S)
    sp = synthetic.make_spectrum(5000 + trial, n, model, mask_fraction=float(rng.uniform(0, 0.15)))
    out = gp.process_qsos_multiple_dlas_meanflux(model, samples, [sp], T.priors([sp], p), params=p)
    bsi = out["base_sample_inds"][0] if md > 1 else np.zeros((0, S), np.uint32)
    ref = T.oracle_multi(oracle, model, samples, sp, bsi, p)
    T.compare(out, 0, ref, p)
    got = out["sample_log_likelihoods_dla"][0].T
    d = float(np.nanmax(np.abs(got - ref["sample_log_likelihoods_dla"]))) if np.isfinite(got).any() else 0.0
    print(trial, "max_dlas", md, "k", k, "n", n, "S", S, f"{d:.2e}", flush=True)
print("all", trials, "cases within tolerance")
