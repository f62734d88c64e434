"""Randomised parity check of the sweep against the CPU oracle (run on the GPU box): random rank
1..40, length, sample count, mask fraction and -- every other trial -- Lyman-series line count 1..31
(voigt.c:16; the run-time-line kernels), three otherwise (set_parameters.m:63); tolerance 1e-8
absolute as in tests/.
    python tools/fuzz_parity.py [trials [kmin [kmax [seed [nmax]]]]]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import gp_dla_detection_amd as gp
from gp_dla_detection_amd import synthetic
from oracle import oracle
from oracle.oracle import OracleParams
args = [int(a) for a in sys.argv[1:]]
trials, kmin, kmax, seed, nmax = (args + [24, 1, 40, 7, 700][len(args):])[:5]
rng = np.random.default_rng(seed)
worst = 0.0
for trial in range(trials):
    k = int(rng.integers(kmin, kmax + 1))
    n = int(rng.integers(30, nmax))
    S = int(rng.integers(1, 90))
    nl = int(rng.integers(1, 32)) if trial % 2 else 3
    model = synthetic.make_model(k)
    samples = synthetic.make_samples(S)
    sp = synthetic.make_spectrum(2000 + trial, n, model, mask_fraction=float(rng.uniform(0, 0.2)))
    out = gp.process_qsos(model, samples, [sp], log_priors=(np.array([-1.0]), np.array([-1.0])),
                          params=gp.Parameters(num_lines=nl))
    ref = oracle.process_spectrum(model, samples["offset_samples"], samples["nhi_samples"], sp["wavelengths"],
                                  sp["flux"], sp["noise_variance"], sp["pixel_mask"], sp["z_qso"],
                                  params=OracleParams(num_lines=nl))
    d = max(float(np.nanmax(np.abs(out["sample_log_likelihoods_dla"][0] - ref["sample_log_likelihoods_dla"]))),
            abs(out["log_likelihoods_no_dla"][0] - ref["log_likelihood_no_dla"]))
    worst = max(worst, d)
    print(trial, k, n, S, nl, f"{d:.2e}", flush=True)
print("worst", worst)
assert worst < 1e-8
