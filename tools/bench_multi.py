#!/usr/bin/env python3
"""Times BASELINE config 4 (multi-DLA driver, S = 10000 samples, up to max_dlas stacked absorbers)
through the one-shot entry point gpdla_process_batch_multi on one GPU.  Prints a JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gp_dla_detection_amd as gp  # noqa: E402
from gp_dla_detection_amd import synthetic  # noqa: E402
from gp_dla_detection_amd.parameters import MultiParameters  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--spectra", type=int, default=32)
ap.add_argument("--pixels", type=int, default=1500)
ap.add_argument("--samples", type=int, default=10000)
ap.add_argument("--max-dlas", type=int, default=4)
args = ap.parse_args()
p = MultiParameters(max_dlas=args.max_dlas)
model = synthetic.make_model(20)
samples = synthetic.make_samples(args.samples)
base = synthetic.make_spectra(min(8, args.spectra), args.pixels, model, first_index=500)
spectra = [base[i % len(base)] for i in range(args.spectra)]
cat = synthetic.make_prior_catalog()
z = np.array([s["z_qso"] for s in spectra])
lp = gp.dla_existence_prior_multi(cat["z_qsos"], cat["dla_ind"], z, 0.31, 0.69, p)
gp.process_qsos_multiple_dlas_meanflux(model, samples, spectra[:2], tuple(x[:2] for x in lp), params=p)
t0 = time.perf_counter()
out = gp.process_qsos_multiple_dlas_meanflux(model, samples, spectra, lp, params=p)
dt = time.perf_counter() - t0
evals = args.spectra * args.samples * (1 + args.max_dlas)  # LLS + max_dlas DLA models
print(json.dumps({"metric": "multi-DLA sample log-likelihoods/sec (host buffers in/out)",
                  "value": evals / dt, "seconds": dt, "spectra": args.spectra, "pixels": args.pixels,
                  "samples": args.samples, "max_dlas": args.max_dlas,
                  "evaluations": evals, "finite_fraction": float(np.isfinite(out["sample_log_likelihoods_dla"]).mean())}))
