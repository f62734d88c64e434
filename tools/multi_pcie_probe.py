#!/usr/bin/env python3
"""Diagnostic: where the host-to-host multi-DLA call spends its time (cProfile of one pipelined call)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gp_dla_detection_amd as gp  # noqa: E402
from gp_dla_detection_amd import synthetic  # noqa: E402
from gp_dla_detection_amd.parameters import MultiParameters  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
p = MultiParameters(max_dlas=4)
model = synthetic.make_model(20)
samples = synthetic.make_samples(10000)
base = synthetic.make_spectra(8, 1500, model, first_index=500)
spectra = [base[i % 8] for i in range(64 * nb)]
cat = synthetic.make_prior_catalog()
z = np.array([s["z_qso"] for s in spectra])
lp = gp.dla_existence_prior_multi(cat["z_qsos"], cat["dla_ind"], z, 0.31, 0.69, p)
gp.process_qsos_multiple_dlas_meanflux(model, samples, spectra[:64], tuple(np.asarray(v)[:64] for v in lp), params=p)
for rep in range(2):
    t0 = time.perf_counter()
    gp.process_qsos_multiple_dlas_meanflux(model, samples, spectra, lp, params=p, max_quasars_per_batch=64)
    print(f"call {rep}: {time.perf_counter() - t0:.3f} s for {nb} batches of 64")
pr = cProfile.Profile()
pr.enable()
gp.process_qsos_multiple_dlas_meanflux(model, samples, spectra, lp, params=p, max_quasars_per_batch=64)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
