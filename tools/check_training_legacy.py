#!/usr/bin/env python3
"""Diagnostic: the training objective of the library in use against the CPU oracle on two ragged shapes.
For the round-1 one-block-per-slot cross-check kernel:
    GPDLA_LIB_PATH=gp_dla_detection_amd/csrc/libgpdla_legacy.so GPDLA_TRAIN_LEGACY=1 python tools/check_training_legacy.py
(the product library does not contain it and reads no environment variable)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_dla_detection_amd import training  # noqa: E402
from oracle import oracle  # noqa: E402

rng = np.random.default_rng(3)
for nq, G, k in ((37, 203, 20), (21, 90, 23)):
    M = rng.standard_normal((G, k)) * 0.3 * 0.8 ** np.arange(k)
    x = np.concatenate([M.ravel(order="F"), rng.uniform(-3, -2, G), [np.log(0.1), np.log(0.0023), np.log(3.65)]])
    L1 = 1 + rng.uniform(1.5, 3.0, (nq, G))
    NV = 10 ** rng.uniform(-3, -1, (nq, G))
    F = rng.standard_normal((nq, k)) @ M.T + np.sqrt(NV) * rng.standard_normal((nq, G))
    F[rng.uniform(size=F.shape) < 0.1] = np.nan
    f, g = training.objective(x, F, L1, NV)
    f_ref, g_ref = oracle.objective(x, F, L1, NV)
    print(f"nq={nq} G={G} k={k}: |f - f_ref| / |f_ref| = {abs(f - f_ref) / abs(f_ref):.2e}, "
          f"max |g - g_ref| / max |g_ref| = {np.abs(g - g_ref).max() / np.abs(g_ref).max():.2e}")
    assert abs(f - f_ref) < 1e-9 * abs(f_ref) and np.abs(g - g_ref).max() < 1e-9 * np.abs(g_ref).max()
print("ok (legacy kernel)" if os.environ.get("GPDLA_TRAIN_LEGACY") else "ok (matrix-core path)")
