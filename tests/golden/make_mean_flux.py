#!/opt/conda/bin/python3.9
"""BUILD CONTAINER ONLY: golden vectors for the multi-DLA driver's mean-flux suppression
(multi_dlas/process_qsos_multiple_dlas_meanflux.m:267-285) from the reference's OWN Python
restatement of it, ``QSOLoader.total_scale_factor`` (CDDF_analysis/qso_loader.py:1777-1822, a static
method; importing the module needs h5py, hence the conda interpreter).

    /opt/conda/bin/python3.9 tests/golden/make_mean_flux.py   ->  tests/golden/mean_flux.npz

Inputs and the numbers the reference computed are stored; tests/test_oracle_driver.py holds the
oracle's restatement of multi :267-285 to them, and the GPU's k_prepare is held to the oracle."""
import os
import sys

import numpy as np

np.bool, np.int, np.float = bool, int, float  # aliases the reference still uses (removed in NumPy 1.24)
sys.path.insert(0, "/root/reference")
from CDDF_analysis.qso_loader import QSOLoader  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
cases = {}
rng = np.random.default_rng(11)
for i, (z_qso, tau, beta, lines) in enumerate([(2.2, 0.0023, 3.65, 31), (3.1, 0.0023, 3.65, 31), (4.6, 0.0023, 3.65, 31),
                                               (2.9, 0.0031, 3.2, 5), (3.7, 0.0023, 3.65, 1)]):
    # rest wavelengths over (and a little beyond) the modelled range, irregular
    rest = np.sort(rng.uniform(905.0, 1225.0, 400))
    out = QSOLoader.total_scale_factor(tau, beta, z_qso, rest, num_lines=lines)
    cases[f"z_qso_{i}"], cases[f"tau_{i}"], cases[f"beta_{i}"], cases[f"lines_{i}"] = z_qso, tau, beta, lines
    cases[f"rest_{i}"], cases[f"scale_{i}"] = rest, out
cases["num_cases"] = 5
np.savez_compressed(os.path.join(HERE, "mean_flux.npz"), **cases)
print("wrote mean_flux.npz:", [float(cases[f"scale_{i}"].min()) for i in range(5)])
