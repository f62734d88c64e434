"""Worker of test_gpu_record_classes.py: one seeded case run in a clean child process (forked from
conftest.py's fork server, which predates any GPU use) with the library's diagnostic environment
switches -- and GPDLA_LIB_PATH = libgpdla_legacy.so, the only library that knows them -- set BEFORE the
library is loaded (both are read once per process)."""
import os

import numpy as np


def build_case(kind, k, num_lines):
    import gp_dla_detection_amd as gp
    from gp_dla_detection_amd import synthetic
    from gp_dla_detection_amd.parameters import MultiParameters, Parameters
    model = synthetic.make_model(k)
    sizes = [333, 801, 64, 1250, 507, 9, 222, 640, 415]
    spectra = [synthetic.make_spectrum(4100 + 7 * i + k, n, model, mask_fraction=0.05 if i % 2 else 0.0)
               for i, n in enumerate(sizes)]
    cat = synthetic.make_prior_catalog()
    z = np.array([s["z_qso"] for s in spectra])
    if kind == "single":
        samples = synthetic.make_samples(300)
        return model, samples, spectra, gp.dla_existence_prior(cat["z_qsos"], cat["dla_ind"], z), \
            Parameters(num_lines=num_lines)
    p = MultiParameters(max_dlas=3, num_lines=num_lines)
    samples = synthetic.make_samples(200)
    lp = gp.dla_existence_prior_multi(cat["z_qsos"], cat["dla_ind"], z, 0.31, 0.69, p)
    return model, samples, spectra, lp, p


def run_case(kind, k, num_lines):
    import gp_dla_detection_amd as gp
    model, samples, spectra, lp, p = build_case(kind, k, num_lines)
    if kind == "single":
        return gp.process_qsos(model, samples, spectra, log_priors=lp, params=p)
    return gp.process_qsos_multiple_dlas_meanflux(model, samples, spectra, lp, params=p)


def run_child(kind, k, num_lines, env, out_path):
    os.environ.update(env)
    out = run_case(kind, k, num_lines)
    np.savez(out_path, **{name: np.asarray(v) for name, v in out.items() if isinstance(v, np.ndarray)})
