"""Readers/writers for the .mat files either side of the hot path.

The reference hands data between stages through MATLAB files: ``learned_qso_model_*.mat``
(learn_qso_model.m:113-123), ``dla_samples.mat`` (generate_dla_samples.m:59-63),
``preloaded_qsos.mat`` (preload_qsos.m:64-79), ``catalog.mat`` (build_catalogs.m:86-91) and the
output ``processed_qsos_*.mat`` (process_qsos.m:236-250).  They are saved with ``-v7.3`` (HDF5).

* v7.3 files need ``h5py`` (not installed in the build image; imported lazily).  HDF5 stores MATLAB
  arrays transposed, and cell arrays as object references; both are undone here so callers see the
  MATLAB shapes.
* v5/v7 files (``save -v7``) go through ``scipy.io``; this is also what the tests use.
* Output is written as a v5 ``.mat`` with exactly the variable names the reference saves, arrays
  in MATLAB orientation (``sample_log_likelihoods_dla`` is [num_quasars x S]), so
  ``CDDF_analysis/qso_loader.py``-style consumers only need their usual ``.T`` when they read it
  through h5py-free paths.
"""
from __future__ import annotations

import numpy as np


def _is_hdf5(path: str) -> bool:
    with open(path, "rb") as f:
        head = f.read(128)
    return b"MATLAB 7.3" in head or head[:8] == b"\x89HDF\r\n\x1a\n"


def _load_mat(path: str, names):
    """Returns {name: array} in MATLAB orientation for the requested variables."""
    if _is_hdf5(path):
        try:
            import h5py
        except ImportError as e:  # pragma: no cover - h5py is absent from the build image
            raise ImportError(f"{path} is a -v7.3 (HDF5) .mat file; reading it needs h5py") from e
        out = {}
        with h5py.File(path, "r") as f:
            for n in names:
                if n not in f:
                    continue
                d = f[n]
                if d.dtype == object:  # cell array: one reference per element
                    out[n] = [np.array(f[r]).T.squeeze() for r in np.array(d).ravel()]
                else:
                    out[n] = np.array(d).T
        return out
    from scipy.io import loadmat
    raw = loadmat(path, variable_names=list(names), squeeze_me=False)
    out = {}
    for n in names:
        if n not in raw:
            continue
        v = raw[n]
        if v.dtype == object:
            out[n] = [np.asarray(c).squeeze() for c in v.ravel()]
        else:
            out[n] = v
    return out


def _vec(a):
    return np.asarray(a, dtype=np.float64).reshape(-1)


def load_learned_model(path: str) -> dict:
    """Variables of process_qsos.m:30-35."""
    names = ("rest_wavelengths", "mu", "M", "log_omega", "log_c_0", "log_tau_0", "log_beta")
    m = _load_mat(path, names)
    missing = [n for n in names if n not in m]
    if missing:
        raise KeyError(f"{path} lacks {missing}")
    G = _vec(m["rest_wavelengths"]).size
    M = np.asarray(m["M"], dtype=np.float64)
    if M.shape[0] != G:
        M = M.T
    return dict(rest_wavelengths=_vec(m["rest_wavelengths"]), mu=_vec(m["mu"]), M=np.asfortranarray(M),
                log_omega=_vec(m["log_omega"]), log_c_0=float(_vec(m["log_c_0"])[0]),
                log_tau_0=float(_vec(m["log_tau_0"])[0]), log_beta=float(_vec(m["log_beta"])[0]))


def load_dla_samples(path: str) -> dict:
    """Variables of process_qsos.m:38-40 (+ lls_nhi_samples when present, set_lls_parameters.m:63)."""
    m = _load_mat(path, ("offset_samples", "log_nhi_samples", "nhi_samples", "lls_nhi_samples",
                         "lls_log_nhi_samples"))
    out = {k: _vec(v) for k, v in m.items()}
    for need in ("offset_samples", "log_nhi_samples", "nhi_samples"):
        if need not in out:
            raise KeyError(f"{path} lacks {need}")
    return out


def load_preloaded_qsos(path: str, z_qsos, test_ind=None) -> list:
    """The ragged cell arrays of preload_qsos.m:64-79 as a list of per-quasar dicts, after the
    ``test_ind`` subset of process_qsos.m:56-61 (boolean mask or index array, 0-based)."""
    m = _load_mat(path, ("all_wavelengths", "all_flux", "all_noise_variance", "all_pixel_mask"))
    n = len(m["all_wavelengths"])
    z = _vec(z_qsos)
    if z.size != n:
        raise ValueError(f"{n} spectra but {z.size} redshifts")
    idx = np.arange(n) if test_ind is None else np.flatnonzero(test_ind) if np.asarray(test_ind).dtype == bool \
        else np.asarray(test_ind)
    return [dict(wavelengths=_vec(m["all_wavelengths"][i]), flux=_vec(m["all_flux"][i]),
                 noise_variance=_vec(m["all_noise_variance"][i]),
                 pixel_mask=np.asarray(m["all_pixel_mask"][i]).reshape(-1).astype(np.uint8),
                 z_qso=float(z[i])) for i in idx]


#: process_qsos.m:236-244
SAVED_VARIABLES = ("min_z_dlas", "max_z_dlas", "log_priors_no_dla", "log_priors_dla",
                   "log_likelihoods_no_dla", "sample_log_likelihoods_dla", "log_likelihoods_dla",
                   "log_posteriors_no_dla", "log_posteriors_dla", "model_posteriors", "p_no_dlas",
                   "p_dlas")


def save_processed_qsos(path: str, results: dict, **run_metadata) -> None:
    """process_qsos.m:236-250: the result variables (column vectors / [nq x S] matrices as MATLAB
    holds them) plus the run metadata the script echoes (training_release, test_set_name, ...)."""
    from scipy.io import savemat
    out = dict(run_metadata)
    for k in ("prior_z_qso_increase", "max_z_cut", "num_lines"):
        if k in results:
            out[k] = results[k]
    for k in SAVED_VARIABLES:
        v = np.asarray(results[k], dtype=np.float64)
        out[k] = v.reshape(-1, 1) if v.ndim == 1 else v
    savemat(path, out, do_compression=True)
