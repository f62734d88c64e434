"""CPU-side checks of the drop-in boundary: libgpdla.so loads, exports every symbol that
include/gpdla.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gp_dla_detection_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpdla.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpdla_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_typed(lib):
    names = declared_symbols()
    assert len(names) >= 20
    typed = {n for n, _, _ in _lib.SYMBOLS}
    for n in names:
        assert hasattr(lib, n), f"libgpdla.so does not export {n}"
        assert n in typed, f"{n} has no ctypes signature in _lib.SYMBOLS"
    assert lib.gpdla_abi_version() == 1


def test_struct_layouts_match_header(lib):
    # gpdla_default_config writes through the C layout; the ctypes mirror must read it back
    cfg = _lib.Config()
    lib.gpdla_default_config(C.byref(cfg))
    assert (cfg.min_lambda, cfg.max_lambda, cfg.width, cfg.num_lines) == (911.75, 1215.75, 3, 3)
    assert cfg.max_dlas == 4 and cfg.num_forest_lines == 31
    assert abs(cfg.min_z_separation - 3000 * 1000 / 299792458) < 1e-18
    assert (cfg.prev_tau_0, cfg.prev_beta) == (0.0023, 3.65)


def test_argument_validation_needs_no_gpu(lib):
    lam = np.linspace(4000.0, 4001.0, 6)
    out = np.zeros(8)
    dp = C.POINTER(C.c_double)
    rc = lib.gpdla_voigt(lam.ctypes.data_as(dp), 6, 2.0, 1e20, 3, out.ctypes.data_as(dp), 0)
    assert rc == -1 and b"n_padded" in lib.gpdla_last_error()
    rc = lib.gpdla_voigt(lam.ctypes.data_as(dp), 6, 2.0, 1e20, 3, None, 0)
    assert rc == -1
    lam = np.linspace(4000.0, 4100.0, 64)
    rc = lib.gpdla_voigt(lam.ctypes.data_as(dp), 64, 2.0, 1e20, 32, out.ctypes.data_as(dp), 0)
    assert rc == -1 and b"num_lines" in lib.gpdla_last_error()


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tests")
    lam = np.linspace(4000.0, 4100.0, 64)
    out = np.zeros(58)
    dp = C.POINTER(C.c_double)
    rc = lib.gpdla_voigt(lam.ctypes.data_as(dp), 64, 2.0, 1e20, 3, out.ctypes.data_as(dp), 0)
    assert rc == -2, "without a GPU the library must fail loudly, not compute on the CPU"
    h = C.c_void_p()
    assert lib.gpdla_context_create(0, C.byref(h)) == -2


def test_prior_matches_counting_definition():
    from gp_dla_detection_amd import dla_existence_prior, Parameters
    rng = np.random.default_rng(5)
    pz = rng.uniform(2.0, 5.0, 400)
    pd = rng.uniform(size=400) < 0.15
    z = rng.uniform(2.2, 4.8, 25)
    p = Parameters()
    log_no, log_dla = dla_existence_prior(pz, pd, z, p)
    for i, zq in enumerate(z):  # process_qsos.m:122-131 verbatim
        less = pz < (zq + p.prior_z_qso_increase)
        n, m = np.count_nonzero(less), np.count_nonzero(pd[less])
        assert log_dla[i] == np.log(m) - np.log(n)
        assert log_no[i] == np.log(n - m) - np.log(n)
