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
    assert lib.gpdla_abi_version() == 6


def test_struct_layouts_match_header(lib):
    # gpdla_default_config writes through the C layout; the ctypes mirror must read it back
    cfg = _lib.Config()
    lib.gpdla_default_config(C.byref(cfg))
    assert (cfg.min_lambda, cfg.max_lambda, cfg.width, cfg.num_lines) == (911.75, 1215.75, 3, 3)
    assert cfg.max_dlas == 4 and cfg.num_forest_lines == 31
    assert abs(cfg.min_z_separation - 3000 * 1000 / 299792458) < 1e-18
    assert (cfg.prev_tau_0, cfg.prev_beta) == (0.0023, 3.65)
    assert cfg.rng_seed == 0x9E3779B97F4A7C15 and cfg.first_quasar_index == 0
    assert cfg.contraction_precision == 0 and cfg.multi_profile_bytes == 0
    assert cfg.pipeline_slots == 0 and cfg.max_quasars_per_batch == 0


C_CONSUMER = r"""
/* A C99 consumer of include/gpdla.h: prints the layout the COMPILER gives every struct of the
 * boundary, calls through the declared prototypes, and checks the default config. */
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include "gpdla.h"
#define OFF(T, f) printf(#T "." #f " %zu\n", offsetof(T, f))
int main(void) {
  printf("sizeof gpdla_model %zu\n", sizeof(gpdla_model));
  printf("sizeof gpdla_samples %zu\n", sizeof(gpdla_samples));
  printf("sizeof gpdla_spectra %zu\n", sizeof(gpdla_spectra));
  printf("sizeof gpdla_spectra_cells %zu\n", sizeof(gpdla_spectra_cells));
  printf("sizeof gpdla_config %zu\n", sizeof(gpdla_config));
  printf("sizeof gpdla_results %zu\n", sizeof(gpdla_results));
  printf("sizeof gpdla_results_multi %zu\n", sizeof(gpdla_results_multi));
  OFF(gpdla_model, rest_wavelengths); OFF(gpdla_model, log_c_0); OFF(gpdla_model, log_beta);
  OFF(gpdla_samples, lls_nhi_samples);
  OFF(gpdla_spectra, pixel_mask); OFF(gpdla_spectra, log_priors_lls);
  OFF(gpdla_spectra_cells, num_pixels); OFF(gpdla_spectra_cells, pixel_mask); OFF(gpdla_spectra_cells, log_priors_lls);
  OFF(gpdla_config, width); OFF(gpdla_config, max_dlas); OFF(gpdla_config, min_z_separation);
  OFF(gpdla_config, rng_seed); OFF(gpdla_config, contraction_precision);
  OFF(gpdla_config, multi_profile_bytes); OFF(gpdla_config, record_pool_bytes);
  OFF(gpdla_config, pipeline_slots); OFF(gpdla_config, max_quasars_per_batch);
  OFF(gpdla_results, status); OFF(gpdla_results, MAP_log_nhis);
  OFF(gpdla_results_multi, base_sample_inds); OFF(gpdla_results_multi, status);
  gpdla_config cfg;
  memset(&cfg, 0xAB, sizeof cfg);
  gpdla_default_config(&cfg);
  if (gpdla_abi_version() != GPDLA_ABI_VERSION) return 2;
  if (cfg.width != 3 || cfg.num_lines != 3 || cfg.max_dlas != 4 || cfg.min_lambda != 911.75) return 3;
  if (cfg.multi_profile_bytes != 0 || cfg.record_pool_bytes != 0) return 4;
  if (cfg.pipeline_slots != 0 || cfg.max_quasars_per_batch != 0) return 4;
  if (gpdla_default_batch_quasars(2048, 1500, 20, 10000, 3, 0, 0) != 256) return 7;
  uint32_t ctr[4] = {0, 0, 0, 0}, key[2] = {0, 0}, out[4];
  gpdla_debug_philox4x32_10(ctr, key, out);
  if (out[0] != 0x6627e8d5u) return 5;
  /* argument validation needs no GPU */
  double lam[6] = {1, 2, 3, 4, 5, 6}, prof[1];
  if (gpdla_voigt(lam, 6, 2.0, 1e20, 3, prof, 0) != GPDLA_ERR_INVALID_ARGUMENT) return 6;
  printf("GPDLA_SUMMARY_COLS %d\n", GPDLA_SUMMARY_COLS);
  printf("GPDLA_SUMMARY_COLS_MULTI4 %d\n", GPDLA_SUMMARY_COLS_MULTI(4));
  return 0;
}
"""


def test_c_consumer_links_against_the_header(lib, tmp_path):
    """include/gpdla.h compiles as C99, a C program links against libgpdla.so through it, and the
    struct layouts the C compiler produces equal the ctypes mirrors the Python host side uses."""
    import subprocess
    src = tmp_path / "consumer.c"
    src.write_text(C_CONSUMER)
    exe = tmp_path / "consumer"
    hip_rt = os.path.dirname(_lib._preload_hip_runtime()._name)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(exe), _lib.DEFAULT_LIB_PATH, "-L", hip_rt, "-lamdhip64",
                    f"-Wl,-rpath,{os.path.dirname(_lib.DEFAULT_LIB_PATH)}", f"-Wl,-rpath,{hip_rt}"],
                   check=True, capture_output=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True)
    assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)
    seen = dict(line.rsplit(" ", 1) for line in res.stdout.strip().splitlines())
    mirrors = {"gpdla_model": _lib.Model, "gpdla_samples": _lib.Samples, "gpdla_spectra": _lib.Spectra, "gpdla_spectra_cells": _lib.SpectraCells,
               "gpdla_config": _lib.Config, "gpdla_results": _lib.Results,
               "gpdla_results_multi": _lib.ResultsMulti}
    for cname, mirror in mirrors.items():
        assert int(seen[f"sizeof {cname}"]) == C.sizeof(mirror), cname
    for key, val in seen.items():
        if "." in key:
            cname, field = key.split(".")
            assert int(val) == getattr(mirrors[cname], field).offset, key
    assert int(seen["GPDLA_SUMMARY_COLS"]) == _lib.SUMMARY_COLS
    assert int(seen["GPDLA_SUMMARY_COLS_MULTI4"]) == _lib.summary_cols_multi(4) == 78


def test_philox4x32_10_known_answers(lib):
    """The generator behind the multi-DLA resampling (multi :467-472 stand-in) against the
    Random123 known-answer vectors (kat_vectors: philox4x32 10), and against an independent
    Python implementation on random counters/keys."""
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]

    def lib_philox(ctr, key):
        c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
        lib.gpdla_debug_philox4x32_10(c, k, o)
        return list(o)

    def py_philox(ctr, key):
        c, k = list(ctr), list(key)
        for _ in range(10):
            p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
            c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xffffffff]
            k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
        return c

    for ctr, key, want in kat:
        assert lib_philox(ctr, key) == want
        assert py_philox(ctr, key) == want
    rng = np.random.default_rng(3)
    for _ in range(50):
        ctr = [int(x) for x in rng.integers(0, 2**32, 4)]
        key = [int(x) for x in rng.integers(0, 2**32, 2)]
        assert lib_philox(ctr, key) == py_philox(ctr, key)


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


def _one_shot_args(nq=3, n=40, k=4, S=8):
    """Host arrays + structs for a gpdla_process_batch call (kept alive by the returned list)."""
    from gp_dla_detection_amd import api, synthetic
    model = synthetic.make_model(k)
    samples = synthetic.make_samples(S)
    spectra = synthetic.make_spectra(nq, n, model, first_index=5)
    csr = api.spectra_to_csr(spectra)
    keep = [csr]
    m, s_ = api._model_struct(model, keep), api._samples_struct(samples, keep)
    sp = api._spectra_struct(csr, np.full(nq, -1.0), np.full(nq, -2.0), None, keep)
    out = api.Batch.empty_results(nq, S)
    r = api._result_struct(_lib.Results, out)
    cfg = _lib.Config()
    _lib.load().gpdla_default_config(C.byref(cfg))
    return m, s_, sp, cfg, r, keep, out


def test_one_shot_entry_validates_before_it_touches_the_gpu(lib):
    """gpdla_process_batch: bad batching fields, decreasing offsets and a multi-DLA prior array are
    refused with GPDLA_ERR_INVALID_ARGUMENT whether or not a GPU exists."""
    m, s_, sp, cfg, r, keep, _ = _one_shot_args()
    cfg.pipeline_slots = -1
    assert lib.gpdla_process_batch(C.byref(m), C.byref(s_), C.byref(sp), C.byref(cfg), C.byref(r), 0) == -1
    assert b"pipeline_slots" in lib.gpdla_last_error()
    cfg.pipeline_slots = 0
    sp.offsets[2] = sp.offsets[1] - 1
    assert lib.gpdla_process_batch(C.byref(m), C.byref(s_), C.byref(sp), C.byref(cfg), C.byref(r), 0) == -1
    assert b"non-decreasing" in lib.gpdla_last_error()
    m, s_, sp, cfg, r, keep, _ = _one_shot_args()
    sp.log_priors_lls = sp.log_priors_no_dla
    assert lib.gpdla_process_batch(C.byref(m), C.byref(s_), C.byref(sp), C.byref(cfg), C.byref(r), 0) == -1
    assert lib.gpdla_process_batch(C.byref(m), C.byref(s_), None, C.byref(cfg), C.byref(r), 0) == -1


def test_no_cpp_exception_crosses_the_boundary(lib):
    """An exception that reaches an extern "C" frame ends the host process (MATLAB through the MEX gateways
    of integration/, Python through ctypes).  Every int-returning entry point of csrc/gpdla.hip is a
    function-try-block closed by the same handlers; the hook throws inside one."""
    assert lib.gpdla_debug_throw(0) == 0
    for kind, text in ((1, b"out of host memory"), (2, b"thrown on request"), (3, b"unexpected C++ exception")):
        assert lib.gpdla_debug_throw(kind) == _lib.ERR_HOST == -6
        assert text in lib.gpdla_last_error()
    # and every such entry point in the source is closed that way
    src = open(os.path.join(ROOT, "gp_dla_detection_amd", "csrc", "gpdla.hip")).read()
    blocks = re.findall(r'extern "C" \{(.*?)\}  // extern "C"', src, flags=re.S)
    defs = [m for b in blocks for m in re.findall(r"^int (gpdla_\w+)\([^;{]*?\)\s*(try )?\{", b, flags=re.M | re.S)]
    assert len(defs) >= 28
    unguarded = [name for name, guarded in defs if not guarded and name != "gpdla_abi_version"]
    assert unguarded == [], unguarded
    assert src.count("} GPDLA_NO_THROW") == sum(1 for _, g in defs if g)


def test_default_batch_rule(lib):
    """gpdla_default_batch_quasars: ~8 batches per run, >= 128 and <= 4096 quasars, `slots` batches within
    the HBM budget; the multi-DLA driver's batches are smaller (2 x models sample tables + records)."""
    f = lib.gpdla_default_batch_quasars
    assert f(2048, 1500, 20, 10000, 3, 0, 0) == 256
    assert f(100, 1500, 20, 10000, 3, 0, 0) == 128
    assert f(10 ** 6, 1500, 20, 10000, 3, 0, 0) == 4096
    assert f(10 ** 6, 1500, 40, 10000, 3, 2 ** 30, 0) < 800
    assert f(10 ** 6, 1500, 20, 10000, 3, 2 ** 33, 5) < f(10 ** 6, 1500, 20, 10000, 3, 2 ** 33, 0)
    assert f(1, 0, 20, 10, 0, 0, 0) >= 1


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
    # the one-shot entry (its upload / download threads never start without a device)
    m, s_, sp, cfg, r, keep, out = _one_shot_args()
    assert lib.gpdla_process_batch(C.byref(m), C.byref(s_), C.byref(sp), C.byref(cfg), C.byref(r), 0) == -2
    assert b"no CPU fallback" in lib.gpdla_last_error()
    assert np.isnan(out["log_likelihoods_dla"]).all()


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


def test_prepare_prior_clears_flags_for_dlas_below_the_lyman_limit():
    """process_qsos.m:11-27: a sightline stops counting as "has a DLA" when ALL its catalogued DLAs
    lie blueward of the quasar's Lyman limit (MATLAB's `if vector` is true only if every element is)."""
    import gp_dla_detection_amd as gp
    p = gp.Parameters()
    z_qsos = np.array([3.0, 3.0, 3.0, 2.5, 3.2])
    z_limit = p.lyman_limit * (1 + 3.0) / p.lya_wavelength - 1  # Lya of a DLA at the z = 3 quasar's limit
    z_dlas = [[z_limit - 0.01], [z_limit + 0.01], [z_limit - 0.2, z_limit + 0.3], [], [1.0]]
    dla_ind = np.array([True, True, True, False, False])
    out = gp.prepare_prior(z_qsos, dla_ind, z_dlas)
    assert out["dla_ind"].tolist() == [False, True, True, False, False]
    assert dla_ind.tolist() == [True, True, True, False, False]  # the input is not modified
    lp_no, lp_dla = gp.dla_existence_prior(out["z_qsos"], out["dla_ind"], np.array([3.5]))
    assert np.isclose(np.exp(lp_dla[0]), 2 / 5) and np.isclose(np.exp(lp_no[0]), 3 / 5)
    with pytest.raises(ValueError):
        gp.prepare_prior(z_qsos, dla_ind[:3], z_dlas)
