"""ctypes binding of libgpdla.so (the C-ABI declared in include/gpdla.h).

The library is built in-tree (``gp_dla_detection_amd/csrc/libgpdla.so``) by ``build()`` below or
``__graft_entry__.build()``.  There is no fallback: if the shared object is missing, or the GPU is,
the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
#: the product library
DEFAULT_LIB_PATH = os.path.join(CSRC, "libgpdla.so")
# GPDLA_LIB_PATH: diagnostic override, read when the library is loaded (ablation builds made by
# tools/ab_build.sh / tools/ablate.sh; libgpdla_legacy.so for the bit-identity tests)
LIB_PATH = os.environ.get("GPDLA_LIB_PATH") or DEFAULT_LIB_PATH


def lib_path() -> str:
    """The library load() opens: GPDLA_LIB_PATH if set (read at call time, so a child process may
    set it after this module was imported), else the in-tree product library."""
    return os.environ.get("GPDLA_LIB_PATH") or DEFAULT_LIB_PATH

# -no-hip-rt: libgpdla.so does NOT carry its own DT_NEEDED on libamdhip64.  A process must hold
# exactly one HIP runtime (a second copy cannot open the GPU, and a hipStream_t only means
# something to the runtime that made it), so the library binds to whichever runtime its host
# process already loaded: PyTorch's bundled one under Python (preloaded below), `-lamdhip64` for a
# C / MEX consumer.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared",
               "-std=c++17", "-no-hip-rt", "-Wno-inline-asm"]

#: the second library: the same source with the superseded kernels and their environment switches
#: compiled in (csrc/gpdla.hip, GPDLA_WITH_LEGACY).  Only bit-identity tests and A/B tools load it,
#: through GPDLA_LIB_PATH; nothing in the package does.
LEGACY_LIB_PATH = os.path.join(CSRC, "libgpdla_legacy.so")

_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)


# gpdla_status (include/gpdla.h)
ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_NOT_POSITIVE_DEFINITE, ERR_UNSUPPORTED, ERR_HOST = -1, -2, -3, -4, -5, -6


class GpdlaError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libgpdla error {code}: {message}")
        self.code = code


class Model(C.Structure):
    _fields_ = [("num_rest_pixels", C.c_int32), ("k", C.c_int32), ("rest_wavelengths", _dp),
                ("mu", _dp), ("M", _dp), ("log_omega", _dp), ("log_c_0", C.c_double),
                ("log_tau_0", C.c_double), ("log_beta", C.c_double)]


class Samples(C.Structure):
    _fields_ = [("num_dla_samples", C.c_int64), ("offset_samples", _dp), ("log_nhi_samples", _dp),
                ("nhi_samples", _dp), ("lls_nhi_samples", _dp)]


class Spectra(C.Structure):
    _fields_ = [("num_quasars", C.c_int64), ("offsets", _i64p), ("wavelengths", _dp),
                ("flux", _dp), ("noise_variance", _dp), ("pixel_mask", _u8p), ("z_qsos", _dp),
                ("log_priors_no_dla", _dp), ("log_priors_dla", _dp), ("log_priors_lls", _dp)]


class SpectraCells(C.Structure):
    """gpdla_spectra_cells: one array per quasar (pointer arrays are passed as void*: uintp NumPy arrays)."""
    _fields_ = [("num_quasars", C.c_int64), ("num_pixels", _i64p), ("wavelengths", C.c_void_p),
                ("flux", C.c_void_p), ("noise_variance", C.c_void_p), ("pixel_mask", C.c_void_p),
                ("z_qsos", _dp), ("log_priors_no_dla", _dp), ("log_priors_dla", _dp), ("log_priors_lls", _dp)]


class Config(C.Structure):
    _fields_ = [("min_lambda", C.c_double), ("max_lambda", C.c_double),
                ("lya_wavelength", C.c_double), ("lyman_limit", C.c_double),
                ("pixel_spacing", C.c_double), ("max_z_cut", C.c_double),
                ("min_z_cut", C.c_double), ("width", C.c_int32), ("num_lines", C.c_int32),
                ("max_dlas", C.c_int32), ("num_forest_lines", C.c_int32),
                ("min_z_separation", C.c_double), ("prev_tau_0", C.c_double),
                ("prev_beta", C.c_double), ("rng_seed", C.c_uint64),
                ("first_quasar_index", C.c_int64), ("contraction_precision", C.c_int32),
                ("multi_profile_bytes", C.c_int64), ("record_pool_bytes", C.c_int64),
                ("pipeline_slots", C.c_int32), ("max_quasars_per_batch", C.c_int64)]


class Results(C.Structure):
    _fields_ = [("min_z_dlas", _dp), ("max_z_dlas", _dp), ("log_likelihoods_no_dla", _dp),
                ("sample_log_likelihoods_dla", _dp), ("log_likelihoods_dla", _dp),
                ("log_posteriors_no_dla", _dp), ("log_posteriors_dla", _dp),
                ("model_posteriors", _dp), ("p_no_dlas", _dp), ("p_dlas", _dp),
                ("status", _i32p), ("MAP_inds", _dp), ("MAP_z_dlas", _dp), ("MAP_log_nhis", _dp)]


class ResultsMulti(C.Structure):
    _fields_ = [(n, _dp) for n in (
        "min_z_dlas", "max_z_dlas", "log_likelihoods_no_dla", "sample_log_likelihoods_dla",
        "sample_log_likelihoods_lls", "log_likelihoods_dla", "log_likelihoods_lls",
        "log_posteriors_no_dla", "log_posteriors_lls", "log_posteriors_dla", "model_posteriors",
        "p_no_dlas", "p_lls", "p_dlas", "MAP_z_dlas", "MAP_log_nhis", "MAP_inds")] + [
        ("base_sample_inds", _u32p), ("status", _i32p)]


SUMMARY_COLS = 15  # GPDLA_SUMMARY_COLS


def summary_cols_multi(max_dlas: int) -> int:
    """GPDLA_SUMMARY_COLS_MULTI"""
    return 14 + 4 * max_dlas + 3 * max_dlas * max_dlas


#: every symbol include/gpdla.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("gpdla_abi_version", C.c_int, []),
    ("gpdla_last_error", C.c_char_p, []),
    ("gpdla_voigt", C.c_int, [_dp, C.c_int64, C.c_double, C.c_double, C.c_int, _dp, C.c_int]),
    ("gpdla_log_mvnpdf_low_rank", C.c_int, [_dp, _dp, _dp, _dp, C.c_int64, C.c_int, _dp, C.c_int]),
    ("gpdla_default_config", None, [C.POINTER(Config)]),
    ("gpdla_process_batch", C.c_int, [C.POINTER(Model), C.POINTER(Samples), C.POINTER(Spectra),
                                      C.POINTER(Config), C.POINTER(Results), C.c_int]),
    ("gpdla_process_cells", C.c_int, [C.POINTER(Model), C.POINTER(Samples), C.POINTER(SpectraCells),
                                      C.POINTER(Config), C.POINTER(Results), C.c_int]),
    ("gpdla_process_cells_multi", C.c_int, [C.POINTER(Model), C.POINTER(Samples), C.POINTER(SpectraCells),
                                            _u32p, C.POINTER(Config), C.POINTER(ResultsMulti), C.c_int]),
    ("gpdla_default_batch_quasars", C.c_int64, [C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int]),
    ("gpdla_context_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("gpdla_context_destroy", None, [C.c_void_p]),
    ("gpdla_context_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("gpdla_context_set_model", C.c_int, [C.c_void_p, C.POINTER(Model)]),
    ("gpdla_context_set_samples", C.c_int, [C.c_void_p, C.POINTER(Samples)]),
    ("gpdla_context_set_config", C.c_int, [C.c_void_p, C.POINTER(Config)]),
    ("gpdla_context_set_first_quasar_index", C.c_int, [C.c_void_p, C.c_int64]),
    ("gpdla_context_synchronize", C.c_int, [C.c_void_p]),
    ("gpdla_batch_upload", C.c_int, [C.c_void_p, C.POINTER(Spectra), C.POINTER(C.c_void_p)]),
    ("gpdla_batch_reload", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Spectra)]),
    ("gpdla_batch_destroy", None, [C.c_void_p]),
    ("gpdla_batch_process", C.c_int, [C.c_void_p, C.c_void_p]),
    ("gpdla_batch_download", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Results)]),
    ("gpdla_batch_summary_device_ptr", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), _i64p]),
    ("gpdla_batch_samples_device_ptr", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), _i64p, _i64p]),
    ("gpdla_context_last_sweep_ms", C.c_double, [C.c_void_p]),
    ("gpdla_context_set_timing", C.c_int, [C.c_void_p, C.c_int]),
    ("gpdla_process_batch_multi", C.c_int, [C.POINTER(Model), C.POINTER(Samples), C.POINTER(Spectra),
                                            _u32p, C.POINTER(Config), C.POINTER(ResultsMulti), C.c_int]),
    ("gpdla_batch_process_multi", C.c_int, [C.c_void_p, C.c_void_p, _u32p]),
    ("gpdla_batch_download_multi", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(ResultsMulti)]),
    ("gpdla_batch_summary_multi_device_ptr", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), _i64p, _i32p]),
    ("gpdla_batch_samples_multi_device_ptr", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p),
                                                       C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("gpdla_training_create", C.c_int, [C.c_int, C.c_int64, C.c_int64, _dp, _dp, _dp,
                                        C.POINTER(C.c_void_p)]),
    ("gpdla_training_objective", C.c_int, [C.c_void_p, _dp, C.c_int, _dp, _dp]),
    ("gpdla_training_set_lyseries", C.c_int, [C.c_void_p, C.c_int, _dp, _dp]),
    ("gpdla_training_destroy", None, [C.c_void_p]),
    ("gpdla_debug_near_poly", C.c_int, [C.c_int, C.c_double, _dp, _dp]),
    ("gpdla_debug_prepared_rows", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, _dp, C.c_int64, _i64p]),
    ("gpdla_debug_philox4x32_10", None, [_u32p, _u32p, _u32p]),
    ("gpdla_debug_throw", C.c_int, [C.c_int]),
]

_lib = None


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/gpdla.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    import glob
    srcs = (glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp"))
            + glob.glob(os.path.join(_HERE, "..", "include", "*.h")))
    if not force and os.path.exists(DEFAULT_LIB_PATH):
        if all(os.path.getmtime(DEFAULT_LIB_PATH) >= os.path.getmtime(s) for s in srcs):
            return DEFAULT_LIB_PATH
    cmd = ["hipcc", *HIPCC_FLAGS, os.path.join(CSRC, "gpdla.hip"), "-o", DEFAULT_LIB_PATH]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode:
        print(res.stdout, res.stderr)
    if res.returncode:
        raise RuntimeError("hipcc failed building libgpdla.so:\n" + res.stderr)
    return DEFAULT_LIB_PATH


def build_legacy(force: bool = False, verbose: bool = False) -> str:
    """Compile libgpdla_legacy.so: csrc/gpdla.hip with -DGPDLA_WITH_LEGACY (the pre-expanded-record
    sweeps, the round-1/-3 training kernels and the GPDLA_* environment switches that select them)."""
    import glob
    srcs = (glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp"))
            + glob.glob(os.path.join(_HERE, "..", "include", "*.h")))
    if not force and os.path.exists(LEGACY_LIB_PATH):
        if all(os.path.getmtime(LEGACY_LIB_PATH) >= os.path.getmtime(s) for s in srcs):
            return LEGACY_LIB_PATH
    cmd = ["hipcc", *HIPCC_FLAGS, "-DGPDLA_WITH_LEGACY", os.path.join(CSRC, "gpdla.hip"), "-o", LEGACY_LIB_PATH]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode:
        print(res.stdout, res.stderr)
    if res.returncode:
        raise RuntimeError("hipcc failed building libgpdla_legacy.so:\n" + res.stderr)
    return LEGACY_LIB_PATH


def _preload_hip_runtime():
    """Put ONE HIP runtime in the global symbol scope before libgpdla.so is opened: the copy
    PyTorch-ROCm bundles when torch is installed (so torch streams/tensors and this library share
    a runtime), else the system ROCm one."""
    candidates = []
    try:
        import torch
        candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:  # torch is optional for the C-ABI itself
        pass
    candidates += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    last = None
    for path in candidates:
        if os.path.isabs(path) and not os.path.exists(path):
            continue
        try:
            return C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError as e:  # try the next candidate
            last = e
    raise OSError(f"no HIP runtime (libamdhip64.so) could be loaded: {last}")


def load():
    """dlopen libgpdla.so and type every declared symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`. "
            "gp_dla_detection_amd has no CPU fallback.")
    _preload_hip_runtime()
    lib = C.CDLL(path)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise GpdlaError(rc, load().gpdla_last_error().decode())
