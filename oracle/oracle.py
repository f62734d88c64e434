"""ctypes front-end for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module -- as the checker / reported CPU baseline.  The product package
(``gp_dla_detection_amd``) never does.

Besides the C restatement (``gpdla_oracle.c``) it holds one *independent* NumPy evaluation,
``dense_log_mvnpdf``, which forms K = M M' + diag(d) explicitly (slogdet + solve).  It is used to
validate the Woodbury restatement of log_mvnpdf_low_rank.m:5-34 at small n.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgpdla_oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_i64p = C.POINTER(C.c_int64)


class _Params(C.Structure):
    _fields_ = [("min_lambda", C.c_double), ("max_lambda", C.c_double),
                ("lya_wavelength", C.c_double), ("lyman_limit", C.c_double),
                ("pixel_spacing", C.c_double), ("max_z_cut", C.c_double),
                ("min_z_cut", C.c_double), ("width", C.c_int32), ("num_lines", C.c_int32)]


class _Model(C.Structure):
    _fields_ = [("num_rest", C.c_int32), ("k", C.c_int32), ("rest_wavelengths", _dp),
                ("mu", _dp), ("M", _dp), ("log_omega", _dp), ("log_c_0", C.c_double),
                ("log_tau_0", C.c_double), ("log_beta", C.c_double)]


class _Dump(C.Structure):
    _fields_ = [("n_kept", _i64p), ("n_unmasked", _i64p), ("this_mu", _dp), ("this_M", _dp),
                ("this_omega2", _dp), ("padded_wavelengths", _dp), ("sample_z_dlas", _dp)]


class _Multi(C.Structure):
    _fields_ = [("max_dlas", C.c_int32), ("num_forest_lines", C.c_int32),
                ("min_z_separation", C.c_double), ("prev_tau_0", C.c_double),
                ("prev_beta", C.c_double), ("lls_nhi_samples", _dp),
                ("base_sample_inds", _u32p), ("log_nhi_samples", _dp)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (make).  Building the checker is not using it."""
    src = os.path.join(_HERE, "gpdla_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _LIB_PATH


def _bind(lib):
    lib.gpdla_oracle_faddeeva_re.restype = C.c_double
    lib.gpdla_oracle_faddeeva_re.argtypes = [C.c_double, C.c_double]
    lib.gpdla_oracle_voigt_line.restype = C.c_double
    lib.gpdla_oracle_voigt_line.argtypes = [C.c_double, C.c_double, C.c_double]
    for name in ("gpdla_oracle_voigt", "gpdla_oracle_voigt_raw"):
        f = getattr(lib, name)
        f.restype = C.c_int
        f.argtypes = [_dp, C.c_int64, C.c_double, C.c_double, C.c_int, _dp]
    lib.gpdla_oracle_log_mvnpdf_low_rank.restype = C.c_int
    lib.gpdla_oracle_log_mvnpdf_low_rank.argtypes = [_dp, _dp, _dp, _dp, C.c_int64, C.c_int, _dp]
    lib.gpdla_oracle_process_spectrum.restype = C.c_int
    lib.gpdla_oracle_process_spectrum.argtypes = [
        C.POINTER(_Params), C.POINTER(_Model), C.c_int64, _dp, _dp, C.c_int64, _dp, _dp, _dp,
        _u8p, C.c_double, C.c_int, _dp, _dp, _dp, _dp, _dp, C.POINTER(_Dump)]
    lib.gpdla_oracle_process_spectrum_multi.restype = C.c_int
    lib.gpdla_oracle_process_spectrum_multi.argtypes = [
        C.POINTER(_Params), C.POINTER(_Model), C.POINTER(_Multi), C.c_int64, _dp, _dp, C.c_int64,
        _dp, _dp, _dp, _u8p, C.c_double, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
    lib.gpdla_oracle_mean_flux_suppression.restype = C.c_double
    lib.gpdla_oracle_mean_flux_suppression.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                                       C.c_int]
    lib.gpdla_oracle_objective.restype = C.c_int
    lib.gpdla_oracle_objective.argtypes = [_dp, C.c_int64, C.c_int64, C.c_int, _dp, _dp, _dp, C.c_int,
                                           _dp, _dp]
    return lib


def load():
    global _lib
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(_LIB_PATH))
    return _lib


_timing_lib = None


def load_timing():
    """The timing build of the same source (-O3 -march=native; ``make timing``), compiled ON the
    machine that calls this -- a -march=native object must not travel between hosts -- for the
    ``cpu_baseline`` leg of bench.py."""
    global _timing_lib
    if _timing_lib is None:
        subprocess.run(["make", "-B", "-C", _HERE, "timing"], check=True, capture_output=True)
        _timing_lib = _bind(C.CDLL(os.path.join(_HERE, "_build", "libgpdla_oracle_timing.so")))
    return _timing_lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def faddeeva_re(x, y):
    lib = load()
    return np.vectorize(lambda a, b: lib.gpdla_oracle_faddeeva_re(float(a), float(b)))(x, y)


def voigt_line(x, sigma, gamma):
    return load().gpdla_oracle_voigt_line(float(x), float(sigma), float(gamma))


def voigt(lambdas, z, N, num_lines=31, raw=False):
    """voigt.c:253-304 (raw=True: before the instrument convolution, as voigt.py:230-275)."""
    lib = load()
    lam, lp = _d(lambdas)
    out = np.zeros(lam.size if raw else lam.size - 6)
    f = lib.gpdla_oracle_voigt_raw if raw else lib.gpdla_oracle_voigt
    rc = f(lp, lam.size, float(z), float(N), int(num_lines), out.ctypes.data_as(_dp))
    if rc:
        raise ValueError(f"oracle voigt rc={rc}")
    return out


def mean_flux_suppression(wavelengths, z_qso, prev_tau_0=0.0023, prev_beta=3.65, num_forest_lines=31,
                          lya_wavelength=1215.6701):
    """exp(-Sum_l tau_l (1 + z_l)^beta) per observed pixel, multi :267-285 (the factor the multi-DLA
    driver multiplies mu and M by)."""
    lib = load()
    return np.array([lib.gpdla_oracle_mean_flux_suppression(float(w), float(z_qso), float(lya_wavelength),
                                                            float(prev_tau_0), float(prev_beta), int(num_forest_lines))
                     for w in np.asarray(wavelengths, dtype=np.float64).reshape(-1)])


def log_mvnpdf_low_rank(y, mu, M, d):
    """log_mvnpdf_low_rank.m:5-34; M is (n, k)."""
    lib = load()
    y, yp = _d(y)
    mu, mup = _d(mu)
    Mf = np.asfortranarray(M, dtype=np.float64)
    d, dpp = _d(d)
    out = C.c_double()
    rc = lib.gpdla_oracle_log_mvnpdf_low_rank(yp, mup, Mf.ctypes.data_as(_dp), dpp, y.size,
                                              Mf.shape[1], C.byref(out))
    return out.value, rc


def dense_log_mvnpdf(y, mu, M, d):
    """Independent check: log N(y; mu, M M' + diag(d)) with K formed explicitly."""
    y = np.asarray(y, dtype=np.float64) - np.asarray(mu, dtype=np.float64)
    M = np.asarray(M, dtype=np.float64)
    K = M @ M.T + np.diag(np.asarray(d, dtype=np.float64))
    sign, logdet = np.linalg.slogdet(K)
    assert sign > 0
    quad = y @ np.linalg.solve(K, y)
    return -0.5 * (quad + logdet + y.size * np.log(2 * np.pi))


@dataclass
class OracleParams:
    """set_parameters.m:5-73 -- the values process_qsos.m reads."""
    min_lambda: float = 911.75
    max_lambda: float = 1215.75
    lya_wavelength: float = 1215.6701
    lyman_limit: float = 911.7633
    pixel_spacing: float = 1e-4
    max_z_cut: float = 3000 * 1000 / 299792458
    min_z_cut: float = 3000 * 1000 / 299792458
    width: int = 3
    num_lines: int = 3

    def c(self):
        return _Params(self.min_lambda, self.max_lambda, self.lya_wavelength, self.lyman_limit,
                       self.pixel_spacing, self.max_z_cut, self.min_z_cut, self.width,
                       self.num_lines)


def _model_struct(model):
    keep = []
    rw, rwp = _d(model["rest_wavelengths"])
    mu, mup = _d(model["mu"])
    Mf = np.asfortranarray(model["M"], dtype=np.float64)
    lo, lop = _d(model["log_omega"])
    keep += [rw, mu, Mf, lo]
    m = _Model(rw.size, Mf.shape[1], rwp, mup, Mf.ctypes.data_as(_dp), lop,
               float(model["log_c_0"]), float(model["log_tau_0"]), float(model["log_beta"]))
    return m, keep


def process_spectrum(model, offset_samples, nhi_samples, wavelengths, flux, noise_variance,
                     pixel_mask, z_qso, params: OracleParams | None = None, num_threads=0,
                     dump=False, lib=None):
    """process_qsos.m:96-213 for one quasar.  Returns a dict (plus intermediates if dump).
    ``lib``: the build to call (default: the checker build; bench.py passes the timing build)."""
    lib = lib or load()
    params = params or OracleParams()
    prm = params.c()
    mdl, keep = _model_struct(model)
    off, offp = _d(offset_samples)
    nhi, nhip = _d(nhi_samples)
    wl, wlp = _d(wavelengths)
    fl, flp = _d(flux)
    nv, nvp = _d(noise_variance)
    mk = np.ascontiguousarray(pixel_mask, dtype=np.uint8)
    S, npx, k = off.size, wl.size, mdl.k
    sll = np.full(S, np.nan)
    zmin, zmax, ll0, ll1 = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    dptr = None
    if dump:
        nk, nu = C.c_int64(), C.c_int64()
        tmu = np.zeros(npx)
        tM = np.zeros(npx * k)
        tom = np.zeros(npx)
        pad = np.zeros(npx + 6)
        sz = np.zeros(S)
        dd = _Dump(C.pointer(nk), C.pointer(nu), tmu.ctypes.data_as(_dp), tM.ctypes.data_as(_dp),
                   tom.ctypes.data_as(_dp), pad.ctypes.data_as(_dp), sz.ctypes.data_as(_dp))
        dptr = C.byref(dd)
    rc = lib.gpdla_oracle_process_spectrum(
        C.byref(prm), C.byref(mdl), S, offp, nhip, npx, wlp, flp, nvp, mk.ctypes.data_as(_u8p),
        float(z_qso), int(num_threads), C.byref(zmin), C.byref(zmax), C.byref(ll0),
        sll.ctypes.data_as(_dp), C.byref(ll1), dptr)
    out = dict(rc=rc, min_z_dla=zmin.value, max_z_dla=zmax.value,
               log_likelihood_no_dla=ll0.value, sample_log_likelihoods_dla=sll,
               log_likelihood_dla=ll1.value)
    if dump and rc == 0:
        n, n_u = nk.value, nu.value
        out.update(n_kept=n, n_unmasked=n_u, this_mu=tmu[:n].copy(),
                   this_M=tM[:n * k].reshape(k, n).T.copy(), this_omega2=tom[:n].copy(),
                   padded_wavelengths=pad[:n_u + 6].copy(), sample_z_dlas=sz)
    return out


def process_spectrum_multi(model, offset_samples, nhi_samples, log_nhi_samples, lls_nhi_samples,
                           base_sample_inds, wavelengths, flux, noise_variance, pixel_mask, z_qso,
                           params: OracleParams | None = None, max_dlas=4, num_forest_lines=31,
                           min_z_separation=3000 * 1000 / 299792458, prev_tau_0=0.0023,
                           prev_beta=3.65, num_threads=0):
    """multi_dlas/process_qsos_multiple_dlas_meanflux.m:141-477 for one quasar.

    base_sample_inds: (max_dlas-1, S) uint32, 1-based (MATLAB this_base_sample_inds, :313)."""
    lib = load()
    params = params or OracleParams()
    prm = params.c()
    mdl, keep = _model_struct(model)
    off, offp = _d(offset_samples)
    nhi, nhip = _d(nhi_samples)
    lnhi, lnhip = _d(log_nhi_samples)
    lls, llsp = _d(lls_nhi_samples)
    S = off.size
    bsi = np.asfortranarray(base_sample_inds, dtype=np.uint32)
    assert bsi.shape == (max_dlas - 1, S)
    wl, wlp = _d(wavelengths)
    fl, flp = _d(flux)
    nv, nvp = _d(noise_variance)
    mk = np.ascontiguousarray(pixel_mask, dtype=np.uint8)
    mul = _Multi(max_dlas, num_forest_lines, float(min_z_separation), float(prev_tau_0),
                 float(prev_beta), llsp, bsi.ctypes.data_as(_u32p), lnhip)
    sll = np.full((S, max_dlas), np.nan, order="F")
    ll_dla = np.full(max_dlas, np.nan)
    sll_lls = np.full(S, np.nan)
    mapz = np.full((max_dlas, max_dlas), np.nan, order="F")
    mapn = np.full((max_dlas, max_dlas), np.nan, order="F")
    mapi = np.full((max_dlas, max_dlas), np.nan, order="F")
    zmin, zmax, ll0, ll_lls = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    rc = lib.gpdla_oracle_process_spectrum_multi(
        C.byref(prm), C.byref(mdl), C.byref(mul), S, offp, nhip, wl.size, wlp, flp, nvp,
        mk.ctypes.data_as(_u8p), float(z_qso), int(num_threads), C.byref(zmin), C.byref(zmax),
        C.byref(ll0), sll.ctypes.data_as(_dp), ll_dla.ctypes.data_as(_dp),
        sll_lls.ctypes.data_as(_dp), C.byref(ll_lls), mapz.ctypes.data_as(_dp),
        mapn.ctypes.data_as(_dp), mapi.ctypes.data_as(_dp))
    return dict(rc=rc, min_z_dla=zmin.value, max_z_dla=zmax.value,
                log_likelihood_no_dla=ll0.value, sample_log_likelihoods_dla=np.array(sll),
                log_likelihoods_dla=ll_dla, sample_log_likelihoods_lls=sll_lls,
                log_likelihood_lls=ll_lls.value, MAP_z_dlas=np.array(mapz),
                MAP_log_nhis=np.array(mapn), MAP_inds=np.array(mapi))


def objective_lyseries(x, centered_rest_fluxes, lya_1pzs, rest_noise_variances, num_forest_lines,
                       all_transition_wavelengths, all_oscillator_strengths, num_threads=0):
    """multi_dlas/objective_lyseries.m:12-78 (same arguments): (f, g) of the mean-flux model's
    training objective."""
    lib = load()
    F = np.asfortranarray(centered_rest_fluxes, dtype=np.float64)
    Lz = np.asfortranarray(lya_1pzs, dtype=np.float64)
    Nv = np.asfortranarray(rest_noise_variances, dtype=np.float64)
    nq, G = F.shape
    x, xp = _d(x)
    k = (x.size - 3) // G - 1
    assert x.size == G * (k + 1) + 3
    wl, wlp = _d(all_transition_wavelengths)
    fs, fsp = _d(all_oscillator_strengths)
    assert wl.size >= num_forest_lines and fs.size >= num_forest_lines
    f = C.c_double()
    g = np.zeros_like(x)
    lib.gpdla_oracle_objective_lyseries.restype = C.c_int
    lib.gpdla_oracle_objective_lyseries.argtypes = [_dp, C.c_int64, C.c_int64, C.c_int, _dp, _dp, _dp, C.c_int,
                                                    _dp, _dp, C.c_int, _dp, _dp]
    rc = lib.gpdla_oracle_objective_lyseries(xp, nq, G, k, F.ctypes.data_as(_dp), Lz.ctypes.data_as(_dp),
                                             Nv.ctypes.data_as(_dp), int(num_forest_lines), wlp, fsp,
                                             int(num_threads), C.byref(f), g.ctypes.data_as(_dp))
    if rc:
        raise ValueError("oracle objective_lyseries: B not positive definite" if rc == -1 else "bad arguments")
    return f.value, g


def objective(x, centered_rest_fluxes, lya_1pzs, rest_noise_variances, num_threads=0):
    """objective.m:12-75: (f, g) for x = [vec M; log omega; log c0; log tau0; log beta]; the data
    matrices are (num_quasars, num_pixels) with NaN for missing pixels."""
    lib = load()
    F = np.asfortranarray(centered_rest_fluxes, dtype=np.float64)
    Lz = np.asfortranarray(lya_1pzs, dtype=np.float64)
    Nv = np.asfortranarray(rest_noise_variances, dtype=np.float64)
    nq, G = F.shape
    x, xp = _d(x)
    k = (x.size - 3) // G - 1
    assert x.size == G * (k + 1) + 3
    f = C.c_double()
    g = np.zeros_like(x)
    rc = lib.gpdla_oracle_objective(xp, nq, G, k, F.ctypes.data_as(_dp), Lz.ctypes.data_as(_dp),
                                    Nv.ctypes.data_as(_dp), int(num_threads), C.byref(f),
                                    g.ctypes.data_as(_dp))
    if rc:
        raise ValueError("oracle objective: B not positive definite")
    return f.value, g
