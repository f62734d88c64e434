"""Python host side of the MI355X inference sweep: the reference's call surface over libgpdla.so.

Mirrors, name for name, what a user of jibanCat/gp_dla_detection calls on this path:

=========================================  =====================================================
reference                                  here
=========================================  =====================================================
``voigt(lambdas, z, N, num_lines)`` MEX    :func:`voigt`                      (voigt.c:253-304)
``log_mvnpdf_low_rank(y, mu, M, d)``       :func:`log_mvnpdf_low_rank`        (log_mvnpdf_low_rank.m:5)
DLA-existence prior of the driver          :func:`dla_existence_prior`        (process_qsos.m:122-131)
``process_qsos`` script                    :func:`process_qsos`               (process_qsos.m:88-244)
``process_qsos_multiple_dlas_meanflux``    :func:`process_qsos_multiple_dlas_meanflux`
                                                                              (multi_dlas/...m:141-510)
=========================================  =====================================================

plus :class:`Context` / :class:`Batch` for callers that keep spectra resident in HBM (what
``bench.py`` times, and what the multi-GPU driver in :mod:`.distributed` uses).  NumPy arrays and
PyTorch-ROCm tensors are staging only; all arithmetic happens in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .parameters import MultiParameters, Parameters

_dp = C.POINTER(C.c_double)


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _config(params: Parameters) -> _lib.Config:
    lib = _lib.load()
    cfg = _lib.Config()
    lib.gpdla_default_config(C.byref(cfg))
    for name in ("min_lambda", "max_lambda", "lya_wavelength", "lyman_limit", "pixel_spacing",
                 "max_z_cut", "min_z_cut", "width", "num_lines"):
        setattr(cfg, name, getattr(params, name))
    for name in ("max_dlas", "num_forest_lines", "min_z_separation", "prev_tau_0", "prev_beta",
                 "rng_seed", "first_quasar_index", "contraction_precision", "multi_profile_bytes",
                 "record_pool_bytes"):
        if hasattr(params, name):
            setattr(cfg, name, getattr(params, name))
    return cfg


# ----------------------------------------------------------------------------------------------
# stand-alone surfaces
# ----------------------------------------------------------------------------------------------

def voigt(lambdas, z, N, num_lines: int = 31, device: int = 0) -> np.ndarray:
    """``profile = voigt(lambdas, z, N[, num_lines])`` -- voigt.c:253-304.

    Returns ``numel(lambdas) - 6`` values (the MEX trims ``width = 3`` pixels per side, :271).
    ``num_lines`` defaults to 31 like the MEX (:16, :266)."""
    lib = _lib.load()
    lam, lp = _f64(lambdas)
    if lam.size <= 6:
        raise _lib.GpdlaError(-1, "lambdas must have more than 2*width = 6 entries")
    out = np.empty(lam.size - 6)
    _lib.check(lib.gpdla_voigt(lp, lam.size, float(z), float(N), int(num_lines),
                               out.ctypes.data_as(_dp), int(device)))
    return out


def log_mvnpdf_low_rank(y, mu, M, d, device: int = 0) -> float:
    """``log_p = log_mvnpdf_low_rank(y, mu, M, d)`` -- log N(y; mu, M M' + diag(d))
    (log_mvnpdf_low_rank.m:5-34).  ``M`` is (n, k).  Raises GpdlaError(-4) where MATLAB's
    ``chol`` would throw (:24)."""
    lib = _lib.load()
    y, yp = _f64(y)
    mu, mup = _f64(mu)
    d, dp = _f64(d)
    Mf = np.asfortranarray(M, dtype=np.float64)
    if Mf.ndim != 2 or Mf.shape[0] != y.size or mu.size != y.size or d.size != y.size:
        raise _lib.GpdlaError(-1, "shape mismatch: y, mu, d are n-vectors and M is n x k")
    out = C.c_double()
    _lib.check(lib.gpdla_log_mvnpdf_low_rank(yp, mup, Mf.ctypes.data_as(_dp), dp, y.size,
                                             Mf.shape[1], C.byref(out), int(device)))
    return out.value


def prepare_prior(prior_z_qsos, prior_dla_ind, prior_z_dlas, params: Parameters | None = None) -> dict:
    """process_qsos.m:11-27: the training catalogue's (z_QSO, has-a-DLA) pairs behind the model
    prior, with a sightline's flag cleared when every one of its catalogued DLAs lies blueward of the
    quasar's Lyman limit -- ``observed_wavelengths(lya_wavelength, z_dla) <
    observed_wavelengths(lyman_limit, z_qso)`` (:21-25) -- where this search never looks.
    ``prior_z_dlas[i]`` is the list of DLA redshifts of sightline i (the cell of
    ``prior_catalog.z_dlas(dla_catalog_name)``); it is read only where ``prior_dla_ind[i]``.
    Returns ``dict(z_qsos, dla_ind)`` as :func:`process_qsos` takes for ``prior_catalog``."""
    p = params or Parameters()
    z = np.asarray(prior_z_qsos, dtype=np.float64).reshape(-1)
    ind = np.array(prior_dla_ind, dtype=bool).reshape(-1)
    if ind.size != z.size or len(prior_z_dlas) != z.size:
        raise ValueError("prior_z_qsos, prior_dla_ind and prior_z_dlas must have one entry per sightline")
    for i in np.flatnonzero(ind):
        z_dlas = np.atleast_1d(np.asarray(prior_z_dlas[i], dtype=np.float64))
        # MATLAB's `if (vector < scalar)` is true only when every element is
        if z_dlas.size and np.all(p.lya_wavelength * (1 + z_dlas) < p.lyman_limit * (1 + z[i])):
            ind[i] = False
    return dict(z_qsos=z, dla_ind=ind)


def dla_existence_prior(prior_z_qsos, prior_dla_ind, z_qsos, params: Parameters | None = None):
    """process_qsos.m:122-131: log p(DLA | z_QSO) and log p(no DLA | z_QSO) from the counts of
    training-catalog quasars with z < z_QSO + prior_z_qso_increase.  Host logic (SURVEY.md
    section 8a row A3)."""
    p = params or Parameters()
    pz = np.asarray(prior_z_qsos, dtype=np.float64)
    pd = np.asarray(prior_dla_ind, dtype=bool)
    z = np.atleast_1d(np.asarray(z_qsos, dtype=np.float64))
    order = np.argsort(pz, kind="stable")
    pz_sorted = pz[order]
    cum_dla = np.concatenate([[0], np.cumsum(pd[order])])
    num_quasars = np.searchsorted(pz_sorted, z + p.prior_z_qso_increase, side="left")  # strict <
    num_dlas = cum_dla[num_quasars]
    with np.errstate(divide="ignore", invalid="ignore"):
        log_dla = np.log(num_dlas.astype(np.float64)) - np.log(num_quasars.astype(np.float64))
        log_no = (np.log((num_quasars - num_dlas).astype(np.float64))
                  - np.log(num_quasars.astype(np.float64)))
    return log_no, log_dla


# ----------------------------------------------------------------------------------------------
# resident form
# ----------------------------------------------------------------------------------------------

class _DeviceArray:
    """Zero-copy view of library-owned HBM for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr: int, shape, owner):
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": "<f8",
                                         "data": (int(ptr), False), "version": 2}
        self._owner = owner  # keeps the batch alive


def spectra_to_csr(spectra):
    """The ragged cell arrays of preloaded_qsos.mat (preload_qsos.m:64-79) as flat CSR arrays."""
    sizes = [np.asarray(s["wavelengths"]).size for s in spectra]
    offsets = np.zeros(len(spectra) + 1, dtype=np.int64)
    np.cumsum(sizes, out=offsets[1:])
    cat = lambda key, dt: (np.concatenate([np.asarray(s[key], dtype=dt).ravel() for s in spectra])
                           if spectra else np.zeros(0, dt))
    return dict(offsets=offsets, wavelengths=cat("wavelengths", np.float64),
                flux=cat("flux", np.float64), noise_variance=cat("noise_variance", np.float64),
                pixel_mask=cat("pixel_mask", np.uint8),
                z_qsos=np.array([float(s["z_qso"]) for s in spectra], dtype=np.float64))


class Context:
    """A device + stream + the replicated GP model and DLA samples (gpdla_context)."""

    def __init__(self, device: int = 0, params: Parameters | None = None, stream=None):
        self.lib = _lib.load()
        self.device = int(device)
        self.params = params or Parameters()
        self._h = C.c_void_p()
        _lib.check(self.lib.gpdla_context_create(self.device, C.byref(self._h)))
        cfg = _config(self.params)
        _lib.check(self.lib.gpdla_context_set_config(self._h, C.byref(cfg)))
        if stream is not None:
            self.set_stream(stream)
        self.num_samples = 0
        self.k = 0

    def set_stream(self, stream):
        """``stream``: a raw hipStream_t (int) or a torch.cuda.Stream."""
        ptr = getattr(stream, "cuda_stream", stream)
        _lib.check(self.lib.gpdla_context_set_stream(self._h, C.c_void_p(int(ptr) if ptr else None)))

    def set_model(self, model: dict):
        """Fields of learned_qso_model_*.mat (process_qsos.m:30-35)."""
        keep = []
        m = _model_struct(model, keep)
        _lib.check(self.lib.gpdla_context_set_model(self._h, C.byref(m)))
        self.k = int(m.k)

    def set_samples(self, samples: dict):
        """Fields of dla_samples.mat (process_qsos.m:38-40)."""
        keep = []
        s = _samples_struct(samples, keep)
        _lib.check(self.lib.gpdla_context_set_samples(self._h, C.byref(s)))
        self.num_samples = int(s.num_dla_samples)

    def set_timing(self, enabled: bool):
        _lib.check(self.lib.gpdla_context_set_timing(self._h, int(bool(enabled))))

    def last_sweep_ms(self) -> float:
        return float(self.lib.gpdla_context_last_sweep_ms(self._h))

    def synchronize(self):
        _lib.check(self.lib.gpdla_context_synchronize(self._h))

    def upload(self, spectra, log_priors_no_dla, log_priors_dla, log_priors_lls=None) -> "Batch":
        """Spectra + priors to HBM.  With ``log_priors_lls`` the batch is a multi-DLA batch
        (``log_priors_dla`` is then [nq, max_dlas]) and is swept with :meth:`Batch.process_multi`."""
        return Batch(self, spectra, log_priors_no_dla, log_priors_dla, log_priors_lls)

    def set_params(self, params: Parameters):
        """Replace the whole configuration.  Not while another thread uploads a batch of this
        context (the upload reads it): a pipeline sets the per-batch key of the multi-DLA resampling
        with :meth:`set_first_quasar_index` instead."""
        self.params = params
        cfg = _config(params)
        _lib.check(self.lib.gpdla_context_set_config(self._h, C.byref(cfg)))

    def set_first_quasar_index(self, index: int):
        """The global index of the next multi-DLA batch's first quasar (keys the Philox resampling,
        multi :467-472).  Safe beside a concurrent upload (gpdla_context_set_first_quasar_index)."""
        _lib.check(self.lib.gpdla_context_set_first_quasar_index(self._h, int(index)))

    def close(self):
        if self._h:
            self.lib.gpdla_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """A CSR batch of spectra resident in HBM together with its result tables (gpdla_batch)."""

    def __init__(self, ctx: Context, spectra, log_priors_no_dla, log_priors_dla, log_priors_lls=None):
        self.ctx = ctx
        self._h = C.c_void_p()
        self._fill(spectra, log_priors_no_dla, log_priors_dla, log_priors_lls)

    def reload(self, spectra, log_priors_no_dla, log_priors_dla, log_priors_lls=None):
        """Replace the batch's spectra in place (gpdla_batch_reload): device allocations are reused
        where large enough, so the batch slots of a pipeline allocate nothing in the steady state.
        The previous results must have been downloaded."""
        self._fill(spectra, log_priors_no_dla, log_priors_dla, log_priors_lls)

    def _fill(self, spectra, log_priors_no_dla, log_priors_dla, log_priors_lls):
        ctx = self.ctx
        csr = spectra if isinstance(spectra, dict) else spectra_to_csr(spectra)
        self.num_quasars = csr["z_qsos"].size
        self.num_samples = ctx.num_samples
        self.max_dlas = int(getattr(ctx.params, "max_dlas", 0)) if log_priors_lls is not None else 0
        keep = []

        def ptr(a, dt, ct):
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(ct))

        lp_dla = np.asarray(log_priors_dla, dtype=np.float64)
        if self.max_dlas:
            lp_dla = lp_dla.reshape(self.num_quasars, self.max_dlas)
        sp = _lib.Spectra(
            self.num_quasars, ptr(csr["offsets"], np.int64, C.c_int64),
            ptr(csr["wavelengths"], np.float64, C.c_double), ptr(csr["flux"], np.float64, C.c_double),
            ptr(csr["noise_variance"], np.float64, C.c_double),
            ptr(csr["pixel_mask"], np.uint8, C.c_uint8), ptr(csr["z_qsos"], np.float64, C.c_double),
            ptr(log_priors_no_dla, np.float64, C.c_double), ptr(lp_dla, np.float64, C.c_double),
            ptr(log_priors_lls, np.float64, C.c_double) if self.max_dlas else None)
        if not self._h:
            _lib.check(ctx.lib.gpdla_batch_upload(ctx._h, C.byref(sp), C.byref(self._h)))
        else:
            _lib.check(ctx.lib.gpdla_batch_reload(ctx._h, self._h, C.byref(sp)))
        self.log_priors_no_dla = np.array(log_priors_no_dla, dtype=np.float64)
        self.log_priors_dla = np.array(lp_dla, dtype=np.float64)
        self.log_priors_lls = None if log_priors_lls is None else np.array(log_priors_lls, dtype=np.float64)

    def process(self):
        """Launch the sweep for every quasar of the batch (asynchronous on the context's stream)."""
        _lib.check(self.ctx.lib.gpdla_batch_process(self.ctx._h, self._h))

    _SINGLE_VECTORS = ("min_z_dlas", "max_z_dlas", "log_likelihoods_no_dla", "log_likelihoods_dla",
                       "log_posteriors_no_dla", "log_posteriors_dla", "p_no_dlas", "p_dlas",
                       "MAP_inds", "MAP_z_dlas", "MAP_log_nhis")  # MAP_*: generate_ascii_catalog.m:73-80

    @classmethod
    def empty_results(cls, nq: int, S: int, with_samples: bool = True) -> dict:
        """Host arrays for the variables process_qsos.m:236-244 saves, for ``nq`` quasars."""
        out = {name: np.full(nq, np.nan) for name in cls._SINGLE_VECTORS}
        out["model_posteriors"] = np.full((nq, 2), np.nan)
        out["status"] = np.zeros(nq, dtype=np.int32)
        out["log_priors_no_dla"] = np.full(nq, np.nan)
        out["log_priors_dla"] = np.full(nq, np.nan)
        if with_samples:  # every row is written by a download (the device table is NaN-prefilled)
            out["sample_log_likelihoods_dla"] = np.empty((nq, S))
        return out

    def download(self, with_samples: bool = True, out: dict | None = None, at: int = 0) -> dict:
        """Results under the field names process_qsos.m:236-244 saves.  ``out`` / ``at``: write this
        batch's rows into rows ``at .. at + num_quasars`` of arrays made by :meth:`empty_results`
        (a pipeline's one set of output arrays) instead of allocating."""
        nq = self.num_quasars
        if out is None:
            out, at = self.empty_results(nq, self.num_samples, with_samples), 0
        r = _lib.Results()
        for name, _ in _lib.Results._fields_:
            if name in out:
                ct = C.c_int32 if name == "status" else C.c_double
                view = out[name][at:at + nq]
                assert view.flags.c_contiguous and view.shape[0] == nq
                setattr(r, name, view.ctypes.data_as(C.POINTER(ct)))
        _lib.check(self.ctx.lib.gpdla_batch_download(self.ctx._h, self._h, C.byref(r)))
        out["log_priors_no_dla"][at:at + nq] = self.log_priors_no_dla
        out["log_priors_dla"][at:at + nq] = self.log_priors_dla
        return out

    def debug_prepared_rows(self, quasar: int = 0, multi: bool = False) -> np.ndarray:
        """Test hook (gpdla_debug_prepared_rows): the (y, mu, omega2, nu) rows of one quasar on the
        unmasked-range grid after the preparation kernel alone, [n_u, 4]."""
        cap = 8192
        rows = np.empty((cap, 4))
        n = C.c_int64()
        _lib.check(self.ctx.lib.gpdla_debug_prepared_rows(self.ctx._h, self._h, int(bool(multi)), int(quasar),
                                                          rows.ctypes.data_as(_dp), cap, C.byref(n)))
        return rows[: n.value].copy()

    def summary_tensor(self):
        """The per-quasar summary table as a zero-copy torch tensor on this GPU: [nq, 15] for a
        single-DLA batch, [nq, GPDLA_SUMMARY_COLS_MULTI(max_dlas)] (78 for max_dlas = 4) for a
        multi-DLA batch.  This is the row a multi-GPU run all-gathers."""
        import torch
        p, n = C.c_void_p(), C.c_int64()
        if self.max_dlas:
            cols = C.c_int32()
            _lib.check(self.ctx.lib.gpdla_batch_summary_multi_device_ptr(self._h, C.byref(p), C.byref(n),
                                                                         C.byref(cols)))
            ncol = cols.value
        else:
            _lib.check(self.ctx.lib.gpdla_batch_summary_device_ptr(self._h, C.byref(p), C.byref(n)))
            ncol = _lib.SUMMARY_COLS
        return torch.as_tensor(_DeviceArray(p.value, (n.value, ncol), self),
                               device=f"cuda:{self.ctx.device}")

    # ---- multi-DLA batch (process_qsos_multiple_dlas_meanflux.m:141-495) ----

    def process_multi(self, base_sample_inds=None):
        """Launch the multi-DLA driver for every quasar of the batch.  ``base_sample_inds``:
        optional uint32 [nq, max_dlas-1, S], 1-based (0 = never drawn); omitted, the resampling
        of :467-472 is drawn on the GPU."""
        nq, md, S = self.num_quasars, self.max_dlas, self.num_samples
        base_ptr = None
        if base_sample_inds is not None:
            base = np.ascontiguousarray(base_sample_inds, dtype=np.uint32)
            if base.shape != (nq, md - 1, S):
                raise _lib.GpdlaError(-1, f"base_sample_inds must be [nq, max_dlas-1, S], got {base.shape}")
            base_ptr = base.ctypes.data_as(C.POINTER(C.c_uint32))
        _lib.check(self.ctx.lib.gpdla_batch_process_multi(self.ctx._h, self._h, base_ptr))

    @staticmethod
    def empty_results_multi(nq: int, md: int, S: int, with_samples: bool = True) -> dict:
        """Host arrays for the variables the multi-DLA script saves (:498-510), for ``nq`` quasars."""
        out = {
            "min_z_dlas": np.full(nq, np.nan), "max_z_dlas": np.full(nq, np.nan),
            "log_likelihoods_no_dla": np.full(nq, np.nan),
            "log_likelihoods_dla": np.full((nq, md), np.nan), "log_likelihoods_lls": np.full(nq, np.nan),
            "log_posteriors_no_dla": np.full(nq, np.nan), "log_posteriors_lls": np.full(nq, np.nan),
            "log_posteriors_dla": np.full((nq, md), np.nan),
            "model_posteriors": np.full((nq, 2 + md), np.nan),
            "p_no_dlas": np.full(nq, np.nan), "p_lls": np.full(nq, np.nan), "p_dlas": np.full(nq, np.nan),
            "MAP_z_dlas": np.full((nq, md, md), np.nan), "MAP_log_nhis": np.full((nq, md, md), np.nan),
            "MAP_inds": np.full((nq, md, md), np.nan),
            "status": np.zeros(nq, dtype=np.int32),
            "log_priors_no_dla": np.full(nq, np.nan), "log_priors_lls": np.full(nq, np.nan),
            "log_priors_dla": np.full((nq, md), np.nan), "all_exceptions": np.full(nq, np.nan),
        }
        if with_samples:
            out["sample_log_likelihoods_dla"] = np.empty((nq, md, S))
            out["sample_log_likelihoods_lls"] = np.empty((nq, S))
            out["base_sample_inds"] = np.zeros((nq, md - 1, S), dtype=np.uint32)
        return out

    def download_multi(self, with_samples: bool = True, out: dict | None = None, at: int = 0) -> dict:
        """Results under the variable names the multi-DLA script saves (:498-510); 3-D arrays are
        ``sample_log_likelihoods_dla [nq, max_dlas, S]`` and ``MAP_* [nq, model, slot]``.
        ``out`` / ``at``: as in :meth:`download`."""
        nq, md, S = self.num_quasars, self.max_dlas, self.num_samples
        if out is None:
            out, at = self.empty_results_multi(nq, md, S, with_samples), 0
        r = _lib.ResultsMulti()
        for name, _ in _lib.ResultsMulti._fields_:
            if name in out and out[name].size:
                ct = {"status": C.c_int32, "base_sample_inds": C.c_uint32}.get(name, C.c_double)
                view = out[name][at:at + nq]
                assert view.flags.c_contiguous and view.shape[0] == nq
                setattr(r, name, view.ctypes.data_as(C.POINTER(ct)))
        _lib.check(self.ctx.lib.gpdla_batch_download_multi(self.ctx._h, self._h, C.byref(r)))
        out["log_priors_no_dla"][at:at + nq] = self.log_priors_no_dla
        out["log_priors_lls"][at:at + nq] = self.log_priors_lls
        out["log_priors_dla"][at:at + nq] = self.log_priors_dla
        out["all_exceptions"][at:at + nq] = np.where(out["status"][at:at + nq] == 1, 1.0, np.nan)  # multi :139, :232
        return out

    def samples_multi_tensors(self):
        """(sample_log_likelihoods_dla [nq, max_dlas, S], sample_log_likelihoods_lls [nq, S]) of a
        multi-DLA batch as zero-copy torch tensors on this GPU."""
        import torch
        a, b = C.c_void_p(), C.c_void_p()
        _lib.check(self.ctx.lib.gpdla_batch_samples_multi_device_ptr(self._h, C.byref(a), C.byref(b), None))
        nq, md, S = self.num_quasars, self.max_dlas, self.num_samples
        dev = f"cuda:{self.ctx.device}"
        return (torch.as_tensor(_DeviceArray(a.value, (nq, md, S), self), device=dev),
                torch.as_tensor(_DeviceArray(b.value, (nq, S), self), device=dev))

    def samples_tensor(self):
        """sample_log_likelihoods_dla [nq, S] as a zero-copy torch tensor on this GPU."""
        import torch
        p, n, s = C.c_void_p(), C.c_int64(), C.c_int64()
        _lib.check(self.ctx.lib.gpdla_batch_samples_device_ptr(self._h, C.byref(p), C.byref(n),
                                                               C.byref(s)))
        return torch.as_tensor(_DeviceArray(p.value, (n.value, s.value), self),
                               device=f"cuda:{self.ctx.device}")

    def close(self):
        if self._h:
            self.ctx.lib.gpdla_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------
# the script surface
# ----------------------------------------------------------------------------------------------

def record_bytes_per_quasar(num_pixels: int, k: int, slim: bool = True) -> int:
    """Bytes of K-step records (one per 4 pixels) a quasar of ``num_pixels`` stored pixels needs in
    the record pool: 896 B per step for k <= 20 (k_sweep_slim / k_sweep_multi_slim: the M rows, pixel
    rows and wavelengths), 1536 B for 20 < k <= 40 (k_sweep_split_slim: the M rows);
    ``slim=False``: the pre-expanded records of k_sweep / k_sweep_split -- 7680 B and 29 696 B --
    used by the fp32 study and libgpdla_legacy.so's ``GPDLA_EXPANDED_RECORDS=1`` only (every fp64 line count has slim records).  The
    pool of the single-DLA sweep is bounded by ``Parameters.record_pool_bytes`` whatever the batch
    size (the library sweeps group by group)."""
    per_step = ((896 if slim else 7680) if k <= 20 else (1536 if slim else 29696))
    return int((num_pixels / 4 + 2) * per_step)


def resident_bytes_per_quasar(num_pixels: int, k: int, num_samples: int, multi_models: int = 0) -> int:
    """HBM a quasar occupies in a resident batch apart from the record pool: its spectrum, the
    interpolated rows (k + 4 doubles per pixel), the padded wavelengths and its result tables."""
    rows = (num_pixels + 8) * (k + 4 + 1 + 3.2) * 8
    return int(rows + 8 * num_samples * max(1, 2 * multi_models))


def default_batch_size(num_quasars: int, longest: int, k: int, num_samples: int, slots: int,
                       budget_bytes: float = 96 * 2**30, multi_models: int = 0) -> int:
    """Quasars per batch of the host pipeline (gpdla_default_batch_quasars, the rule the one-shot C
    entries apply to themselves): small enough that ``slots`` batches fit ``budget_bytes`` of HBM next
    to the record pool and that a run has ~8 batches to overlap (the first upload and the last
    download are the only copies not hidden behind a sweep), at least 128 so that a launch fills the
    256 CUs many times over."""
    return int(_lib.load().gpdla_default_batch_quasars(int(num_quasars), int(longest), int(k), int(num_samples),
                                                       int(slots), int(budget_bytes), int(multi_models)))


def batch_blocks(num_quasars: int, per_batch: int) -> list:
    """[lo, hi) blocks of a pipelined run.  (Equal blocks: a short first block would start the GPU
    2 ms earlier, but the slot it leaves behind has to grow when it is re-filled, and a hipFree
    waits for the sweeps in flight.)"""
    per_batch = max(1, int(per_batch))
    return [(lo, min(lo + per_batch, num_quasars)) for lo in range(0, num_quasars, per_batch)]


def prefault(*arrays):
    """Touch every page of freshly allocated output arrays (one write per 4 KiB).  Done on the
    download thread while the first batch is swept: a device-to-host copy into untouched pageable
    memory runs at page-fault speed (4 GB/s measured; 10+ once the pages exist)."""
    for a in arrays:
        flat = a.reshape(-1)
        flat[::max(1, 4096 // a.itemsize)] = 0


def run_pipeline(ctx: "Context", num_blocks: int, inputs, process, download, slots: int = 3, warm=None):
    """The host loop of process_qsos.m:88 as a three-stage pipeline over HBM-resident batches:
    while batch i is swept, batch i+1 is prepared and uploaded by one thread and batch i-1
    downloaded by another (the library's copy streams run beside the compute stream; ctypes
    releases the GIL inside the calls).  ``inputs(i)`` returns the arguments of ``Context.upload``
    for block i; ``process(i, batch)`` launches its sweep (main thread, in order);
    ``download(i, batch)`` fetches its results.  ``slots`` batches exist at a time and are
    re-filled in place, so the steady state allocates nothing.  ``warm``: a callable run once on the
    download thread before the first download (e.g. :func:`prefault` of the output arrays)."""
    from concurrent.futures import ThreadPoolExecutor
    if num_blocks <= 0:
        return
    slots = max(1, min(slots, num_blocks))
    batches = [None] * slots
    done = [None] * num_blocks
    up_pool, down_pool = ThreadPoolExecutor(1), ThreadPoolExecutor(1)

    def upload(i):
        slot = i % slots
        if i >= slots:
            done[i - slots].result()  # the slot's previous results are on the host
        args = inputs(i)
        if batches[slot] is None:
            batches[slot] = ctx.upload(*args)
        else:
            batches[slot].reload(*args)
        return batches[slot]

    try:
        nxt = up_pool.submit(upload, 0)
        warmed = down_pool.submit(warm) if warm is not None else None
        for i in range(num_blocks):
            batch = nxt.result()
            # upload(i + 1) waits for done[i + 1 - slots]: with one slot that is THIS block's
            # download, which does not exist before the sweep is launched
            if slots > 1 and i + 1 < num_blocks:
                nxt = up_pool.submit(upload, i + 1)
            process(i, batch)
            done[i] = down_pool.submit(download, i, batch)
            if slots == 1 and i + 1 < num_blocks:
                nxt = up_pool.submit(upload, i + 1)
        for f in done:
            f.result()
        if warmed is not None:
            warmed.result()
    finally:
        up_pool.shutdown(wait=True)
        down_pool.shutdown(wait=True)
        for b in batches:
            if b is not None:
                b.close()


def _model_struct(model: dict, keep: list) -> "_lib.Model":
    rw, rwp = _f64(model["rest_wavelengths"])
    mu, mup = _f64(model["mu"])
    Mf = np.asfortranarray(model["M"], dtype=np.float64)
    lo, lop = _f64(model["log_omega"])
    keep += [rw, mu, Mf, lo]
    return _lib.Model(rw.size, Mf.shape[1], rwp, mup, Mf.ctypes.data_as(_dp), lop,
                      float(model["log_c_0"]), float(model["log_tau_0"]), float(model["log_beta"]))


def _samples_struct(samples: dict, keep: list) -> "_lib.Samples":
    off, offp = _f64(samples["offset_samples"])
    nhi, nhip = _f64(samples["nhi_samples"])
    keep += [off, nhi]
    lnp = llp = None
    if samples.get("log_nhi_samples") is not None:
        a, lnp = _f64(samples["log_nhi_samples"])
        keep.append(a)
    if samples.get("lls_nhi_samples") is not None:
        a, llp = _f64(samples["lls_nhi_samples"])
        keep.append(a)
    return _lib.Samples(off.size, offp, lnp, nhip, llp)


def _spectra_struct(csr: dict, lp_no, lp_dla, lp_lls, keep: list) -> "_lib.Spectra":
    def ptr(a, dt, ct):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data_as(C.POINTER(ct))
    return _lib.Spectra(
        csr["z_qsos"].size, ptr(csr["offsets"], np.int64, C.c_int64),
        ptr(csr["wavelengths"], np.float64, C.c_double), ptr(csr["flux"], np.float64, C.c_double),
        ptr(csr["noise_variance"], np.float64, C.c_double), ptr(csr["pixel_mask"], np.uint8, C.c_uint8),
        ptr(csr["z_qsos"], np.float64, C.c_double), ptr(lp_no, np.float64, C.c_double),
        ptr(lp_dla, np.float64, C.c_double), None if lp_lls is None else ptr(lp_lls, np.float64, C.c_double))


_addressof, _char_from_buffer = C.addressof, C.c_char.from_buffer


def _data_address(a: np.ndarray) -> int:
    """Address of an array's first byte.  Through the buffer protocol where that works (a writable,
    non-empty array: 0.3 us), else through ``__array_interface__`` (1 us: a dict is built per call) --
    a list of 2048 quasars is 8192 arrays."""
    try:
        return _addressof(_char_from_buffer(a))
    except (TypeError, ValueError, BufferError):
        return a.__array_interface__["data"][0]


def _cells_struct(spectra, lp_no, lp_dla, lp_lls, keep: list) -> "_lib.SpectraCells":
    """gpdla_spectra_cells over a list of per-quasar dicts: pointers to the arrays as they are (an
    array that is not contiguous float64 -- uint8 / bool for the mask -- is converted, that one only).
    Column by column, in comprehensions: the per-quasar Python work is what a 2048-quasar call pays
    before the library starts (5 us per quasar as one loop with ``__array_interface__``, 1.5 us so)."""
    n = len(spectra)
    f64, u8, b1, nd = np.dtype(np.float64), np.dtype(np.uint8), np.dtype(np.bool_), np.ndarray
    ptrs = np.empty((4, n), dtype=np.uintp)
    sizes = []
    for j, key in enumerate(("wavelengths", "flux", "noise_variance", "pixel_mask")):
        col = [s[key] for s in spectra]
        if j < 3:
            fix = [i for i, a in enumerate(col) if not (type(a) is nd and a.dtype == f64 and a.flags.c_contiguous)]
            for i in fix:
                col[i] = np.ascontiguousarray(col[i], dtype=np.float64)
        else:
            fix = [i for i, a in enumerate(col)
                   if not (type(a) is nd and (a.dtype == u8 or a.dtype == b1) and a.flags.c_contiguous)]
            for i in fix:
                col[i] = np.ascontiguousarray(col[i], dtype=np.uint8)
        keep.append(col)
        ptrs[j] = [_data_address(a) for a in col]
        sizes.append(np.array([a.size for a in col], dtype=np.int64))
    npix = sizes[0]
    same = (sizes[1] == npix) & (sizes[2] == npix) & (sizes[3] == npix)
    if not same.all():
        raise _lib.GpdlaError(-1, f"quasar {int(np.flatnonzero(~same)[0])}: wavelengths, flux, noise_variance and pixel_mask differ in length")
    z = np.array([s["z_qso"] for s in spectra], dtype=np.float64)
    keep += [ptrs, npix, z]

    def dptr(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        keep.append(a)
        return a.ctypes.data_as(_dp)
    return _lib.SpectraCells(n, npix.ctypes.data_as(C.POINTER(C.c_int64)), ptrs[0].ctypes.data, ptrs[1].ctypes.data,
                             ptrs[2].ctypes.data, ptrs[3].ctypes.data, z.ctypes.data_as(_dp), dptr(lp_no), dptr(lp_dla),
                             None if lp_lls is None else dptr(lp_lls))


def _result_struct(struct_type, out: dict):
    r = struct_type()
    for name, _ in struct_type._fields_:
        if name in out and out[name].size:
            ct = {"status": C.c_int32, "base_sample_inds": C.c_uint32}.get(name, C.c_double)
            assert out[name].flags.c_contiguous
            setattr(r, name, out[name].ctypes.data_as(C.POINTER(ct)))
    return r


def process_qsos(model: dict, samples: dict, spectra, prior_catalog: dict | None = None,
                 params: Parameters | None = None, device: int = 0,
                 log_priors: tuple | None = None, max_quasars_per_batch: int | None = None,
                 pipeline_slots: int = 3, with_samples: bool = True) -> dict:
    """The ``process_qsos`` script (process_qsos.m:4-250) for a list of quasars: a thin caller of
    ``gpdla_process_cells`` (a list: one array per quasar, handed over as it is) or
    ``gpdla_process_batch`` (CSR arrays), the one-shot C entries a MEX gateway binds (INTEGRATION.md
    section 3).

    ``spectra``: list of dicts with ``wavelengths, flux, noise_variance, pixel_mask, z_qso`` (one
    entry of the ``all_*`` cell arrays each, after the ``test_ind`` subset of :56-61), or the CSR
    dict of :func:`spectra_to_csr` (no host copy then).
    ``prior_catalog``: ``{"z_qsos", "dla_ind"}`` of the training release after the Lyman-limit
    filter of :15-25; or pass ``log_priors=(log_priors_no_dla, log_priors_dla)`` directly.
    Quasars are independent, so the library sweeps the list in HBM-resident batches of at most
    ``max_quasars_per_batch`` (default: :func:`default_batch_size`) through ``pipeline_slots`` batch
    slots: uploads and downloads overlap the sweeps.  Results do not depend on the batching.
    Returns the variables the script saves (:236-244)."""
    p = params or Parameters()
    is_csr = isinstance(spectra, dict)
    if not is_csr:
        spectra = list(spectra)
    z_all = spectra["z_qsos"] if is_csr else np.array([float(s_["z_qso"]) for s_ in spectra], dtype=np.float64)
    nq = z_all.size
    if log_priors is None:
        if prior_catalog is None:
            raise ValueError("need prior_catalog or log_priors")
        log_priors = dla_existence_prior(prior_catalog["z_qsos"], prior_catalog["dla_ind"], z_all, p)
    lp_no, lp_dla = (np.ascontiguousarray(x, dtype=np.float64) for x in log_priors)
    S = np.asarray(samples["offset_samples"]).size
    out = Batch.empty_results(nq, S, with_samples) if nq else {}
    if nq:
        lib = _lib.load()
        cfg = _config(p)
        cfg.pipeline_slots = int(pipeline_slots)
        cfg.max_quasars_per_batch = int(max_quasars_per_batch or 0)
        keep = []
        m, s_ = _model_struct(model, keep), _samples_struct(samples, keep)
        r = _result_struct(_lib.Results, out)
        if is_csr:
            sp = _spectra_struct(spectra, lp_no, lp_dla, None, keep)
            rc = lib.gpdla_process_batch(C.byref(m), C.byref(s_), C.byref(sp), C.byref(cfg), C.byref(r), int(device))
        else:  # one array per quasar, as they are: the library flattens block by block beside the sweeps
            sp = _cells_struct(spectra, lp_no, lp_dla, None, keep)
            rc = lib.gpdla_process_cells(C.byref(m), C.byref(s_), C.byref(sp), C.byref(cfg), C.byref(r), int(device))
        _lib.check(rc)
        out["log_priors_no_dla"][:] = lp_no
        out["log_priors_dla"][:] = lp_dla
    out["num_lines"] = p.num_lines
    out["prior_z_qso_increase"] = p.prior_z_qso_increase
    out["max_z_cut"] = p.max_z_cut
    return out


# ----------------------------------------------------------------------------------------------
# multi-DLA driver
# ----------------------------------------------------------------------------------------------

def dla_existence_prior_multi(prior_z_qsos, prior_dla_ind, z_qsos, Z_lls: float, Z_dla: float,
                              params: MultiParameters | None = None):
    """process_qsos_multiple_dlas_meanflux.m:189-216: priors for exactly 1..max_dlas DLAs
    ((M/N)^k - (M/N)^(k+1)), for a sub-DLA (M/N * Z_lls/Z_dla) and for no absorber.
    Returns (log_priors_no_dla [nq], log_priors_lls [nq], log_priors_dla [nq, max_dlas])."""
    p = params or MultiParameters()
    pz = np.asarray(prior_z_qsos, dtype=np.float64)
    pd = np.asarray(prior_dla_ind, dtype=bool)
    z = np.atleast_1d(np.asarray(z_qsos, dtype=np.float64))
    order = np.argsort(pz, kind="stable")
    cum = np.concatenate([[0], np.cumsum(pd[order])])
    N = np.searchsorted(pz[order], z + p.prior_z_qso_increase, side="left").astype(np.float64)
    M = cum[N.astype(np.int64)].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        pk = (M / N)[:, None] ** np.arange(1, p.max_dlas + 1)[None, :]      # :194
        pk[:, :-1] = pk[:, :-1] - pk[:, 1:]                                  # :197-199
        log_dla = np.log(pk)                                                 # :204
        log_lls = np.log(M) - np.log(N) + np.log(Z_lls) - np.log(Z_dla)      # :208-210
        log_no = np.log(N - M - Z_lls * M / Z_dla) - np.log(N)               # :214-216
    return log_no, log_lls, log_dla


def process_qsos_multiple_dlas_meanflux(model: dict, samples: dict, spectra, log_priors,
                                        params: MultiParameters | None = None,
                                        base_sample_inds=None, device: int = 0,
                                        max_quasars_per_batch: int | None = None,
                                        pipeline_slots: int = 3) -> dict:
    """The multi-DLA driver (multi_dlas/process_qsos_multiple_dlas_meanflux.m:141-510).

    ``samples`` additionally carries ``log_nhi_samples`` and ``lls_nhi_samples``
    (set_lls_parameters.m:59-63).  ``log_priors = (no_dla [nq], lls [nq], dla [nq, max_dlas])`` as
    returned by :func:`dla_existence_prior_multi`.  ``base_sample_inds``: optional uint32
    ``[nq, max_dlas-1, S]``, 1-based (the reference's saved variable, :476, transposed to
    quasar-slowest; rows the reference left zero after its early exit make the samples that
    would consume them NaN); when omitted the resampling of :467-472 is drawn on the GPU
    (Philox4x32-10, ``params.rng_seed``, keyed by ``params.first_quasar_index`` + position, so
    batches and shards of one run draw what the whole run would).  Quasars are independent, so a
    long list is swept in HBM-resident batches of at most ``max_quasars_per_batch``.
    Returns the variables the script saves (:498-510); 3-D arrays are
    ``sample_log_likelihoods_dla [nq, max_dlas, S]`` and ``MAP_* [nq, model, slot]``."""
    p = params or MultiParameters()
    md = p.max_dlas
    S = np.asarray(samples["offset_samples"]).size
    is_csr = isinstance(spectra, dict)
    if not is_csr:
        spectra = list(spectra)
    nq = spectra["z_qsos"].size if is_csr else len(spectra)
    lp_no, lp_lls, lp_dla = (np.ascontiguousarray(x, dtype=np.float64) for x in log_priors)
    lp_dla = lp_dla.reshape(nq, md)
    base_ptr = None
    if base_sample_inds is not None:
        base_sample_inds = np.ascontiguousarray(base_sample_inds, dtype=np.uint32)
        if base_sample_inds.shape != (nq, md - 1, S):
            raise _lib.GpdlaError(-1, "base_sample_inds must be [nq, max_dlas-1, S], got "
                                  f"{base_sample_inds.shape}")
        base_ptr = base_sample_inds.ctypes.data_as(C.POINTER(C.c_uint32))
    out = Batch.empty_results_multi(nq, md, S) if nq else {}
    if nq:
        lib = _lib.load()
        cfg = _config(p)
        cfg.pipeline_slots = int(pipeline_slots)
        cfg.max_quasars_per_batch = int(max_quasars_per_batch or 0)
        keep = []
        m, s_ = _model_struct(model, keep), _samples_struct(samples, keep)
        r = _result_struct(_lib.ResultsMulti, out)
        if is_csr:
            sp = _spectra_struct(spectra, lp_no, lp_dla, lp_lls, keep)
            rc = lib.gpdla_process_batch_multi(C.byref(m), C.byref(s_), C.byref(sp), base_ptr, C.byref(cfg), C.byref(r),
                                               int(device))
        else:
            sp = _cells_struct(spectra, lp_no, lp_dla, lp_lls, keep)
            rc = lib.gpdla_process_cells_multi(C.byref(m), C.byref(s_), C.byref(sp), base_ptr, C.byref(cfg), C.byref(r),
                                               int(device))
        _lib.check(rc)
        out["log_priors_no_dla"][:] = lp_no
        out["log_priors_lls"][:] = lp_lls
        out["log_priors_dla"][:] = lp_dla
        out["all_exceptions"][:] = np.where(out["status"] == 1, 1.0, np.nan)  # multi :139, :232
    return out
