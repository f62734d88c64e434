"""Synthetic GP model, DLA samples and quasar spectra of the shapes BASELINE.json names.

There is no SDSS data in the build container or on the GPU box, so tests and ``bench.py`` use the
generator specified in SURVEY.md section 8(d): a smooth mean, k orthonormal cosine modes with
geometric scales, Halton (z-offset, log N_HI) samples, and spectra drawn from the model with a DLA
injected into half of them.  Everything is seeded, so the oracle and the HIP path see identical
inputs.  The shapes mirror the reference's files: the model fields of learn_qso_model.m:113-123,
the samples of generate_dla_samples.m:59-63 and the ragged per-quasar arrays of
preload_qsos.m:64-79.
"""
from __future__ import annotations

import numpy as np

from .parameters import Parameters

MODEL_SEED = 12345
SPECTRUM_SEED0 = 20260101


def halton(n: int, base: int) -> np.ndarray:
    """Plain (unscrambled) Halton sequence, points 1..n."""
    out = np.zeros(n)
    for i in range(n):
        f, r, j = 1.0, 0.0, i + 1
        while j > 0:
            f /= base
            r += f * (j % base)
            j //= base
        out[i] = r
    return out


def make_model(k: int = 20, params: Parameters | None = None) -> dict:
    p = params or Parameters()
    rng = np.random.default_rng(MODEL_SEED)
    G = int(round((p.max_lambda - p.min_lambda) / p.dlambda)) + 1  # 1217, learn_qso_model.m:29
    rest = p.min_lambda + p.dlambda * np.arange(G)
    u = (rest - rest[0]) / (rest[-1] - rest[0])
    mu = (1.0 + 0.3 * np.exp(-0.5 * ((u - 0.95) / 0.08) ** 2)
          + 0.1 * np.exp(-0.5 * ((u - 0.38) / 0.03) ** 2))
    modes = np.cos(np.pi * (np.arange(G)[:, None] + 0.5) * (np.arange(k)[None, :] + 1) / G)
    modes *= np.sqrt(2.0 / G)  # orthonormal columns
    M = modes * (0.3 * 0.8 ** np.arange(k))[None, :] * np.sqrt(G) * 0.25
    log_omega = rng.uniform(-3.0, -2.0, size=G)
    return dict(rest_wavelengths=rest, mu=mu, M=np.asfortranarray(M), log_omega=log_omega,
                log_c_0=float(np.log(0.1)), log_tau_0=float(np.log(0.0023)),
                log_beta=float(np.log(3.65)))


def make_samples(num_samples: int = 10000) -> dict:
    offset = halton(num_samples, 2)
    log_nhi = 20.0 + 3.0 * halton(num_samples, 3)
    lls_log_nhi = 19.5 + 0.5 * halton(num_samples, 5)  # set_lls_parameters.m:59-63 range
    return dict(offset_samples=offset, log_nhi_samples=log_nhi, nhi_samples=10.0 ** log_nhi,
                lls_log_nhi_samples=lls_log_nhi, lls_nhi_samples=10.0 ** lls_log_nhi)


def _injected_absorption(wavelengths, z_dla, nhi, num_lines):
    # raw (un-broadened) Lyman-series profile, for data synthesis only
    from scipy.special import wofz

    from ._lyman import C_CGS, LINES, SIGMA_CGS
    total = np.zeros_like(wavelengths)
    for j in range(num_lines):
        wl, lead, gam = LINES[j][0], LINES[j][3], LINES[j][4]
        v = wavelengths * (C_CGS / (wl * (1 + z_dla)) / 1e8) - C_CGS
        zc = (v + 1j * gam) / (np.sqrt(2) * SIGMA_CGS)
        total += -lead * np.real(wofz(zc)) / (np.sqrt(2 * np.pi) * SIGMA_CGS)
    return np.exp(nhi * total)


def make_spectrum(index: int, n: int, model: dict, params: Parameters | None = None,
                  mask_fraction: float = 0.0, edge_pixels: int = 2) -> dict:
    """One synthetic quasar with exactly ``n`` pixels inside the modelled rest range (plus
    ``edge_pixels`` outside on each side so the range test of process_qsos.m:104-105 is
    exercised)."""
    p = params or Parameters()
    rng = np.random.default_rng(SPECTRUM_SEED0 + index)
    z_qso = float(rng.uniform(2.2, 4.5))
    span = np.log10(p.max_lambda / p.min_lambda)
    dlog = span / n
    idx = np.arange(-edge_pixels, n + edge_pixels)
    log_wl = np.log10(p.min_lambda * (1 + z_qso)) + (idx + 0.5) * dlog
    wl = 10.0 ** log_wl
    rest = wl / (1 + z_qso)
    k = model["M"].shape[1]
    grid = model["rest_wavelengths"]
    mu = np.interp(rest, grid, model["mu"])
    Mi = np.stack([np.interp(rest, grid, model["M"][:, c]) for c in range(k)], 1)
    omega2 = np.exp(2 * np.interp(rest, grid, model["log_omega"]))
    nv = 10.0 ** rng.uniform(-3.0, -1.0, size=wl.size)
    flux = (mu + Mi @ rng.standard_normal(k)
            + np.sqrt(omega2 * 0.04 + nv) * rng.standard_normal(wl.size))
    has_dla = bool(index % 2)
    z_dla = log_nhi = None
    if has_dla:
        zmin = p.min_z_dla(wl, z_qso)
        zmax = p.max_z_dla(wl, z_qso)
        z_dla = float(rng.uniform(zmin, zmax))
        log_nhi = float(rng.uniform(20.0, 22.0))
        flux = flux * _injected_absorption(wl, z_dla, 10.0 ** log_nhi, p.num_lines)
    mask = np.zeros(wl.size, dtype=np.uint8)
    if mask_fraction > 0:
        mask = (rng.uniform(size=wl.size) < mask_fraction).astype(np.uint8)
        # masked pixels look like preload_qsos.m's: zero inverse variance -> infinite variance
        nv = np.where(mask == 1, np.inf, nv)
        flux = np.where(mask == 1, np.nan, flux)
    return dict(wavelengths=wl, flux=flux, noise_variance=nv, pixel_mask=mask, z_qso=z_qso,
                true_z_dla=z_dla, true_log_nhi=log_nhi)


BOSS_LOGLAM0, BOSS_NPIX = 3.5563, 4608   # BOSS spectrograph grid: 3600 .. 10400 A at 1e-4 dex


def sample_dr12q_redshifts(num: int, seed: int = 4321) -> np.ndarray:
    """Quasar redshifts shaped like the DR12Q Lyman-alpha-forest sample the reference searches
    (z_qso >= 2.15, build_catalogs.m; strongly peaked at 2.2 - 2.6 with a tail past 4)."""
    rng = np.random.default_rng(seed)
    z = 2.15 + rng.gamma(shape=1.6, scale=0.38, size=num)
    return np.where(z > 5.8, 2.15 + (z - 2.15) % 3.6, z)


def make_boss_spectrum(index: int, z_qso: float, model: dict, params: Parameters | None = None,
                       mask_fraction: float = 0.05, edge_pixels: int = 2, mask_runs: bool = False) -> dict:
    """One synthetic quasar on the BOSS pixel grid (log10 lambda = 3.5563 + 1e-4 j): the modelled
    rest range [911.75, 1215.75] A covers at most log10(1215.75 / 911.75) / 1e-4 = 1250 pixels and
    fewer when the spectrograph's blue edge (3600 A) cuts it (z_qso < 2.95) -- the real length mix
    of a DR12Q run (preload_qsos.m:46 keeps quasars with >= 200 pixels)."""
    p = params or Parameters()
    rng = np.random.default_rng(SPECTRUM_SEED0 + 7919 * 1000 + index)
    loglam = BOSS_LOGLAM0 + p.pixel_spacing * np.arange(BOSS_NPIX)
    rest_all = 10.0 ** loglam / (1 + z_qso)
    inside = np.flatnonzero((rest_all >= p.min_lambda) & (rest_all <= p.max_lambda))
    lo, hi = max(inside[0] - edge_pixels, 0), min(inside[-1] + edge_pixels + 1, BOSS_NPIX)
    wl = 10.0 ** loglam[lo:hi]
    rest = wl / (1 + z_qso)
    k = model["M"].shape[1]
    grid = model["rest_wavelengths"]
    mu = np.interp(rest, grid, model["mu"])
    Mi = np.stack([np.interp(rest, grid, model["M"][:, c]) for c in range(k)], 1)
    omega2 = np.exp(2 * np.interp(rest, grid, model["log_omega"]))
    nv = 10.0 ** rng.uniform(-3.0, -1.0, size=wl.size)
    flux = (mu + Mi @ rng.standard_normal(k)
            + np.sqrt(omega2 * 0.04 + nv) * rng.standard_normal(wl.size))
    z_dla = log_nhi = None
    if index % 2:
        zmin, zmax = p.min_z_dla(wl, z_qso), p.max_z_dla(wl, z_qso)
        if zmax > zmin:
            z_dla = float(rng.uniform(zmin, zmax))
            log_nhi = float(rng.uniform(20.0, 22.0))
            flux = flux * _injected_absorption(wl, z_dla, 10.0 ** log_nhi, p.num_lines)
    mask = (rng.uniform(size=wl.size) < mask_fraction).astype(np.uint8)
    if mask_runs:
        # the same masked fraction in contiguous runs of 4..12 pixels, as sky-line residuals and bad
        # columns mask a real spectrum (preload_qsos.m: BRIGHTSKY | zero inverse variance), instead of
        # independent pixels; a separate generator, so that everything else of the spectrum is unchanged
        r2 = np.random.default_rng(SPECTRUM_SEED0 + 104729 * 1000 + index)
        mask = np.zeros(wl.size, dtype=np.uint8)
        starts = np.flatnonzero(r2.uniform(size=wl.size) < mask_fraction / 8.0)
        for s0, ln in zip(starts, r2.integers(4, 13, size=starts.size)):
            mask[s0:s0 + ln] = 1
    nv = np.where(mask == 1, np.inf, nv)      # preload_qsos.m: zero inverse variance
    flux = np.where(mask == 1, np.nan, flux)
    return dict(wavelengths=wl, flux=flux, noise_variance=nv, pixel_mask=mask, z_qso=float(z_qso),
                true_z_dla=z_dla, true_log_nhi=log_nhi)


def boss_pixel_counts(z_qsos, params: Parameters | None = None, edge_pixels: int = 2) -> np.ndarray:
    """Stored pixels of :func:`make_boss_spectrum` per redshift, without making any spectrum (what
    a sharded run balances its blocks by: ``PreloadedReader.pixel_counts`` reads the same number
    from the dataset headers of a real file)."""
    p = params or Parameters()
    z = np.atleast_1d(np.asarray(z_qsos, dtype=np.float64))
    loglam = BOSS_LOGLAM0 + p.pixel_spacing * np.arange(BOSS_NPIX)
    lam = 10.0 ** loglam
    first = np.searchsorted(lam, p.min_lambda * (1 + z), side="left")
    last = np.searchsorted(lam, p.max_lambda * (1 + z), side="right") - 1
    lo = np.maximum(first - edge_pixels, 0)
    hi = np.minimum(last + edge_pixels + 1, BOSS_NPIX)
    return np.maximum(hi - lo, 0).astype(np.int64)


def _dr12q_mix_block(args):
    lo, hi, k, mask_fraction, seed = args
    return make_dr12q_mix(hi - lo, make_model(k), None, mask_fraction, first_index=lo, seed=seed)


def make_dr12q_mix_parallel(first_index: int, num: int, k: int, workers: int, mask_fraction: float = 0.05,
                            seed: int = 4321) -> list:
    """:func:`make_dr12q_mix` for the default model of rank ``k`` over ``workers`` processes (0.6 ms per
    spectrum on one core: a DR12Q-sized run makes 162 861 of them).  The workers come from a fork
    server, so this may be called from a process that will use -- but has not yet touched -- the GPU."""
    if workers <= 1 or num < 4096:
        return make_dr12q_mix(num, make_model(k), None, mask_fraction, first_index=first_index, seed=seed)
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    step = max(512, -(-num // (4 * workers)))
    jobs = [(lo, min(lo + step, first_index + num), k, mask_fraction, seed)
            for lo in range(first_index, first_index + num, step)]
    with ProcessPoolExecutor(workers, mp_context=mp.get_context("forkserver")) as pool:
        parts = list(pool.map(_dr12q_mix_block, jobs))
    return [s for part in parts for s in part]


def make_dr12q_mix(num: int, model: dict, params: Parameters | None = None,
                   mask_fraction: float = 0.05, first_index: int = 0, seed: int = 4321,
                   mask_runs: bool = False) -> list:
    """``num`` DISTINCT quasars with the DR12Q length mix: redshifts from
    :func:`sample_dr12q_redshifts`, spectra on the BOSS grid, 5 % of the pixels masked
    (independently, or -- ``mask_runs`` -- in contiguous runs of 4..12 pixels)."""
    z = sample_dr12q_redshifts(first_index + num, seed)[first_index:]
    return [make_boss_spectrum(first_index + i, float(z[i]), model, params, mask_fraction, mask_runs=mask_runs)
            for i in range(num)]


def kept_pixel_counts(spectra, params: Parameters | None = None) -> np.ndarray:
    """n of process_qsos.m:110-115 for each quasar (in the rest range and not masked)."""
    p = params or Parameters()
    out = np.zeros(len(spectra), dtype=np.int64)
    for i, s in enumerate(spectra):
        rest = np.asarray(s["wavelengths"]) / (1 + s["z_qso"])
        out[i] = np.count_nonzero((rest >= p.min_lambda) & (rest <= p.max_lambda)
                                  & (np.asarray(s["pixel_mask"]) == 0))
    return out


def make_spectra(num: int, n: int, model: dict, params: Parameters | None = None,
                 mask_fraction: float = 0.0, first_index: int = 0) -> list:
    return [make_spectrum(first_index + i, n, model, params, mask_fraction) for i in range(num)]


def make_prior_catalog(num: int = 5000, seed: int = 777) -> dict:
    """A stand-in for the training-catalog fields process_qsos.m:11-27 reads."""
    rng = np.random.default_rng(seed)
    z_qsos = rng.uniform(2.15, 5.0, size=num)
    dla_ind = rng.uniform(size=num) < 0.1
    return dict(z_qsos=z_qsos, dla_ind=dla_ind)


def write_file_set(directory: str, num_quasars: int = 24, num_samples: int = 256, k: int = 20,
                   skip_every: int = 5, first_index: int = 4000, empty_quasar: int | None = 7) -> dict:
    """A small, seeded stand-in for the files either side of the path, with MATLAB's conventions
    (``-v7.3``, column vectors, cell arrays, logical masks): ``catalog.mat`` (build_catalogs.m:86-91:
    the plain per-quasar columns), ``preloaded_qsos.mat`` (preload_qsos.m:64-79),
    ``learned_qso_model_synthetic.mat`` (learn_qso_model.m:113-123), ``dla_samples.mat``
    (generate_dla_samples.m:59-63 + set_lls_parameters.m:59-63), ``snrs_qsos.mat`` (one entry per
    searched quasar), the training
    release's ``prior_catalog.npz`` and the two text catalogues ``QSOLoader`` opens
    (qso_loader.py:410-424: ``dla_catalog`` = thing_id, z_dla, log_nhi; ``los_catalog`` = thing_id).
    Every ``skip_every``-th quasar has ``filter_flags != 0`` (outside ``test_ind``); quasar
    ``empty_quasar`` is fully masked (the sweep skips it: NaN results).  Returns the paths and
    the arrays behind them."""
    import os

    from . import io
    os.makedirs(directory, exist_ok=True)
    d = str(directory)
    model = make_model(k)
    samples = make_samples(num_samples)
    spectra = make_dr12q_mix(num_quasars, model, first_index=first_index)
    if empty_quasar is not None and empty_quasar < num_quasars:
        s = spectra[empty_quasar]
        s["pixel_mask"] = np.ones_like(s["pixel_mask"])
        s["flux"] = np.full_like(s["flux"], np.nan)
        s["noise_variance"] = np.full_like(s["noise_variance"], np.inf)
    rng = np.random.default_rng(99)
    col = lambda a, dt=np.float64: np.asarray(a, dtype=dt).reshape(-1, 1)
    filter_flags = np.array([(i % skip_every == skip_every - 1) * 2 for i in range(num_quasars)], dtype=np.uint8)
    cat = dict(z_qsos=np.array([s["z_qso"] for s in spectra]), thing_ids=100000 + 7 * np.arange(num_quasars),
               plates=3586 + np.arange(num_quasars) // 4, mjds=55181 + np.arange(num_quasars) % 3,
               fiber_ids=1 + np.arange(num_quasars), snrs=rng.uniform(0.5, 12.0, num_quasars),
               ras=rng.uniform(0, 360, num_quasars), decs=rng.uniform(-10, 60, num_quasars),
               filter_flags=filter_flags)
    paths = dict(catalog=f"{d}/catalog.mat", preloaded=f"{d}/preloaded_qsos.mat",
                 learned=f"{d}/learned_qso_model_synthetic.mat", samples=f"{d}/dla_samples.mat",
                 snrs=f"{d}/snrs_qsos.mat", prior=f"{d}/prior_catalog.npz",
                 dla_catalog=f"{d}/dla_catalog", los_catalog=f"{d}/los_catalog")
    io.savemat73(paths["catalog"], {k_: col(v, np.uint8 if k_ == "filter_flags" else np.float64)
                                    for k_, v in cat.items()})
    cells = {}
    for key, src in (("all_wavelengths", "wavelengths"), ("all_flux", "flux"),
                     ("all_noise_variance", "noise_variance"), ("all_pixel_mask", "pixel_mask")):
        cells[key] = [np.asarray(s[src]).astype(bool if src == "pixel_mask" else np.float64).reshape(-1, 1)
                      for s in spectra]
    io.savemat73(paths["preloaded"], cells, compress=True)
    # learn_qso_model.m: rest_wavelengths, mu and log_omega are ROW vectors (read back as
    # f['mu'][:, 0], qso_loader.py:211-217), M is [G x k]
    io.savemat73(paths["learned"], {k_: (np.asarray(v).reshape(1, -1) if np.ndim(v) == 1 else v)
                                    for k_, v in model.items()}, compress=True)
    io.savemat73(paths["samples"], {k_: v.reshape(1, -1) for k_, v in samples.items()})
    test_ind = filter_flags == 0
    # snrs_qsos_*.mat is written for the SEARCHED quasars (calc_cddf.compute_all_snrs runs over the
    # test set; qso_loader.py:111, :169 and calc_cddf.py:131 index it without test_ind)
    io.savemat73(paths["snrs"], dict(snrs=col(cat["snrs"][test_ind])))
    prior = make_prior_catalog()
    np.savez(paths["prior"], **prior)
    # a "concordance" catalogue naming the injected absorbers of some searched quasars
    with open(paths["dla_catalog"], "w") as f:
        for i, s in enumerate(spectra):
            if test_ind[i] and s["true_z_dla"] is not None and i % 3 == 1:
                f.write("%d %.6f %.4f\n" % (cat["thing_ids"][i], s["true_z_dla"], s["true_log_nhi"]))
    with open(paths["los_catalog"], "w") as f:
        for i in np.flatnonzero(test_ind):
            f.write("%d\n" % cat["thing_ids"][i])
    return dict(paths=paths, model=model, samples=samples, spectra=spectra, catalog=cat, prior=prior,
                test_ind=test_ind, Z_lls=0.31, Z_dla=0.69)
