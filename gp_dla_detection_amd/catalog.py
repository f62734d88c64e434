"""MAP extraction and the ASCII catalogue of generate_ascii_catalog.m.

Host-side formatting of results the sweep already produced; no compute.
"""
from __future__ import annotations

import re

import numpy as np


def map_estimates_host(results: dict, samples: dict):
    """MAP (z_DLA, log10 N_HI, 0-based index) per quasar recomputed on the host from the full
    sample table: generate_ascii_catalog.m:73-80 -- the sample with the largest log-likelihood
    (``nanmax``; first index on ties), mapped to a redshift with the quasar's own search range
    (process_qsos.m:162-164).  Needs ``sample_log_likelihoods_dla`` (80 KB per quasar)."""
    sll = np.asarray(results["sample_log_likelihoods_dla"])
    nq = sll.shape[0]
    map_ind = np.zeros(nq, dtype=np.int64)
    ok = ~np.all(np.isnan(sll), axis=1)
    map_ind[ok] = np.nanargmax(sll[ok], axis=1)  # all-NaN rows: MATLAB's nanmax returns index 1
    off = np.asarray(samples["offset_samples"])[map_ind]
    z = results["min_z_dlas"] + (results["max_z_dlas"] - results["min_z_dlas"]) * off
    return z, np.asarray(samples["log_nhi_samples"])[map_ind], map_ind


def map_estimates(results: dict, samples: dict | None = None):
    """MAP (z_DLA, log10 N_HI, 0-based index) per quasar.  The evidence kernel already finds them
    while it walks the sample table on the GPU (summary columns MAP_inds / MAP_z_dlas /
    MAP_log_nhis), so results that carry those columns need no 80 KB-per-quasar table on the host;
    older result dicts fall back to :func:`map_estimates_host`."""
    if "MAP_z_dlas" in results and np.ndim(results["MAP_z_dlas"]) == 1:
        ind = np.asarray(results["MAP_inds"], dtype=np.float64)
        return (np.asarray(results["MAP_z_dlas"]), np.asarray(results["MAP_log_nhis"]),
                np.where(np.isnan(ind), 0, ind - 1).astype(np.int64))
    return map_estimates_host(results, samples)


def _e3(x: float) -> str:
    """sprintf('%0.5e') with the exponent widened to three digits (generate_ascii_catalog.m:68-71)."""
    return re.sub(r"e([+-])(\d\d)$", r"e\g<1>0\2", "%0.5e" % x)


def write_dla_samples(path: str, samples: dict) -> None:
    """<test_set_name>_dla_samples.dat, generate_ascii_catalog.m:9-20."""
    with open(path, "w") as f:
        for o, n in zip(samples["offset_samples"], samples["log_nhi_samples"]):
            f.write("%06f %09f\n" % (o, n))


def write_results(path: str, thing_ids, results: dict, samples: dict) -> None:
    """<test_set_name>_results.dat, generate_ascii_catalog.m:52-83 (one line per searched quasar).
    The reference's first fprintf names two fields but passes one argument, so MATLAB emits the
    thing_id and stops at the unmatched %-18s (:60); that is reproduced."""
    z_map, n_map, _ = map_estimates(results, samples)
    mp = np.asarray(results["model_posteriors"])
    with open(path, "w") as f:
        for i, tid in enumerate(thing_ids):
            f.write("%09i " % int(tid))
            f.write("%06.4f %06.4f %8.5f %8.5f %12.5e %12.5e %s %s " % (
                results["min_z_dlas"][i], results["max_z_dlas"][i],
                results["log_priors_no_dla"][i], results["log_priors_dla"][i],
                results["log_likelihoods_no_dla"][i], results["log_likelihoods_dla"][i],
                _e3(mp[i, 0]), _e3(mp[i, 1])))
            f.write("%06.4f %07.4f\n" % (z_map[i], n_map[i]))


# ----------------------------------------------------------------------------------------------
# JSON catalogues of the multi-DLA run (CDDF_analysis/qso_loader.py:1927-2094)
# ----------------------------------------------------------------------------------------------

_INFO_FIELDS = (("ra", "ras"), ("snr", "snrs"), ("dec", "decs"), ("plate", "plates"), ("mjd", "mjds"),
                ("fiber_id", "fiber_ids"), ("thing_id", "thing_ids"), ("z_qso", "z_qsos"))


def _py(x):
    """numpy scalar -> Python scalar (json cannot serialise numpy types; the reference calls .item())."""
    return x.item() if hasattr(x, "item") else x


def occams_model_posteriors(model_posteriors, occams_razor: float = 10000.0) -> np.ndarray:
    """``QSOLoader._occams_model_posteriors`` (qso_loader.py:235-257): every model but the null one
    is penalised by ``1 / occams_razor`` and the row renormalised.  Returns a new array."""
    mp = np.array(model_posteriors, dtype=np.float64)
    mp[:, 1:] = mp[:, 1:] / occams_razor                              # :247
    return mp / np.sum(mp, axis=1)[:, None]                           # :250-252


def loader_view(results: dict, quasar_info: dict, sub_dla: bool = True, occams_razor: float = 10000.0,
                drop_nan: bool = True):
    """What ``QSOLoader.__init__`` holds after loading a processed file (qso_loader.py:97-170), i.e.
    what its catalogue methods then work on: the Occam-penalised ``model_posteriors`` (:136), ``p_dlas``
    / ``p_no_dlas`` recomputed from them (:137-138: with ``sub_dla`` the sub-DLA model counts as "no
    DLA"), and -- ``drop_nan`` -- the quasars whose posterior row is NaN removed from every per-quasar
    array (:147-170; these are the quasars the sweep skipped).  ``occams_razor=1`` leaves the
    posteriors as saved.  Returns (results view, quasar_info view, kept row indices)."""
    mp = occams_model_posteriors(results["model_posteriors"], occams_razor)
    nq = mp.shape[0]
    first_dla = 1 + int(bool(sub_dla))
    view = dict(model_posteriors=mp, p_dlas=mp[:, first_dla:].sum(axis=1),       # :137
                p_no_dlas=mp[:, :first_dla].sum(axis=1))                         # :138
    for key in ("min_z_dlas", "max_z_dlas", "MAP_z_dlas", "MAP_log_nhis"):
        if key in results:
            view[key] = np.asarray(results[key])
    for _, col in _INFO_FIELDS:
        if len(quasar_info[col]) != nq:
            raise ValueError(f"quasar_info[{col!r}] has {len(quasar_info[col])} entries for {nq} quasars")
    info = {col: np.asarray(quasar_info[col]) for _, col in _INFO_FIELDS}
    keep = np.arange(nq)
    if drop_nan:
        with np.errstate(invalid="ignore"):
            top = mp[np.arange(nq), np.argmax(mp, axis=1)]                       # :143-144 (NaN wins argmax)
        keep = np.flatnonzero(~np.isnan(top))                                    # :147
        view = {k: v[keep] for k, v in view.items()}
        info = {k: v[keep] for k, v in info.items()}
    return view, info, keep


def generate_json_catalogue(results: dict, quasar_info: dict, outfile: str | None = None,
                            sub_dla: bool = True, occams_razor: float = 10000.0,
                            drop_nan: bool = True) -> list:
    """``QSOLoader.generate_json_catalogue`` (qso_loader.py:1927-2031) as the reference produces it,
    i.e. on the loader's view of the processed file (:func:`loader_view`: Occam factor 10000 on
    every absorber model, ``p_dla`` / ``p_no_dla`` recomputed, NaN rows dropped -- the defaults of
    ``QSOLoader(occams_razor=10000)``).  One record per remaining quasar in the form of Parks et al.
    (2018): ``p_dla, p_no_dla, max_model_posterior, num_dlas, dlas: [{log_nhi, z_dla}], min_z_dla,
    max_z_dla, ra, snr, dec, plate, mjd, fiber_id, thing_id, z_qso``.  ``results``: the multi-DLA
    variables (``model_posteriors`` [nq, 2+max_dlas] = (no DLA, sub-DLA, 1..max_dlas DLAs),
    ``MAP_z_dlas`` / ``MAP_log_nhis`` [nq, model, slot]).  ``quasar_info``: ``ras, decs, plates,
    mjds, fiber_ids, thing_ids, z_qsos, snrs`` of the searched quasars (the catalogue columns after
    the ``test_ind`` subset, qso_loader.py:114-133)."""
    view, info, _ = loader_view(results, quasar_info, sub_dla, occams_razor, drop_nan)
    mp, p_dlas, p_no = view["model_posteriors"], view["p_dlas"], view["p_no_dlas"]
    nq = mp.shape[0]
    filled = np.where(np.isnan(mp), -np.inf, mp) if not drop_nan else mp
    model_index = filled.argmax(axis=1)                              # :1973
    max_mp = mp[np.arange(nq), model_index].copy()                   # :1974
    num_dlas = model_index.copy()
    if sub_dla:
        inds = model_index < 2                                       # :1979
        max_mp[inds] = p_no[inds]                                    # :1980
        num_dlas = model_index - 1                                   # :1983
        num_dlas[num_dlas < 0] = 0                                   # :1984
    map_z, map_n = view["MAP_z_dlas"], view["MAP_log_nhis"]
    out = []
    for i in range(nq):
        spec = {"p_dla": _py(p_dlas[i]), "p_no_dla": _py(p_no[i]), "max_model_posterior": _py(max_mp[i]),
                "num_dlas": int(num_dlas[i]), "min_z_dla": _py(view["min_z_dlas"][i]),
                "max_z_dla": _py(view["max_z_dlas"][i])}
        for key, col in _INFO_FIELDS:
            spec[key] = _py(info[col][i])
        n = int(num_dlas[i])
        spec["dlas"] = [{"log_nhi": _py(map_n[i, n - 1, j]), "z_dla": _py(map_z[i, n - 1, j])}
                        for j in range(n)]                           # :2007-2018
        out.append(spec)
    if outfile is not None:
        import json
        with open(outfile, "w") as f:
            json.dump(out, f, indent=2)
    return out


def generate_sub_dla_catalogue(results: dict, quasar_info: dict, outfile: str | None = None,
                               sub_dla: bool = True, occams_razor: float = 10000.0,
                               drop_nan: bool = True) -> list:
    """``QSOLoader.generate_sub_dla_catalogue`` (qso_loader.py:2033-2087): the quasars whose most
    probable model -- after the loader's Occam factor and NaN drop, :func:`loader_view` -- is the
    sub-DLA one, with ``p_sub_dla`` and the spectrum identifiers."""
    view, info, _ = loader_view(results, quasar_info, sub_dla, occams_razor, drop_nan)
    mp = view["model_posteriors"]
    model_index = np.where(np.isnan(mp), -np.inf, mp).argmax(axis=1)   # dla_map_model_index, :143
    out = []
    for i in np.flatnonzero(model_index == 1):
        rec = {"p_sub_dla": _py(mp[i, 1])}
        for key, col in _INFO_FIELDS:
            rec[key] = _py(info[col][i])
        out.append(rec)
    if outfile is not None:
        import json
        with open(outfile, "w") as f:
            json.dump(out, f, indent=2)
    return out
