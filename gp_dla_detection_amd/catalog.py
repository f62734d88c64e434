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
