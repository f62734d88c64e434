#!/opt/conda/bin/python3.9
"""BUILD CONTAINER ONLY: let the reference's own downstream code open this package's files.

    /opt/conda/bin/python3.9 tests/golden/make_consumer_fixtures.py [chunk_dir]

Run under the interpreter that has h5py (3.3.0, libhdf5 1.10.6); imports, from /root/reference,
``CDDF_analysis.sbatch_reunion.mat_combine`` (the reference's only multi-node mechanism,
sbatch_reunion.py:13-63), ``qso_loader.QSOLoader`` (:76-232, :1927-2087) and
``calc_cddf.DLACatalogue`` (:43-160).  Inputs:

* the synthetic ``-v7.3`` file set written by ``gp_dla_detection_amd.synthetic.write_file_set``
  (regenerated here, by the system python, into a scratch directory: it is seeded);
* the per-rank chunk files ``processed_qsos_*_<lo>-<hi>.mat`` of a world-2 run of
  ``gp_dla_detection_amd.run_dr12q`` on a real MI355X (``tools/make_consumer_chunks.py`` on the GPU
  box; committed under tests/golden/consumer/ -- they are this package's OUTPUT, i.e. data).

What the reference code READ from those files is stored as ``tests/golden/consumer/expected_*.npz``
/ ``*.json``; tests/test_consumers.py then checks, without the reference, that this package's own
readers and catalogue code give the same.  Nothing of the reference is copied: only numbers it
computed travel.

The reference predates NumPy 1.24 and uses the removed aliases np.bool / np.int / np.float; this
script defines them before importing it (the reference files are untouched)."""
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

np.bool, np.int, np.float = bool, int, float  # aliases the reference still uses (removed in NumPy 1.24)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "consumer")
sys.path.insert(0, "/root/reference")

import h5py  # noqa: E402
from CDDF_analysis import calc_cddf, qso_loader, sbatch_reunion  # noqa: E402

NQ, S = 40, 24  # more searched quasars (32) than samples: calc_cddf.py:940 indexes quasars by sample index


def make_inputs(d):
    code = ("import sys; sys.path.insert(0, %r); from gp_dla_detection_amd import synthetic; "
            "synthetic.write_file_set(%r, num_quasars=%d, num_samples=%d, empty_quasar=None)" % (ROOT, d, NQ, S))
    subprocess.run(["/usr/bin/python3", "-c", code], check=True)


def reference_combine(chunks, out_path):
    """mat_combine as the reference's user calls it, then the run-level test_ind: mat_combine copies
    every variable that is not per-quasar from the FIRST chunk (sbatch_reunion.py:39-43), so the
    combined file carries that chunk's selection; the union of the chunks' masks is put in its
    place (in the reference's workflow the same edit is done by hand)."""
    sizes = [h5py.File(c, "r")["p_dlas"].shape[-1] for c in chunks]
    sbatch_reunion.mat_combine(chunks, out_path, chunk_size=sizes[0], maxshape_size=sum(sizes))
    union = None
    for c in chunks:
        with h5py.File(c, "r") as f:
            t = f["test_ind"][()].astype(bool)
            union = t if union is None else (union | t)
    with h5py.File(out_path, "r+") as f:
        f["test_ind"][()] = union.astype(f["test_ind"].dtype)
    return sizes


def dump_loader(q, multi):
    keep = dict(test_ind=q.test_ind, test_real_index=q.test_real_index, model_posteriors=q.model_posteriors,
                p_dlas=q.p_dlas, p_no_dlas=q.p_no_dlas, log_priors_dla=q.log_priors_dla,
                min_z_dlas=q.min_z_dlas, max_z_dlas=q.max_z_dlas, thing_ids=q.thing_ids, plates=q.plates,
                mjds=q.mjds, fiber_ids=q.fiber_ids, z_qsos=q.z_qsos, snrs=q.snrs, snrs_cat=q.snrs_cat,
                nan_inds=q.nan_inds, multi_p_dlas=q.multi_p_dlas, dla_map_model_index=q.dla_map_model_index,
                dla_map_num_dla=q.dla_map_num_dla, model_posteriors_dla=q.model_posteriors_dla,
                GP_mu=q.GP.mu, GP_M=q.GP.M, GP_log_omega=q.GP.log_omega, GP_rest_wavelengths=q.GP.rest_wavelengths,
                GP_scalars=np.array([q.GP.log_tau_0, q.GP.log_beta, q.GP.log_c_0]),
                concordance_real_index=q.dla_catalog.real_index, concordance_real_index_los=q.dla_catalog.real_index_los,
                concordance_z_dlas=q.dla_catalog.z_dlas, concordance_log_nhis=q.dla_catalog.log_nhis,
                flux_3=q.find_this_flux(3), wavelengths_3=q.find_this_wavelengths(3),
                noise_variance_3=q.find_this_noise_variance(3))
    if not multi:  # the MAP sample of every quasar, found by the reference from the sample table (:303-373)
        q.prepare_roman_map_vals(sample_file=q.sample_file)
        keep.update(all_log_nhis=q.all_log_nhis, all_z_dlas=q.all_z_dlas)
    if multi:
        keep.update(map_log_nhis=q.map_log_nhis, map_z_dlas=q.map_z_dlas, all_log_nhis=q.all_log_nhis,
                    all_z_dlas=q.all_z_dlas)
        dz, dn = q.make_MAP_comparison(q.dla_catalog)
        keep.update(map_comparison_dz=dz, map_comparison_dlognhi=dn)
    try:  # make_ROC asserts an ordering of the data (qso_loader.py:697) that synthetic quasars need not have
        tpr, fpr = q.make_ROC(q.dla_catalog, occams_razor=q.occams_razor)
        keep.update(roc_tpr=np.array(tpr), roc_fpr=np.array(fpr))
        roc = "ok"
    except AssertionError:
        roc = "AssertionError at qso_loader.py:697 (data-dependent ordering check)"
    return {k: np.asarray(v) for k, v in keep.items()}, roc


def dump_cddf(c, multi):
    keep = dict(z_min=c._z_min, z_max=c._z_max, z_qsos=c.z_qsos, real_index=c.real_index, snrs=c.snrs,
                model_posteriors=c.model_posteriors, p_dla=c.p_dla, p_no_dla=c.p_no_dla,
                z_offsets=c.z_offsets, lnhi_vals=c.lnhi_vals,
                cached_spectra=np.array(sorted(c.log_norm_like_cache)),
                log_norm_like=np.stack([c.log_norm_like_cache[s] for s in sorted(c.log_norm_like_cache)])
                if c.log_norm_like_cache else np.zeros((0, S)))
    if multi:
        keep.update(p_dla_2=c.p_dla_2, cached_spectra_2=np.array(sorted(c.log_norm_like_2_cache)))
        if c.log_norm_like_2_cache:
            keep.update(log_norm_like_2=np.stack([c.log_norm_like_2_cache[s] for s in sorted(c.log_norm_like_2_cache)]),
                        base_sample_inds_2=np.stack([c.base_sample_inds_2_cache[s] for s in sorted(c.base_sample_inds_2_cache)]))
    ran = {}
    for name, call in (("line_density", lambda: c.line_density(z_min=2, z_max=5)),
                       ("column_density_function", lambda: c.column_density_function(z_min=2., z_max=5., lnhi_nbins=6)),
                       ("omega_dla", lambda: c.omega_dla(z_min=2, z_max=5))):
        try:
            res = call()
            for j, part in enumerate(res if isinstance(res, tuple) else (res,)):
                keep[f"{name}_{j}"] = np.asarray(part, dtype=np.float64)
            ran[name] = "ok"
        except Exception as e:  # recorded, not hidden: the fixture says how far the reference got
            ran[name] = f"{type(e).__name__}: {e}"
    return {k: np.asarray(v) for k, v in keep.items()}, ran


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "consumer")
    os.makedirs(OUT, exist_ok=True)
    for kind in ("single", "multi"):  # the chunk files become committed fixtures
        for f in sorted(glob.glob(os.path.join(src, kind, "processed_qsos_*[0-9].mat"))) if src != "-" else []:
            shutil.copy(f, os.path.join(OUT, os.path.basename(f)))
    report = {}
    with tempfile.TemporaryDirectory() as d:
        make_inputs(d)
        for kind, stem, multi in (("single", "processed_qsos_synth_", False),
                                  ("multi", "processed_qsos_multi_meanfluxsynth_", True)):
            chunks = sorted(glob.glob(os.path.join(OUT, stem + "[0-9]*.mat")))
            assert len(chunks) == 2, chunks
            combined = os.path.join(d, f"combined_{kind}.h5")
            sizes = reference_combine(chunks, combined)
            with h5py.File(combined, "r") as f:
                raw = {k: f[k][()] for k in f.keys()}
            np.savez_compressed(os.path.join(OUT, f"expected_combined_{kind}.npz"), **raw)
            q = qso_loader.QSOLoader(
                preloaded_file=f"{d}/preloaded_qsos.mat", catalogue_file=f"{d}/catalog.mat",
                learned_file=f"{d}/learned_qso_model_synthetic.mat", processed_file=combined,
                dla_concordance=f"{d}/dla_catalog", los_concordance=f"{d}/los_catalog",
                snrs_file=f"{d}/snrs_qsos.mat", sub_dla=multi, sample_file=f"{d}/dla_samples.mat",
                occams_razor=10000)
            ql, roc = dump_loader(q, multi)
            np.savez_compressed(os.path.join(OUT, f"expected_qsoloader_{kind}.npz"), **ql)
            if multi:
                cat = q.generate_json_catalogue(outfile=os.path.join(OUT, "expected_predictions_multi_DLAs.json"))
                sub = q.generate_sub_dla_catalogue(outfile=os.path.join(OUT, "expected_predictions_sub_DLA_candidates.json"))
                report["json_records"] = [len(cat), len(sub)]
                # The reference restates the (2 + max_dlas)-way softmax of multi :482-495 in Python:
                # QSOLoader.reevaluate_model_posteriors (qso_loader.py:260-283) recomputes model_posteriors
                # from the file's log_posteriors_*.  It only does so when some row of the posteriors it
                # holds sums to more than 1.2, so that condition is made true first (LAST use of q: the
                # call replaces q.model_posteriors).  What it returns anchors k_multi_posteriors.
                q.model_posteriors = np.asarray(q.model_posteriors) * 2.0
                q.reevaluate_model_posteriors()
                np.savez_compressed(os.path.join(OUT, "expected_reevaluated_posteriors_multi.npz"),
                                    model_posteriors=np.asarray(q.model_posteriors),
                                    log_posteriors=np.asarray(q.log_posteriors))
                report["reevaluate_model_posteriors"] = "ok: %d quasars x %d models" % np.asarray(q.model_posteriors).shape
            c = calc_cddf.DLACatalogue(processed_file=combined, sample_file=f"{d}/dla_samples.mat",
                                       raw_file=f"{d}/preloaded_qsos.mat", snrs_file=f"{d}/snrs_qsos.mat",
                                       snr=-2, second=1 if multi else False, sub_dla=multi, occams_razor=10000)
            cd, ran = dump_cddf(c, multi)
            np.savez_compressed(os.path.join(OUT, f"expected_dlacatalogue_{kind}.npz"), **cd)
            report[kind] = dict(chunk_sizes=sizes, combined_keys=sorted(raw), dlacatalogue_methods=ran, make_ROC=roc,
                                qsoloader_quasars=int(q.test_ind.sum()))
    report["versions"] = dict(python=sys.version.split()[0], numpy=np.__version__, h5py=h5py.__version__,
                              hdf5=h5py.version.hdf5_version)
    with open(os.path.join(OUT, "report.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print(json.dumps(report, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
