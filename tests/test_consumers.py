"""N1's purpose, checked against the consumer it exists for: the files this
package writes, opened by the reference's OWN downstream code.

tests/golden/make_consumer_fixtures.py ran -- in the build container, under the interpreter that
has h5py -- the reference's ``mat_combine`` (sbatch_reunion.py:13-63) on the per-rank chunk files of
a world-2 GPU run, then ``QSOLoader`` (qso_loader.py:76-232, 1927-2087) and ``DLACatalogue``
(calc_cddf.py:43-160, line_density / column_density_function / omega_dla) on the result, and stored
what they READ.  Here, without the reference:

* this package's recombination of the same chunks equals what mat_combine produced, byte for byte;
* its catalogue code gives the records the reference's generate_json_catalogue gave;
* what QSOLoader / DLACatalogue read is what the inputs and the chunk files hold;
* libhdf5's own tools (h5dump / h5ls, where installed) read savemat73 output bit-exactly.
"""
import glob
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from gp_dla_detection_amd import catalog, hdf5, io, synthetic

HERE = os.path.dirname(os.path.abspath(__file__))
CONS = os.path.join(HERE, "golden", "consumer")
NQ, S = 40, 24  # tools/make_consumer_chunks.py


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    return synthetic.write_file_set(str(tmp_path_factory.mktemp("consumer_in")), num_quasars=NQ, num_samples=S,
                                    empty_quasar=None)


def chunks(multi):
    stem = "processed_qsos_multi_meanfluxsynth_" if multi else "processed_qsos_synth_"
    out = sorted(glob.glob(os.path.join(CONS, stem + "[0-9]*.mat")))
    assert len(out) == 2
    return out


@pytest.mark.parametrize("multi", [False, True])
def test_own_recombination_equals_the_references_mat_combine(tmp_path, multi):
    exp = np.load(os.path.join(CONS, f"expected_combined_{'multi' if multi else 'single'}.npz"))
    out = str(tmp_path / "combined.mat")
    io.combine_processed_chunks(chunks(multi), out)
    with hdf5.File(out) as f:
        assert sorted(k for k in f.keys()) == sorted(exp.files)
        for k in exp.files:  # HDF5 orientation on both sides; dtypes included
            a = f[k].read()
            assert a.dtype == exp[k].dtype and a.shape == exp[k].shape, (k, a.dtype, exp[k].dtype, a.shape, exp[k].shape)
            np.testing.assert_array_equal(a, exp[k], err_msg=k)
    # and the in-package reader returns the run in quasar order
    whole = io.load_processed_qsos(out)
    parts = [io.load_processed_qsos(c) for c in chunks(multi)]
    np.testing.assert_array_equal(whole["sample_log_likelihoods_dla"],
                                  np.concatenate([p["sample_log_likelihoods_dla"] for p in parts]))
    assert whole["test_set_name"] == "synth" and whole["p_dlas"].shape == (32,)


def quasar_info(fs):
    t = fs["test_ind"]
    c = fs["catalog"]
    return dict(ras=c["ras"][t], decs=c["decs"][t], plates=c["plates"][t], mjds=c["mjds"][t],
                fiber_ids=c["fiber_ids"][t], thing_ids=c["thing_ids"][t], z_qsos=c["z_qsos"][t], snrs=c["snrs"][t])


def whole_run(tmp_path, multi):
    out = str(tmp_path / f"whole_{int(multi)}.mat")
    io.combine_processed_chunks(chunks(multi), out)
    return io.load_processed_qsos(out)


def test_json_catalogues_equal_the_references(tmp_path, inputs):
    """generate_json_catalogue / generate_sub_dla_catalogue of the reference's QSOLoader, run on
    these chunk files, against catalog.py on the same files: every record, every field."""
    res = whole_run(tmp_path, True)
    info = quasar_info(inputs)
    mine = catalog.generate_json_catalogue(res, info, str(tmp_path / "mine.json"))
    ref = json.load(open(os.path.join(CONS, "expected_predictions_multi_DLAs.json")))
    assert len(ref) == 32 and json.load(open(tmp_path / "mine.json")) == ref
    assert any(r["num_dlas"] >= 1 for r in ref) and any(r["num_dlas"] == 0 for r in ref)
    for a, b in zip(mine, ref):
        assert a == b
    sub = catalog.generate_sub_dla_catalogue(res, info)
    ref_sub = json.load(open(os.path.join(CONS, "expected_predictions_sub_DLA_candidates.json")))
    assert len(ref_sub) == 8 and json.loads(json.dumps(sub)) == ref_sub


@pytest.mark.parametrize("multi", [False, True])
def test_what_qsoloader_read(tmp_path, inputs, multi):
    """QSOLoader.__init__ (qso_loader.py:84-232) on the combined file + the input files."""
    q = np.load(os.path.join(CONS, f"expected_qsoloader_{'multi' if multi else 'single'}.npz"))
    fs = inputs
    res = whole_run(tmp_path, multi)
    np.testing.assert_array_equal(q["test_ind"], fs["test_ind"])
    np.testing.assert_array_equal(q["test_real_index"], np.flatnonzero(fs["test_ind"]))
    assert not q["nan_inds"].any()
    info = quasar_info(fs)
    view, vinfo, keep = catalog.loader_view(res, info, sub_dla=multi)
    np.testing.assert_array_equal(keep, np.arange(32))
    np.testing.assert_array_equal(q["model_posteriors"], view["model_posteriors"])  # Occam factor applied (:136)
    np.testing.assert_array_equal(q["p_dlas"], view["p_dlas"])
    np.testing.assert_array_equal(q["p_no_dlas"], view["p_no_dlas"])
    for key, col in (("thing_ids", "thing_ids"), ("plates", "plates"), ("mjds", "mjds"), ("fiber_ids", "fiber_ids"),
                     ("z_qsos", "z_qsos"), ("snrs", "snrs"), ("snrs_cat", "snrs")):
        np.testing.assert_array_equal(q[key], info[col], err_msg=key)
    np.testing.assert_array_equal(q["min_z_dlas"], res["min_z_dlas"])
    np.testing.assert_array_equal(q["max_z_dlas"], res["max_z_dlas"])
    lp = res["log_priors_dla"]
    np.testing.assert_array_equal(q["log_priors_dla"], lp if lp.ndim == 1 else lp[:, 0])  # f['log_priors_dla'][0, :]
    # the learned model, as GPLoader holds it (:210-218)
    m = fs["model"]
    np.testing.assert_array_equal(q["GP_mu"], m["mu"])
    np.testing.assert_array_equal(q["GP_M"], m["M"])
    np.testing.assert_array_equal(q["GP_log_omega"], m["log_omega"])
    np.testing.assert_array_equal(q["GP_rest_wavelengths"], m["rest_wavelengths"])
    np.testing.assert_array_equal(q["GP_scalars"], [m["log_tau_0"], m["log_beta"], m["log_c_0"]])
    # a spectrum through the cell references of preloaded_qsos.mat (:1592-1627)
    sp = fs["spectra"][int(np.flatnonzero(fs["test_ind"])[3])]
    np.testing.assert_array_equal(q["flux_3"], sp["flux"])
    np.testing.assert_array_equal(q["wavelengths_3"], sp["wavelengths"])
    np.testing.assert_array_equal(q["noise_variance_3"], sp["noise_variance"])
    # MAP number of absorbers (:140-183)
    idx = np.argmax(view["model_posteriors"], axis=1)
    np.testing.assert_array_equal(q["dla_map_model_index"], idx)
    num = idx - int(multi)
    num[num < 0] = 0
    np.testing.assert_array_equal(q["dla_map_num_dla"], num)
    if not multi:
        # prepare_roman_map_vals (:303-373): the reference's argmax over OUR sample table, against the
        # MAP columns the evidence kernel found on the GPU (generate_ascii_catalog.m:73-80)
        np.testing.assert_array_equal(q["all_log_nhis"], res["MAP_log_nhis"])
        np.testing.assert_array_equal(q["all_z_dlas"], res["MAP_z_dlas"])
    if multi:
        np.testing.assert_array_equal(q["map_z_dlas"], res["MAP_z_dlas"])      # f['MAP_z_dlas'][()].T (:107-109)
        np.testing.assert_array_equal(q["map_log_nhis"], res["MAP_log_nhis"])
        for i, n in enumerate(num):  # prepare_map_vals (:285-301)
            np.testing.assert_array_equal(q["all_z_dlas"][i, :n], res["MAP_z_dlas"][i, n - 1, :n] if n else [])
            assert np.isnan(q["all_z_dlas"][i, n:]).all()
        assert q["roc_tpr"].size == 32 and q["map_comparison_dz"].size >= 1


@pytest.mark.parametrize("multi", [False, True])
def test_what_dlacatalogue_read(tmp_path, inputs, multi):
    """DLACatalogue.__init__ (calc_cddf.py:72-160, 205-284): redshift ranges, the normalised sample
    likelihoods of the spectra it keeps, the DLA(2) tables and their base sample indices."""
    c = np.load(os.path.join(CONS, f"expected_dlacatalogue_{'multi' if multi else 'single'}.npz"))
    res = whole_run(tmp_path, multi)
    fs = inputs
    np.testing.assert_array_equal(c["z_min"], res["min_z_dlas"])
    np.testing.assert_array_equal(c["z_max"], res["max_z_dlas"])
    np.testing.assert_array_equal(c["real_index"], np.flatnonzero(fs["test_ind"]))
    np.testing.assert_array_equal(c["snrs"], fs["catalog"]["snrs"][fs["test_ind"]])
    np.testing.assert_array_equal(c["z_offsets"], fs["samples"]["offset_samples"])
    np.testing.assert_array_equal(c["lnhi_vals"], fs["samples"]["log_nhi_samples"])
    np.testing.assert_array_equal(c["model_posteriors"], catalog.occams_model_posteriors(res["model_posteriors"]))
    sll = res["sample_log_likelihoods_dla"]
    first = sll[:, 0, :] if multi else sll                     # f[...][0, :, :] (:217-220)
    ll = res["log_likelihoods_dla"][:, 0] if multi else res["log_likelihoods_dla"]
    assert c["cached_spectra"].size >= 1
    for row, spec in zip(c["log_norm_like"], c["cached_spectra"]):   # :223-229
        np.testing.assert_array_equal(row, first[spec] - (ll[spec] + np.log(S)))
    if multi:
        np.testing.assert_array_equal(c["p_dla_2"], c["model_posteriors"][:, 3])
        for row, spec in zip(c["base_sample_inds_2"], c["cached_spectra_2"]):   # :266-282: 0-based
            np.testing.assert_array_equal(row, res["base_sample_inds"][spec, 0].astype(np.int64) - 1)
    report = json.load(open(os.path.join(CONS, "report.json")))
    assert report["multi" if multi else "single"]["dlacatalogue_methods"] == {
        "line_density": "ok", "column_density_function": "ok", "omega_dla": "ok"}
    assert np.isfinite(c["line_density_1"]).all() and c["column_density_function_1"].size == 6


def test_committed_chunks_are_what_the_package_reads(inputs):
    """The chunk fixtures carry their own test_ind (the chunk's quasars only) and the run metadata."""
    sel = np.flatnonzero(inputs["test_ind"])
    lo = 0
    for path in chunks(False):
        part = io.load_processed_qsos(path)
        n = part["p_dlas"].size
        mask = np.zeros(NQ, dtype=bool)
        mask[sel[lo:lo + n]] = True
        np.testing.assert_array_equal(np.asarray(part["test_ind"]).reshape(-1).astype(bool), mask)
        assert part["release"] == "dr12q" and "MAP_z_dlas" in part and part["MAP_z_dlas"].shape == (n,)
        with hdf5.File(path) as f:  # QSOLoader keys a multi-DLA file on 'MAP_log_nhis' (qso_loader.py:106)
            assert "MAP_log_nhis" not in f and "single_MAP_log_nhis" in f
        lo += n
    assert lo == 32


def test_model_posteriors_equal_the_references_reevaluation():
    """A reference-PRODUCED anchor of the posterior softmax (multi :482-495): the reference restates
    it as QSOLoader.reevaluate_model_posteriors (qso_loader.py:260-283), which was run on the
    committed multi-DLA chunk files (tests/golden/make_consumer_fixtures.py).  The model_posteriors
    in those files were computed on the GPU by k_multi_posteriors from the same log posteriors."""
    ref = np.load(os.path.join(CONS, "expected_reevaluated_posteriors_multi.npz"))
    parts = [io.load_processed_qsos(p) for p in chunks(True)]
    got = np.concatenate([p["model_posteriors"] for p in parts], axis=0)
    lpost = np.concatenate([np.column_stack([p["log_posteriors_no_dla"], p["log_posteriors_lls"],
                                             p["log_posteriors_dla"]]) for p in parts], axis=0)
    np.testing.assert_array_equal(lpost, ref["log_posteriors"])  # the reference read what the files hold
    assert got.shape == ref["model_posteriors"].shape == (32, 5)
    np.testing.assert_allclose(got, ref["model_posteriors"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.concatenate([p["p_dlas"] for p in parts]),
                               1 - ref["model_posteriors"][:, 0] - ref["model_posteriors"][:, 1], rtol=0, atol=1e-12)
    assert (ref["model_posteriors"].max(axis=1) > 0.5).all() and len({int(i) for i in ref["model_posteriors"].argmax(axis=1)}) >= 2


# ---------------------------------------------------------------------------------------------
# libhdf5 cross-read of the writer (h5dump / h5ls from the conda environment of the build image)
# ---------------------------------------------------------------------------------------------

H5DUMP = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
H5LS = shutil.which("h5ls") or "/opt/conda/bin/h5ls"
needs_libhdf5 = pytest.mark.skipif(not (os.path.exists(H5DUMP) and os.path.exists(H5LS)),
                                   reason="no libhdf5 command-line tools on this box")


def h5dump_binary(path, name, dtype, tmp):
    out = os.path.join(tmp, "dump.bin")
    subprocess.run([H5DUMP, "-d", name, "-b", "LE", "-o", out, path], check=True, capture_output=True)
    return np.fromfile(out, dtype=dtype)


@needs_libhdf5
def test_libhdf5_reads_savemat73_output_bit_exactly(tmp_path):
    """Contiguous, chunked + deflate, logical, uint32, char and cell (object-reference) variables
    behind MATLAB's 512-byte user block, read back by libhdf5 1.10's own h5dump / h5ls."""
    rng = np.random.default_rng(5)
    big = rng.normal(size=(300, 700))                       # > 1 MiB: several deflate chunks
    variables = dict(table=big, vec=rng.normal(size=17), flag=rng.uniform(size=(9, 1)) < 0.5,
                     inds=rng.integers(0, 2 ** 32 - 1, size=(4, 5, 3), dtype=np.uint32), name="dr12q",
                     cells=[rng.normal(size=(n, 1)) for n in (5, 1, 12)], scalar=np.float64(3.25))
    for compress in (False, True):
        p = str(tmp_path / f"x{int(compress)}.mat")
        io.savemat73(p, variables, compress=compress)
        ls = subprocess.run([H5LS, "-r", p], check=True, capture_output=True, text=True).stdout
        for name in ("/table", "/vec", "/flag", "/inds", "/name", "/cells", "/scalar", "/#refs#"):
            assert name in ls, ls
        assert "{700, 300}" in ls and "{3, 5, 4}" in ls  # dimensions reversed, as MATLAB stores them
        np.testing.assert_array_equal(h5dump_binary(p, "/table", "<f8", str(tmp_path)).reshape(700, 300).T, big)
        np.testing.assert_array_equal(h5dump_binary(p, "/vec", "<f8", str(tmp_path)), variables["vec"])
        np.testing.assert_array_equal(h5dump_binary(p, "/flag", "u1", str(tmp_path)), variables["flag"].ravel())
        np.testing.assert_array_equal(h5dump_binary(p, "/inds", "<u4", str(tmp_path)).reshape(3, 5, 4).T,
                                      variables["inds"])
        assert "".join(map(chr, h5dump_binary(p, "/name", "<u2", str(tmp_path)))) == "dr12q"
        assert h5dump_binary(p, "/scalar", "<f8", str(tmp_path))[0] == 3.25
        # the cell array: object references that libhdf5 resolves to the datasets under #refs#
        txt = subprocess.run([H5DUMP, "-d", "/cells", p], check=True, capture_output=True, text=True).stdout
        refs = [w for w in txt.replace(",", " ").split() if w.startswith("/#refs#/")]
        assert len(refs) == 3, txt
        for ref, cell in zip(refs, variables["cells"]):
            np.testing.assert_array_equal(h5dump_binary(p, ref, "<f8", str(tmp_path)), cell.ravel())
        head = subprocess.run([H5DUMP, "-H", "-B", p], check=True, capture_output=True, text=True).stdout
        assert "USERBLOCK_SIZE 512" in head and 'MATLAB_class' in subprocess.run(
            [H5DUMP, "-A", p], check=True, capture_output=True, text=True).stdout


@needs_libhdf5
def test_libhdf5_reads_the_committed_chunk_files():
    for multi in (False, True):
        for path in chunks(multi):
            part = io.load_processed_qsos(path)
            n = part["p_dlas"].size
            tmp = os.path.dirname(path)
            out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gpdla_chunk_dump.bin")
            subprocess.run([H5DUMP, "-d", "/sample_log_likelihoods_dla", "-b", "LE", "-o", out, path], check=True,
                           capture_output=True)
            raw = np.fromfile(out, dtype="<f8")
            if multi:
                np.testing.assert_array_equal(raw.reshape(3, S, n).transpose(2, 0, 1), part["sample_log_likelihoods_dla"])
            else:
                np.testing.assert_array_equal(raw.reshape(S, n).T, part["sample_log_likelihoods_dla"])
            os.remove(out)


@needs_libhdf5
def test_libhdf5_reads_the_streamed_chunk_writer(tmp_path):
    """What a sharded run writes batch by batch (io.ProcessedStreamWriter: one column of unfiltered
    HDF5 chunks per batch, ragged last batch, chunk index written at the end), read back by
    libhdf5's h5dump / h5ls and equal to the one-shot writers' files variable for variable."""
    from gp_dla_detection_amd.api import Batch
    rng = np.random.default_rng(11)
    nq, S_, B = 29, 70, 8
    mask = np.zeros(40, bool)
    mask[:nq] = True
    for md in (0, 3):
        res = Batch.empty_results_multi(nq, md, S_) if md else Batch.empty_results(nq, S_)
        for k, v in res.items():
            if isinstance(v, np.ndarray) and v.dtype == np.float64:
                v[...] = rng.standard_normal(v.shape)
            if isinstance(v, np.ndarray) and v.dtype == np.uint32:
                v[...] = rng.integers(1, S_ + 1, v.shape)
        res.update(num_lines=3, prior_z_qso_increase=0.1, max_z_cut=0.1, k=20, min_z_cut=0.1, num_dla_samples=S_)
        one, streamed = str(tmp_path / f"one{md}.mat"), str(tmp_path / f"streamed{md}.mat")
        (io.save_processed_qsos_multi if md else io.save_processed_qsos)(one, res, test_ind=mask, test_set_name="t")
        w = io.ProcessedStreamWriter(streamed, nq, S_, B, max_dlas=md)
        for at in range(0, nq, B):
            w.append(at, {k: res[k][at:at + B] for k in w.streamed})
        w.finish(res, test_ind=mask, test_set_name="t")
        a, b = io.load_processed_qsos(one), io.load_processed_qsos(streamed)
        assert sorted(a) == sorted(b)
        for k in a:
            x, y = np.asarray(a[k]), np.asarray(b[k])
            assert x.shape == y.shape and x.dtype == y.dtype, k
            assert np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y), k
        ls = subprocess.run([H5LS, "-v", streamed], check=True, capture_output=True, text=True).stdout
        assert "Chunks:" in ls and (f"{{{S_}/{S_}, {nq}/{nq}}}" in ls)
        raw = h5dump_binary(streamed, "/sample_log_likelihoods_dla", "<f8", str(tmp_path))
        want = res["sample_log_likelihoods_dla"]
        if md:
            np.testing.assert_array_equal(raw.reshape(md, S_, nq).transpose(2, 0, 1), want)
            np.testing.assert_array_equal(h5dump_binary(streamed, "/base_sample_inds", "<u4", str(tmp_path))
                                          .reshape(md - 1, S_, nq).transpose(2, 0, 1), res["base_sample_inds"])
        else:
            np.testing.assert_array_equal(raw.reshape(S_, nq).T, want)
