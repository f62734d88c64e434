"""Round trips of the .mat readers/writers: v5 files through scipy, -v7.3 (HDF5) files through the
package's own HDF5 implementation (tests/test_hdf5.py tests that layer by itself)."""
import numpy as np
import pytest
from scipy.io import savemat

from gp_dla_detection_amd import io, synthetic


def test_model_and_samples_round_trip(tmp_path):
    model = synthetic.make_model(20)
    savemat(tmp_path / "m.mat", {k: (np.asarray(v).reshape(-1, 1) if np.ndim(v) == 1 else v)
                                 for k, v in model.items()})
    got = io.load_learned_model(str(tmp_path / "m.mat"))
    for k in ("rest_wavelengths", "mu", "log_omega"):
        np.testing.assert_array_equal(got[k], model[k])
    np.testing.assert_array_equal(got["M"], model["M"])
    assert got["log_beta"] == model["log_beta"]
    s = synthetic.make_samples(50)
    savemat(tmp_path / "s.mat", {k: v.reshape(1, -1) for k, v in s.items()})  # row vectors, as MATLAB
    gs = io.load_dla_samples(str(tmp_path / "s.mat"))
    np.testing.assert_array_equal(gs["nhi_samples"], s["nhi_samples"])
    np.testing.assert_array_equal(gs["lls_nhi_samples"], s["lls_nhi_samples"])


def test_preloaded_cells_and_output(tmp_path):
    model = synthetic.make_model(20)
    spectra = [synthetic.make_spectrum(i, 50 + 7 * i, model, mask_fraction=0.1) for i in range(4)]
    cells = {}
    for key, src in (("all_wavelengths", "wavelengths"), ("all_flux", "flux"),
                     ("all_noise_variance", "noise_variance"), ("all_pixel_mask", "pixel_mask")):
        c = np.empty((4, 1), dtype=object)
        for i, sp in enumerate(spectra):
            c[i, 0] = np.asarray(sp[src]).reshape(-1, 1)
        cells[key] = c
    savemat(tmp_path / "p.mat", cells)
    z = [sp["z_qso"] for sp in spectra]
    got = io.load_preloaded_qsos(str(tmp_path / "p.mat"), z, test_ind=np.array([True, False, True, True]))
    assert len(got) == 3 and got[1]["z_qso"] == z[2]
    np.testing.assert_array_equal(got[1]["wavelengths"], spectra[2]["wavelengths"])
    np.testing.assert_array_equal(got[2]["pixel_mask"], spectra[3]["pixel_mask"])
    assert np.array_equal(np.isnan(got[0]["flux"]), np.isnan(spectra[0]["flux"]))


def cells_of(spectra):
    out = {}
    for key, src in (("all_wavelengths", "wavelengths"), ("all_flux", "flux"),
                     ("all_noise_variance", "noise_variance"), ("all_pixel_mask", "pixel_mask")):
        out[key] = [np.asarray(sp[src]).astype(bool if src == "pixel_mask" else np.float64).reshape(-1, 1)
                    for sp in spectra]
    return out


def test_v73_inputs_round_trip(tmp_path):
    """The three input files of process_qsos.m:30-49 as -v7.3 (HDF5) files, as the reference
    saves them: model, samples (row vectors), preloaded ragged cells incl. logical masks."""
    model = synthetic.make_model(20)
    io.savemat73(str(tmp_path / "m.mat"), {k: (np.asarray(v).reshape(-1, 1) if np.ndim(v) == 1 else v)
                                           for k, v in model.items()}, compress=True)
    got = io.load_learned_model(str(tmp_path / "m.mat"))
    for k in ("rest_wavelengths", "mu", "log_omega", "M"):
        np.testing.assert_array_equal(got[k], model[k])
    assert got["log_tau_0"] == model["log_tau_0"]
    s = synthetic.make_samples(64)
    io.savemat73(str(tmp_path / "s.mat"), {k: v.reshape(1, -1) for k, v in s.items()})
    gs = io.load_dla_samples(str(tmp_path / "s.mat"))
    for k in ("offset_samples", "log_nhi_samples", "nhi_samples", "lls_nhi_samples"):
        np.testing.assert_array_equal(gs[k], s[k])
    spectra = [synthetic.make_spectrum(i, 50 + 7 * i, model, mask_fraction=0.1) for i in range(5)]
    io.savemat73(str(tmp_path / "p.mat"), cells_of(spectra), compress=True)
    z = [sp["z_qso"] for sp in spectra]
    got = io.load_preloaded_qsos(str(tmp_path / "p.mat"), z, test_ind=np.array([True, False, True, True, False]))
    assert len(got) == 3 and got[1]["z_qso"] == z[2]
    np.testing.assert_array_equal(got[1]["wavelengths"], spectra[2]["wavelengths"])
    np.testing.assert_array_equal(got[2]["pixel_mask"], spectra[3]["pixel_mask"])
    np.testing.assert_array_equal(got[0]["noise_variance"], spectra[0]["noise_variance"])
    assert np.array_equal(np.isnan(got[0]["flux"]), np.isnan(spectra[0]["flux"]))
    by_index = io.load_preloaded_qsos(str(tmp_path / "p.mat"), z, test_ind=np.array([4, 0]))
    np.testing.assert_array_equal(by_index[0]["wavelengths"], spectra[4]["wavelengths"])


def test_processed_qsos_v73_as_the_consumer_indexes_it(tmp_path):
    """processed_qsos_*.mat as -v7.3, opened the way CDDF_analysis does (qso_loader.py:84-112,
    calc_cddf.py:217-220): ``f[name][0, :]`` for column vectors, ``f['model_posteriors'][()].T``,
    ``f['test_ind'][0, :]``, ``sample_log_likelihoods_dla`` as [S, nq]."""
    from gp_dla_detection_amd import hdf5
    rng = np.random.default_rng(2)
    nq, S = 6, 40
    res = {k: rng.standard_normal(nq) for k in io.SAVED_VARIABLES}
    res["sample_log_likelihoods_dla"] = rng.standard_normal((nq, S))
    res["model_posteriors"] = rng.uniform(size=(nq, 2))
    res.update(num_lines=3, max_z_cut=0.01, prior_z_qso_increase=0.1,
               MAP_inds=np.arange(1.0, nq + 1), MAP_z_dlas=rng.uniform(2, 3, nq), MAP_log_nhis=rng.uniform(20, 22, nq))
    test_ind = np.array([True, False, True, True, False, True, True, True, False])
    p = str(tmp_path / "processed_qsos_dr12q.mat")
    io.save_processed_qsos(p, res, test_ind=test_ind, test_set_name="dr12q", training_release="dr12q",
                           release="dr12q", training_set_name="dr9q_minus_concordance",
                           dla_catalog_name="dr9q_concordance", prior_ind="prior_catalog.in_dr9")
    with hdf5.File(p) as f:  # h5py-style indexing
        assert f.userblock()[:10] == b"MATLAB 7.3"
        np.testing.assert_array_equal(f["test_ind"][0, :].astype(bool), test_ind)
        np.testing.assert_array_equal(f["p_dlas"][0, :], res["p_dlas"])
        np.testing.assert_array_equal(f["min_z_dlas"][0, :], res["min_z_dlas"])
        np.testing.assert_array_equal(f["model_posteriors"][()].T, res["model_posteriors"])
        assert f["sample_log_likelihoods_dla"].shape == (S, nq)
        np.testing.assert_array_equal(f["sample_log_likelihoods_dla"][:, 2], res["sample_log_likelihoods_dla"][2])
        assert f["num_lines"][0, 0] == 3
    back = io.load_processed_qsos(p)
    assert back["test_set_name"] == "dr12q" and back["prior_ind"] == "prior_catalog.in_dr9"
    for k in io.SAVED_VARIABLES + ("MAP_inds", "MAP_z_dlas"):
        np.testing.assert_array_equal(back[k], res[k])


def test_processed_qsos_multi_v73(tmp_path):
    """The multi-DLA variable list (multi :498-523) with MATLAB's axis order: sample table
    [nq x S x max_dlas] read back as [max_dlas, S, nq] (calc_cddf.py:266), base_sample_inds uint32
    [nq x S x max_dlas-1] (:116), MAP_* [nq x model x slot] (qso_loader.py:107-109)."""
    from gp_dla_detection_amd import hdf5
    rng = np.random.default_rng(3)
    nq, S, md = 5, 24, 4
    res = {k: rng.standard_normal(nq) for k in io.SAVED_VARIABLES_MULTI}
    res["sample_log_likelihoods_dla"] = rng.standard_normal((nq, md, S))
    res["sample_log_likelihoods_lls"] = rng.standard_normal((nq, S))
    res["base_sample_inds"] = rng.integers(1, S + 1, size=(nq, md - 1, S)).astype(np.uint32)
    for k in ("log_priors_dla", "log_likelihoods_dla", "log_posteriors_dla"):
        res[k] = rng.standard_normal((nq, md))
    res["model_posteriors"] = rng.uniform(size=(nq, 2 + md))
    for k in ("MAP_z_dlas", "MAP_log_nhis", "MAP_inds"):
        res[k] = rng.standard_normal((nq, md, md))
    res["all_exceptions"] = np.array([np.nan, 1.0, np.nan, np.nan, np.nan])
    p = str(tmp_path / "processed_qsos_multi_meanflux.mat")
    io.save_processed_qsos_multi(p, res, test_ind=np.ones(nq, bool), k=20, num_dla_samples=S,
                                 test_set_name="dr12q")
    with hdf5.File(p) as f:
        assert f["sample_log_likelihoods_dla"].shape == (md, S, nq)
        np.testing.assert_array_equal(f["sample_log_likelihoods_dla"][1, :, 3], res["sample_log_likelihoods_dla"][3, 1])
        assert f["base_sample_inds"].shape == (md - 1, S, nq) and f["base_sample_inds"].dtype == np.dtype("<u4")
        assert f["base_sample_inds"].attrs["MATLAB_class"] == "uint32"
        np.testing.assert_array_equal(f["MAP_log_nhis"][()].T, res["MAP_log_nhis"])
        np.testing.assert_array_equal(f["model_posteriors"][()].T, res["model_posteriors"])
        np.testing.assert_array_equal(f["p_lls"][0, :], res["p_lls"])
    back = io.load_processed_qsos(p)
    for k in ("sample_log_likelihoods_dla", "base_sample_inds", "MAP_inds", "log_likelihoods_dla", "all_exceptions"):
        np.testing.assert_array_equal(back[k], res[k])


def test_chunked_dataset_stream(tmp_path):
    """hdf5.FileWriter.open_chunked_dataset: chunks written in any order between other datasets,
    ragged edges zero-filled, index and header written by close(); misuse is refused."""
    from gp_dla_detection_amd import hdf5
    a = np.arange(23 * 37, dtype=np.float64).reshape(23, 37)
    p = str(tmp_path / "c.h5")
    w = hdf5.FileWriter(p)
    st = w.open_chunked_dataset("x", a.shape, np.float64, (5, 8))
    offs = [(i, j) for i in range(0, 23, 5) for j in range(0, 37, 8)]
    np.random.default_rng(1).shuffle(offs)
    for n, (i, j) in enumerate(offs):
        st.write_chunk((i, j), a[i:i + 5, j:j + 8])
        if n == 3:
            w.create_dataset("y", np.arange(5.0))
    with pytest.raises(hdf5.HDF5Error):
        st.write_chunk((0, 0), a[:5, :8])      # twice
    with pytest.raises(hdf5.HDF5Error):
        st.write_chunk((1, 0), a[:5, :8])      # off the grid
    st.close()
    w.close()
    with hdf5.File(p) as f:
        np.testing.assert_array_equal(f["x"].read(), a)
        np.testing.assert_array_equal(f["y"].read(), np.arange(5.0))
    # more chunks than one index node holds (64): a two-level B-tree, chunks arriving column-major
    b = np.random.default_rng(0).standard_normal((97, 211))
    w = hdf5.FileWriter(str(tmp_path / "e.h5"))
    st = w.open_chunked_dataset("x", b.shape, np.float64, (4, 3))
    for j in range(0, 211, 3):
        for i in range(0, 97, 4):
            st.write_chunk((i, j), b[i:i + 4, j:j + 3])
    st.close()
    w.close()
    with hdf5.File(str(tmp_path / "e.h5")) as f:
        np.testing.assert_array_equal(f["x"].read(), b)
    w = hdf5.FileWriter(str(tmp_path / "d.h5"))
    st = w.open_chunked_dataset("x", (4, 4), np.float64, (2, 2))
    st.write_chunk((0, 0), np.zeros((2, 2)))
    with pytest.raises(hdf5.HDF5Error):
        st.close()                              # three chunks missing
    w.close()


def test_native_cell_reader_equals_the_python_reader(tmp_path, monkeypatch):
    """PreloadedReader.read_csr (csrc/h5cells.c: object headers, chunk B-trees, zlib, threads) against
    the pure-Python reader: deflate-compressed cells as MATLAB writes them, contiguous cells, an
    empty cell and a non-vector cell (both outside the native subset: Python fallback), any index
    order; and the whole thing again with the native library unavailable."""
    from gp_dla_detection_amd import api
    rng = np.random.default_rng(2)
    n = 37
    lens = rng.integers(1, 1500, n)   # >= 512 doubles: written chunked + deflate when compress=True
    lens[5] = 0                       # an empty cell (MATLAB_empty: rank-1 dimension list)
    lens[9] = 150000                  # several 1 MiB chunks behind one B-tree node
    cells = {k: [rng.standard_normal((m, 1)) for m in lens] for k in ("all_wavelengths", "all_flux", "all_noise_variance")}
    cells["all_flux"][3][::7] = np.nan
    cells["all_pixel_mask"] = [rng.uniform(size=(m, 1)) < 0.2 for m in lens]
    z = rng.uniform(2, 5, n)
    idx = np.concatenate([[5, 9], rng.permutation(n)[:27]])
    for compress in (True, False):
        p = str(tmp_path / f"pre{int(compress)}.mat")
        io.savemat73(p, cells, compress=compress)
        for native in (True, False):
            if not native:
                monkeypatch.setattr(io, "_h5cells", False)
            with io.PreloadedReader(p) as r:
                assert (r._native() is not None) == native
                want = api.spectra_to_csr(r.read(idx, z))
                got = r.read_csr(idx, z)
                np.testing.assert_array_equal(r.pixel_counts(idx), lens[idx])
            assert sorted(got) == sorted(want)
            for k in want:
                a, b = np.asarray(want[k]), np.asarray(got[k])
                assert a.dtype == b.dtype and a.shape == b.shape, k
                np.testing.assert_array_equal(a, b, err_msg=k)
            monkeypatch.setattr(io, "_h5cells", None)


def test_integer_cells_are_converted_not_reinterpreted(tmp_path):
    """A cell stored as 8-byte integers (a spectrum someone saved as int64) must reach the float64
    arrays CONVERTED: the native reader copies bytes and therefore refuses a cell whose stored class
    is not the array's (csrc/h5cells.c), the Python reader takes it and converts."""
    rng = np.random.default_rng(3)
    lens = [40, 600, 25]
    cells = {k: [rng.standard_normal((m, 1)) for m in lens] for k in ("all_wavelengths", "all_noise_variance")}
    cells["all_flux"] = [rng.standard_normal((lens[0], 1)), rng.integers(-5, 5, (lens[1], 1)).astype(np.int64),
                         rng.standard_normal((lens[2], 1))]
    cells["all_pixel_mask"] = [rng.uniform(size=(m, 1)) < 0.2 for m in lens]
    p = str(tmp_path / "ints.mat")
    io.savemat73(p, cells)
    with io.PreloadedReader(p) as r:
        got = r.read_csr(np.arange(3), np.array([2.5, 3.0, 3.5]))
    flux = np.concatenate([c.reshape(-1).astype(np.float64) for c in cells["all_flux"]])
    assert got["flux"].dtype == np.float64
    np.testing.assert_array_equal(got["flux"], flux)


def test_streamed_processed_writer_refuses_misuse(tmp_path):
    """io.ProcessedStreamWriter: batches must sit on the chunk grid (any multiple of it, the last one
    ragged), may arrive in any order, and finish() refuses an incomplete run."""
    from gp_dla_detection_amd.api import Batch
    nq, S, B = 21, 12, 4
    res = Batch.empty_results(nq, S)
    res["sample_log_likelihoods_dla"][...] = np.arange(nq * S, dtype=np.float64).reshape(nq, S)
    for k, v in res.items():
        if isinstance(v, np.ndarray) and v.dtype == np.float64 and k != "sample_log_likelihoods_dla":
            v[...] = 1.0
    res.update(num_lines=3, prior_z_qso_increase=0.1, max_z_cut=0.1)
    tab = lambda lo, hi: {"sample_log_likelihoods_dla": res["sample_log_likelihoods_dla"][lo:hi]}  # noqa: E731
    w = io.ProcessedStreamWriter(str(tmp_path / "a.mat"), nq, S, B)
    with pytest.raises(ValueError):
        w.append(2, tab(2, 6))          # not on the grid
    with pytest.raises(ValueError):
        w.append(0, tab(0, 6))          # not a multiple of the grid (and not the last batch)
    w.append(12, tab(12, 21))           # two chunk columns and the ragged last one, out of order
    w.append(0, tab(0, 12))             # three chunk columns
    w.finish(res, test_set_name="t")
    got = io.load_processed_qsos(str(tmp_path / "a.mat"))
    np.testing.assert_array_equal(got["sample_log_likelihoods_dla"], res["sample_log_likelihoods_dla"])
    w = io.ProcessedStreamWriter(str(tmp_path / "b.mat"), nq, S, B)
    w.append(0, tab(0, 8))
    with pytest.raises(ValueError):
        w.finish(res, test_set_name="t")   # 13 quasars never arrived


def test_combine_streams_3d_tables_by_model_and_sample_range(tmp_path, monkeypatch):
    from gp_dla_detection_amd import hdf5
    """combine_processed_chunks on multi-DLA chunks: the [max_dlas, S, nq] table is read and written in
    (model, range of samples) pieces -- never a whole 13-GB model row -- and the result equals the
    concatenation along the quasar axis.  The slab budget is shrunk so that the small table here
    really goes through several pieces per model."""
    rng = np.random.default_rng(5)
    md, S = 3, 37
    paths, parts = [], []
    for c, nq in enumerate((5, 9, 2)):
        sll = rng.standard_normal((md, S, nq))
        parts.append(sll)
        p = str(tmp_path / f"chunk{c}.mat")
        w = hdf5.FileWriter(p, userblock=io.matlab_userblock())
        w.create_dataset("p_dlas", rng.uniform(size=(1, nq)))
        w.create_dataset("sample_log_likelihoods_dla", sll, chunks=(1, 8, nq))       # chunked, as the streamed writer leaves it
        w.create_dataset("sample_log_likelihoods_lls", rng.standard_normal((S, nq)))  # contiguous 2-D
        w.create_dataset("test_ind", np.zeros((1, 16), dtype=np.uint8))
        w.close()
        paths.append(p)
    calls = []
    real = hdf5.Dataset.read_slab

    def spy(self, lo, hi, axis1=None):
        out = real(self, lo, hi, axis1)
        calls.append((self.shape, out.nbytes))
        return out
    monkeypatch.setattr(hdf5.Dataset, "read_slab", spy)
    monkeypatch.setattr(io, "COMBINE_SLAB_BYTES", 2048)
    out = str(tmp_path / "combined.mat")
    io.combine_processed_chunks(paths, out)
    with hdf5.File(out) as f:
        np.testing.assert_array_equal(f["sample_log_likelihoods_dla"].read(), np.concatenate(parts, axis=-1))
        assert f["sample_log_likelihoods_lls"].shape == (S, 16)
    three_d = [n for shape, n in calls if len(shape) == 3]
    assert len(three_d) > 3 * md and max(three_d) <= 2048 + 8 * 9 * 8  # pieces of (model, a few sample rows), never a model row
