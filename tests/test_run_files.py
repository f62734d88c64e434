"""Host pieces of the file-to-file sharded run (gp_dla_detection_amd/run_dr12q.py) that need no GPU:
the synthetic file set, header-only pixel counts, chunk naming, and the streamed recombination of
chunk files -- the reference's mat_combine (CDDF_analysis/sbatch_reunion.py:13-63) restated."""
import numpy as np
import pytest

from gp_dla_detection_amd import hdf5, io, run_dr12q, synthetic
from gp_dla_detection_amd.api import Batch, default_batch_size, record_bytes_per_quasar


@pytest.fixture(scope="module")
def fileset(tmp_path_factory):
    return synthetic.write_file_set(str(tmp_path_factory.mktemp("fileset")), num_quasars=12, num_samples=64)


def test_file_set_reads_back(fileset):
    fs = fileset
    cat = io.load_catalog(fs["paths"]["catalog"])
    np.testing.assert_array_equal(cat["z_qsos"], fs["catalog"]["z_qsos"])
    sel = run_dr12q.select_test_ind(cat)
    np.testing.assert_array_equal(sel, np.flatnonzero(fs["test_ind"]))
    np.testing.assert_array_equal(run_dr12q.select_test_ind(cat, fs["test_ind"]), sel)
    np.testing.assert_array_equal(run_dr12q.select_test_ind(cat, [3, 1]), [3, 1])
    with io.PreloadedReader(fs["paths"]["preloaded"]) as r:
        assert r.num_quasars == 12
        counts = r.pixel_counts(sel)
        np.testing.assert_array_equal(counts, [fs["spectra"][i]["wavelengths"].size for i in sel])
        got = r.read(sel[2:5], cat["z_qsos"])
    for g, i in zip(got, sel[2:5]):
        s = fs["spectra"][i]
        np.testing.assert_array_equal(g["wavelengths"], s["wavelengths"])
        np.testing.assert_array_equal(g["flux"], s["flux"])
        np.testing.assert_array_equal(g["pixel_mask"], s["pixel_mask"])
        assert g["z_qso"] == s["z_qso"]
    m = io.load_learned_model(fs["paths"]["learned"])
    np.testing.assert_array_equal(m["M"], fs["model"]["M"])
    smp = io.load_dla_samples(fs["paths"]["samples"])
    np.testing.assert_array_equal(smp["lls_nhi_samples"], fs["samples"]["lls_nhi_samples"])


def fake_results(nq, S, md, seed):
    rng = np.random.default_rng(seed)
    if md:
        out = Batch.empty_results_multi(nq, md, S)
    else:
        out = Batch.empty_results(nq, S)
    for k, v in out.items():
        if v.dtype == np.float64:
            v[...] = rng.normal(size=v.shape)
        elif v.dtype == np.uint32:
            v[...] = rng.integers(1, S + 1, size=v.shape)
    mp = rng.uniform(size=out["model_posteriors"].shape)
    out["model_posteriors"][...] = mp / mp.sum(axis=1, keepdims=True)
    return out


@pytest.mark.parametrize("md", [0, 3])
def test_chunks_recombine_to_the_whole_run(tmp_path, md):
    """Three chunk files (5 + 1 + 6 quasars) -> one file equal to the run saved whole."""
    S, n_all = 16, 20
    sizes = [5, 1, 6]
    sel = np.array([0, 2, 3, 4, 6, 7, 9, 10, 12, 13, 15, 19])
    parts = [fake_results(n, S, md, 10 + i) for i, n in enumerate(sizes)]
    save = io.save_processed_qsos_multi if md else io.save_processed_qsos
    paths, lo = [], 0
    for part, n in zip(parts, sizes):
        mask = np.zeros(n_all, dtype=bool)
        mask[sel[lo:lo + n]] = True
        paths.append(io.chunk_filename(str(tmp_path), "dr12q", lo, lo + n, multi=bool(md)))
        extra = dict(k=20, num_dla_samples=S) if md else {}
        save(paths[-1], dict(part, **extra), test_ind=mask, test_set_name="dr12q", release="dr12q")
        lo += n
    assert paths == sorted(paths)  # sorting the names orders the chunks
    assert paths[0].endswith(("processed_qsos_multi_meanfluxdr12q_000000-000005.mat" if md
                              else "processed_qsos_dr12q_000000-000005.mat"))
    out = str(tmp_path / "combined.mat")
    io.combine_processed_chunks(paths, out)
    whole = {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}
    back = io.load_processed_qsos(out)
    for k, v in whole.items():
        if k in ("status",):
            continue
        np.testing.assert_array_equal(np.asarray(back[k]).reshape(v.shape), v, err_msg=k)
    mask = np.zeros(n_all, dtype=bool)
    mask[sel] = True
    np.testing.assert_array_equal(np.asarray(back["test_ind"]).reshape(-1).astype(bool), mask)
    assert back["test_set_name"] == "dr12q"
    with hdf5.File(out) as f:  # MATLAB's conventions survive: class attributes, reversed dimensions
        assert f["p_dlas"].attrs["MATLAB_class"] == "double" and f["p_dlas"].shape == (1, 12)
        assert f["test_ind"].attrs["MATLAB_class"] == "logical"
        if md:
            assert f["sample_log_likelihoods_dla"].shape == (md, S, 12)
            assert f["base_sample_inds"].attrs["MATLAB_class"] == "uint32"


def test_batch_size_defaults():
    assert 0.33e6 < record_bytes_per_quasar(1500, 20) < 0.4e6          # slim records: 896 B per K-step
    assert 2.8e6 < record_bytes_per_quasar(1500, 20, slim=False) < 3.0e6
    assert 0.57e6 < record_bytes_per_quasar(1500, 40) < 0.6e6          # k <= 40 slim records: 1536 B per K-step
    assert 11e6 < record_bytes_per_quasar(1500, 40, slim=False) < 11.3e6
    assert default_batch_size(2048, 1500, 20, 10000, 3) == 256
    assert default_batch_size(100, 1500, 20, 10000, 3) in (100, 128)
    assert default_batch_size(10 ** 6, 1500, 20, 10000, 3) == 4096
    # memory-bound case: a small HBM budget
    assert default_batch_size(10 ** 6, 1500, 40, 10000, 3, budget_bytes=2 ** 30) < 800


def test_ramped_blocks_cover_the_run_on_the_chunk_grid():
    """run_dr12q.ramped_blocks: contiguous cover, a small first and last batch, and every batch but
    the last on the chunk grid of the streamed writer."""
    from gp_dla_detection_amd.run_dr12q import ramped_blocks
    for n, b in ((20358, 4096), (4000, 4096), (5000, 4096), (9000, 4096), (100, 2), (33, 16), (1000, 128),
                 (4097, 4096), (5120, 4096), (130, 128), (1, 64), (512, 512), (513, 512)):
        blocks, grid = ramped_blocks(n, b)
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(a[1] == c[0] for a, c in zip(blocks, blocks[1:]))
        assert all(hi > lo for lo, hi in blocks)
        assert all(lo % grid == 0 for lo, _ in blocks) and all((hi - lo) % grid == 0 for lo, hi in blocks[:-1])
        assert max(hi - lo for lo, hi in blocks) <= b + grid
        if n > b and b // 8 >= 16:
            assert blocks[0][1] - blocks[0][0] == grid == b // 8 and blocks[-1][1] - blocks[-1][0] < 2 * grid


def test_shard_workload_helpers():
    """bench.py --workload dr12q-shard: blocks are balanced by pixel counts computed from the redshifts
    alone -- they must be the stored lengths of the spectra the generator then makes -- and a rank's
    block made by worker processes equals the serially made one (same seeds, same order)."""
    model = synthetic.make_model(20)
    z = synthetic.sample_dr12q_redshifts(300)
    spectra = synthetic.make_dr12q_mix(300, model)
    np.testing.assert_array_equal(synthetic.boss_pixel_counts(z), [s["wavelengths"].size for s in spectra])
    part = synthetic.make_dr12q_mix_parallel(40, 4200, 20, workers=2)
    assert len(part) == 4200
    for i in (0, 1, 4199):
        one = synthetic.make_dr12q_mix(1, model, first_index=40 + i)[0]
        for key in ("wavelengths", "flux", "noise_variance", "pixel_mask"):
            np.testing.assert_array_equal(part[i][key], one[key])
        assert part[i]["z_qso"] == one["z_qso"]
    runs = synthetic.make_dr12q_mix(50, model, mask_runs=True)  # the sky-line-style mask of the 7(b) experiment
    frac = np.mean([s["pixel_mask"].mean() for s in runs])
    assert 0.02 < frac < 0.08
    for a, b in zip(runs[:5], spectra[:5]):  # only the mask (and what it blanks) differs
        np.testing.assert_array_equal(a["wavelengths"], b["wavelengths"])
        keep = (a["pixel_mask"] == 0) & (b["pixel_mask"] == 0)
        np.testing.assert_array_equal(a["flux"][keep], b["flux"][keep])


def test_staging_buffer_without_a_gpu():
    """run_dr12q._staging: one batch's host arrays (page-locked when torch can; plain arrays here)."""
    for md in (0, 3):
        st = run_dr12q._staging(5, 16, md)
        assert st["sample_log_likelihoods_dla"].shape == ((5, md, 16) if md else (5, 16))
        assert st["sample_log_likelihoods_dla"].flags.c_contiguous and st["sample_log_likelihoods_dla"].flags.writeable
        if md:
            assert st["base_sample_inds"].dtype == np.uint32 and st["base_sample_inds"].shape == (5, md - 1, 16)
            assert st["sample_log_likelihoods_lls"].shape == (5, 16)
        st["sample_log_likelihoods_dla"][...] = 1.0


def test_streamed_writer_chunk_rows_divide_the_sample_axis(tmp_path):
    """Every chunk is stored whole, so a ragged last chunk row is padding on disk: the sample axis is
    split evenly (2545-quasar batches once made a 20 358-quasar chunk file 32 % padding)."""
    nq, S, grid = 40, 1000, 13
    w = io.ProcessedStreamWriter(str(tmp_path / "c.mat"), nq, S, grid)
    st = w.streams["sample_log_likelihoods_dla"][0]
    rows = st.chunks[-2]
    assert S % rows == 0 or (-(-S // rows)) * rows - S < -(-S // rows)  # at most one row of padding per chunk row
    w.abort()
    big = io.ProcessedStreamWriter(str(tmp_path / "d.mat"), 20358, 10000, 318)
    rows = big.streams["sample_log_likelihoods_dla"][0].chunks[-2]
    assert rows == 2500  # 8 MB / (8 B x 318) = 3297 wanted -> four equal rows instead of 3297 + 3297 + 3297 + 109
    big.abort()
