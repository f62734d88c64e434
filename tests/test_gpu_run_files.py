"""BASELINE config 3's pipeline on what one GPU can prove: the file-to-file sharded run
(gp_dla_detection_amd/run_dr12q.py) at world 1 and at world 2 (two processes sharing cuda:0 over
gloo) -- -v7.3 inputs, blocks balanced by pixel count, bounded double-buffered batches, one chunk
file per rank, the gathered posterior table -- against the unsharded in-memory run, bit for bit.
The chunk files are the ones the reference's own mat_combine recombines
(tests/golden/make_consumer_fixtures.py, run in the build container where h5py exists)."""
import glob
import multiprocessing as mp
import os

import numpy as np
import pytest

import gp_dla_detection_amd as gp
import sharded_worker
from gp_dla_detection_amd import io, synthetic
from gp_dla_detection_amd.parameters import MultiParameters
from test_gpu_sharded import free_port

pytestmark = pytest.mark.gpu

NQ, S = 14, 96


@pytest.fixture(scope="module")
def fileset(tmp_path_factory):
    return synthetic.write_file_set(str(tmp_path_factory.mktemp("run_in")), num_quasars=NQ, num_samples=S)


def reference_run(fs, multi):
    sel = np.flatnonzero(fs["test_ind"])
    spectra = [fs["spectra"][i] for i in sel]
    z = fs["catalog"]["z_qsos"][sel]
    if multi:
        p = MultiParameters(max_dlas=3)
        lp = gp.dla_existence_prior_multi(fs["prior"]["z_qsos"], fs["prior"]["dla_ind"], z, fs["Z_lls"], fs["Z_dla"], p)
        return gp.process_qsos_multiple_dlas_meanflux(fs["model"], fs["samples"], spectra, lp, params=p), sel
    return gp.process_qsos(fs["model"], fs["samples"], spectra, prior_catalog=fs["prior"]), sel


def run_world(world, multi, in_dir, out_dir, per_batch):
    if world == 1:
        sharded_worker.run_files_rank(0, 1, 0, multi, in_dir, str(out_dir), per_batch)
    else:
        ctx = mp.get_context("forkserver")
        port = free_port()
        procs = [ctx.Process(target=sharded_worker.run_files_rank,
                             args=(r, world, port, multi, in_dir, str(out_dir), per_batch)) for r in range(world)]
        for pr in procs:
            pr.start()
        for pr in procs:
            pr.join(900)
        for pr in procs:
            if pr.is_alive():
                pr.kill()
                pr.join()
        assert [pr.exitcode for pr in procs] == [0] * world
    kind = "multi" if multi else "single"
    return [np.load(out_dir / f"files_{kind}_w{world}_r{r}.npz") for r in range(world)]


@pytest.mark.parametrize("world,multi,per_batch", [(1, False, 4), (2, False, 3), (2, True, 2), (1, True, None)])
def test_file_to_file_run_equals_the_unsharded_run(fileset, tmp_path, world, multi, per_batch):
    fs = fileset
    in_dir = os.path.dirname(fs["paths"]["catalog"])
    ranks = run_world(world, multi, in_dir, tmp_path, per_batch)
    ref, sel = reference_run(fs, multi)
    nsel = sel.size
    # every rank holds the whole run's posterior table
    for r in ranks:
        np.testing.assert_array_equal(r["selected"], sel)
        for name in ("p_dlas", "p_no_dlas", "model_posteriors", "log_likelihoods_dla", "log_likelihoods_no_dla",
                     "min_z_dlas", "max_z_dlas", "MAP_z_dlas", "MAP_log_nhis", "MAP_inds"):
            np.testing.assert_array_equal(r["f_" + name], ref[name], err_msg=name)
    blocks = [tuple(int(x) for x in r["block"]) for r in ranks]
    assert blocks[0][0] == 0 and blocks[-1][1] == nsel
    assert all(blocks[i][1] == blocks[i + 1][0] for i in range(len(blocks) - 1))
    # the chunk files, in name order, are the run
    stem = "processed_qsos_multi_meanfluxsynth_" if multi else "processed_qsos_synth_"
    chunks = sorted(glob.glob(str(tmp_path / (stem + "[0-9]*.mat"))))
    assert chunks == [str(r["chunk"]) for r in ranks if str(r["chunk"])]
    lo = 0
    for path in chunks:
        part = io.load_processed_qsos(path)
        n = part["p_dlas"].size
        mask = np.zeros(NQ, dtype=bool)
        mask[sel[lo:lo + n]] = True
        np.testing.assert_array_equal(np.asarray(part["test_ind"]).reshape(-1).astype(bool), mask)
        for name in ("sample_log_likelihoods_dla", "p_dlas", "model_posteriors") + (
                ("sample_log_likelihoods_lls", "base_sample_inds", "MAP_z_dlas") if multi else ()):
            np.testing.assert_array_equal(part[name], ref[name][lo:lo + n], err_msg=name)
        lo += n
    assert lo == nsel
    combined = str(tmp_path / "combined.mat")
    io.combine_processed_chunks(chunks, combined)
    whole = io.load_processed_qsos(combined)
    np.testing.assert_array_equal(np.asarray(whole["test_ind"]).reshape(-1).astype(bool), fs["test_ind"])
    for name in ("sample_log_likelihoods_dla", "log_posteriors_dla", "model_posteriors", "p_dlas"):
        np.testing.assert_array_equal(whole[name], ref[name], err_msg=name)
    # the fully masked quasar was skipped, not dropped
    skipped = np.flatnonzero(np.isnan(ref["p_dlas"]))
    assert skipped.size == 1 and sel[skipped[0]] == 7
    summary = io.loadmat73(str(tmp_path / (stem + "summary.mat")))
    np.testing.assert_array_equal(summary["p_dlas"].reshape(-1), ref["p_dlas"])


def test_a_failing_rank_ends_the_file_run_on_every_rank(fileset, tmp_path):
    """World 2 on the GPU, rank 1's input reader raises at its second batch: rank 1 re-raises its
    error and removes its partial chunk, rank 0 -- whose chunk is complete and stays -- raises
    ShardFailure instead of waiting in the all-gather; no summary file is written."""
    in_dir = os.path.dirname(fileset["paths"]["catalog"])
    ctx = mp.get_context("forkserver")
    port = free_port()
    procs = [ctx.Process(target=sharded_worker.run_files_rank, args=(r, 2, port, False, in_dir, str(tmp_path), 2, 1))
             for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(600)
    for pr in procs:
        if pr.is_alive():
            pr.kill()
            pr.join()
    assert [pr.exitcode for pr in procs] == [0, 0]  # (the worker records the exception and exits)
    ended = [open(tmp_path / f"ended_r{r}.txt").read().split("\n") for r in range(2)]
    assert ended[0][0] == "ShardFailure" and ended[1][0] == "OSError"
    assert all(float(e[1]) < 120.0 for e in ended)
    chunks = sorted(glob.glob(str(tmp_path / "processed_qsos_synth_[0-9]*.mat")))
    assert len(chunks) == 1 and "_000000-" in chunks[0]
    io.load_processed_qsos(chunks[0])  # complete and readable
    assert not glob.glob(str(tmp_path / "*summary.mat"))


def test_committed_consumer_chunks_are_reproduced(tmp_path):
    """The chunk files under tests/golden/consumer/ -- the ones the reference's mat_combine, QSOLoader
    and DLACatalogue were run on (tests/test_consumers.py) -- are what a world-2 run on this GPU
    writes: same quasars per chunk, values within the parity tolerance (they were produced by an
    earlier build of the kernels)."""
    cons = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "consumer")
    fs = synthetic.write_file_set(str(tmp_path / "in"), num_quasars=40, num_samples=24, empty_quasar=None)
    in_dir = os.path.dirname(fs["paths"]["catalog"])
    reevaluated = np.load(os.path.join(cons, "expected_reevaluated_posteriors_multi.npz"))["model_posteriors"]
    for multi in (False, True):
        out = tmp_path / ("multi" if multi else "single")
        out.mkdir()
        run_world(2, multi, in_dir, out, 4)
        stem = "processed_qsos_multi_meanfluxsynth_" if multi else "processed_qsos_synth_"
        new = sorted(glob.glob(str(out / (stem + "[0-9]*.mat"))))
        old = sorted(glob.glob(os.path.join(cons, stem + "[0-9]*.mat")))
        assert [os.path.basename(p) for p in new] == [os.path.basename(p) for p in old] and len(old) == 2
        if multi:  # k_multi_posteriors of THIS run against the reference's own softmax (qso_loader.py:260-283)
            now = np.concatenate([io.load_processed_qsos(p)["model_posteriors"] for p in new], axis=0)
            np.testing.assert_allclose(now, reevaluated, rtol=0, atol=1e-8)
        for a, b in zip(new, old):
            pa, pb = io.load_processed_qsos(a), io.load_processed_qsos(b)
            assert sorted(pa) == sorted(pb)
            for key, vb in pb.items():
                va = pa[key]
                if isinstance(vb, str):
                    assert va == vb
                elif key == "base_sample_inds":
                    assert np.mean(np.asarray(va) == np.asarray(vb)) > 0.999
                elif "sample_log_likelihoods" in key or key.startswith(("log_likelihoods", "log_posteriors")):
                    np.testing.assert_allclose(va, vb, rtol=0, atol=1e-8, err_msg=key)
                else:
                    np.testing.assert_allclose(np.asarray(va, dtype=np.float64), np.asarray(vb, dtype=np.float64),
                                               rtol=1e-9, atol=1e-9, err_msg=key)


def test_command_line_entry_point(fileset, tmp_path):
    """`python -m gp_dla_detection_amd.run_dr12q --preloaded ... --out ...` (world 1): the chunk it writes
    is the unsharded run."""
    fs = fileset
    p = fs["paths"]
    argv = ["--preloaded", p["preloaded"], "--catalog", p["catalog"], "--learned", p["learned"], "--samples",
            p["samples"], "--prior", p["prior"], "--out", str(tmp_path), "--name", "cli", "--batch", "5"]
    ctx = mp.get_context("forkserver")
    pr = ctx.Process(target=sharded_worker.run_cli, args=(argv,))
    pr.start()
    pr.join(900)
    if pr.is_alive():
        pr.kill()
        pr.join()
    assert pr.exitcode == 0
    ref, sel = reference_run(fs, False)
    chunks = sorted(glob.glob(str(tmp_path / "processed_qsos_cli_[0-9]*.mat")))
    assert len(chunks) == 1 and chunks[0].endswith(f"_000000-{sel.size:06d}.mat")
    part = io.load_processed_qsos(chunks[0])
    np.testing.assert_array_equal(part["sample_log_likelihoods_dla"], ref["sample_log_likelihoods_dla"])
    np.testing.assert_array_equal(part["p_dlas"], ref["p_dlas"])
    assert part["test_set_name"] == "cli"
