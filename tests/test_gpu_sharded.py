"""The multi-GPU drivers on ONE GPU: world 1 in-process, and a world of 2 processes that share
cuda:0 and talk over gloo (an 8-GPU node is not available to tests).  The
gathered summary table must equal the unsharded run bit for bit, and each rank's per-sample shard
must equal the matching slice -- for the single-DLA driver (process_qsos.m) and for the multi-DLA
driver, whose Philox resampling stream is keyed by the global quasar index so that shards draw what
the whole run draws (the behaviour of CDDF_analysis/sbatch_reunion.py:29-55: chunks concatenate to
the full run)."""
import multiprocessing as mp
import socket

import numpy as np
import pytest

import gp_dla_detection_amd as gp
import sharded_worker

pytestmark = pytest.mark.gpu

SUMMARY_SINGLE = ("min_z_dlas", "max_z_dlas", "log_priors_no_dla", "log_priors_dla",
                  "log_likelihoods_no_dla", "log_likelihoods_dla", "log_posteriors_no_dla",
                  "log_posteriors_dla", "model_posteriors", "p_no_dlas", "p_dlas", "MAP_inds",
                  "MAP_z_dlas", "MAP_log_nhis")
SUMMARY_MULTI = ("min_z_dlas", "max_z_dlas", "log_priors_no_dla", "log_priors_lls", "log_priors_dla",
                 "log_likelihoods_no_dla", "log_likelihoods_lls", "log_likelihoods_dla",
                 "log_posteriors_no_dla", "log_posteriors_lls", "log_posteriors_dla",
                 "model_posteriors", "p_no_dlas", "p_lls", "p_dlas", "MAP_z_dlas", "MAP_log_nhis",
                 "MAP_inds")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def unsharded(kind, num_quasars):
    model, samples, spectra, lp, p = sharded_worker.build_case(kind)
    spectra = spectra[:num_quasars]
    lp = tuple(np.asarray(x)[:num_quasars] for x in lp)
    if kind == "single":
        return gp.process_qsos(model, samples, spectra, log_priors=lp)
    return gp.process_qsos_multiple_dlas_meanflux(model, samples, spectra, lp, params=p)


def run_world(world, kind, out_dir, num_quasars):
    ctx = mp.get_context("forkserver")  # clean children: the server predates any GPU use
    port = free_port()
    procs = [ctx.Process(target=sharded_worker.run_rank, args=(r, world, port, kind, str(out_dir), num_quasars))
             for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(600)
    for pr in procs:  # only ever our own children, by handle
        if pr.is_alive():
            pr.kill()
            pr.join()
    assert [pr.exitcode for pr in procs] == [0] * world
    return [np.load(out_dir / f"{kind}_w{world}_r{r}.npz") for r in range(world)]


def check(kind, ranks, ref, num_quasars):
    names = SUMMARY_SINGLE if kind == "single" else SUMMARY_MULTI
    covered = []
    for res in ranks:
        lo, hi = (int(x) for x in res["block"])
        covered.append((lo, hi))
        for name in names:  # EVERY rank holds the full gathered table
            np.testing.assert_array_equal(res["f_" + name], ref[name], err_msg=name)
        if kind == "multi":
            ae = res["f_all_exceptions"]
            np.testing.assert_array_equal(np.isnan(ae), ref["status"] == 0)
        if hi > lo:
            for name in ("sample_log_likelihoods_dla",) + (
                    ("sample_log_likelihoods_lls", "base_sample_inds") if kind == "multi" else ()):
                np.testing.assert_array_equal(res["l_" + name], ref[name][lo:hi], err_msg=name)
    assert covered[0][0] == 0 and covered[-1][1] == num_quasars
    assert all(covered[r][1] == covered[r + 1][0] for r in range(len(covered) - 1))


@pytest.mark.parametrize("kind", ["single", "multi"])
def test_world_1_in_process(kind, tmp_path):
    """process_qsos_sharded / its multi-DLA sibling with no process group: the whole list is this
    rank's block."""
    sharded_worker.run_rank(0, 1, 0, kind, str(tmp_path), 7)
    res = np.load(tmp_path / f"{kind}_w1_r0.npz")
    check(kind, [res], unsharded(kind, 7), 7)


@pytest.mark.parametrize("kind", ["single", "multi"])
def test_world_2_on_one_gpu_over_gloo(kind, tmp_path):
    ranks = run_world(2, kind, tmp_path, 7)
    check(kind, ranks, unsharded(kind, 7), 7)
    assert all(int(r["block"][1]) > int(r["block"][0]) for r in ranks)


def test_world_2_with_fewer_quasars_than_ranks(tmp_path):
    """One quasar, two ranks: rank 1's block is empty; it must still reach the all-gather."""
    ranks = run_world(2, "single", tmp_path, 1)
    check("single", ranks, unsharded("single", 1), 1)
    assert [tuple(int(x) for x in r["block"]) for r in ranks] == [(0, 1), (1, 1)]


def test_rccl_backend_gathers_the_library_owned_table(tmp_path):
    """The RCCL leg at world size 1 (see sharded_worker.run_rccl_world1): backend nccl initialises
    on the box and all-gathers the zero-copy summary tensor on the sweep's own stream."""
    ctx = mp.get_context("forkserver")
    pr = ctx.Process(target=sharded_worker.run_rccl_world1, args=(free_port(), str(tmp_path)))
    pr.start()
    pr.join(600)
    if pr.is_alive():
        pr.kill()
        pr.join()
    assert pr.exitcode == 0
    res = np.load(tmp_path / "rccl_w1.npz")
    assert str(res["backend"]) == "nccl"
    np.testing.assert_array_equal(res["gathered"], res["table"])
    np.testing.assert_array_equal(res["table"][:, 5], res["ll"])
    assert res["table"].shape == (7, 15)
