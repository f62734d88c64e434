"""The multi-GPU path on CPU: shard balancing, and the summary all-gather under gloo with
world_size 2 (the same code path RCCL runs on the GPUs; only the backend differs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gp_dla_detection_amd.distributed import (SUMMARY_COLUMNS, gather_summaries, run_sharded,
                                              shard_bounds, summary_to_fields,
                                              summary_to_fields_multi)

NCOL = len(SUMMARY_COLUMNS)


def test_shard_bounds_cover_and_balance():
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        sizes = rng.integers(200, 1300, size=1000)
        b = shard_bounds(sizes, world)
        assert b[0][0] == 0 and b[-1][1] == sizes.size
        assert all(b[r][1] == b[r + 1][0] for r in range(world - 1))
        load = np.array([sizes[lo:hi].sum() for lo, hi in b])
        assert load.max() / load.mean() < 1.01  # balanced by pixel count, not by quasar count


def test_shard_bounds_edge_cases():
    assert shard_bounds([10, 10], 1) == [(0, 2)]
    b = shard_bounds([5, 5], 4)  # fewer quasars than ranks: contiguous, some blocks empty
    assert b[0][0] == 0 and b[-1][1] == 2 and all(lo <= hi for lo, hi in b)
    b = shard_bounds([1000, 1, 1, 1], 4)  # one huge quasar: every rank still gets one
    assert [hi - lo for lo, hi in b] == [1, 1, 1, 1]
    with pytest.raises(ValueError):
        shard_bounds([1, 2, 3], 0)


def test_summary_to_fields_names():
    from gp_dla_detection_amd import _lib
    assert NCOL == _lib.SUMMARY_COLS == 15
    t = np.arange(2 * NCOL, dtype=np.float64).reshape(2, NCOL)
    f = summary_to_fields(t)
    assert f["model_posteriors"].shape == (2, 2)
    assert f["min_z_dlas"][1] == NCOL and f["p_dlas"][0] == 11
    assert f["MAP_inds"][0] == 12 and f["MAP_log_nhis"][1] == NCOL + 14


def test_summary_to_fields_multi_layout():
    """Column order of GPDLA_SUMMARY_COLS_MULTI (include/gpdla.h): 78 columns for max_dlas = 4,
    SURVEY.md section 8(e)."""
    from gp_dla_detection_amd import _lib
    md = 4
    ncol = _lib.summary_cols_multi(md)
    assert ncol == 78
    t = np.arange(3 * ncol, dtype=np.float64).reshape(3, ncol)
    f = summary_to_fields_multi(t, md)
    assert f["min_z_dlas"][0] == 0 and f["log_priors_lls"][0] == 3
    assert list(f["log_priors_dla"][0]) == [4, 5, 6, 7]
    assert f["log_likelihoods_no_dla"][0] == 8 and list(f["log_likelihoods_dla"][0]) == [10, 11, 12, 13]
    assert f["model_posteriors"].shape == (3, 6) and f["model_posteriors"][0, 0] == 20
    assert f["p_dlas"][0] == 28
    assert f["MAP_z_dlas"].shape == (3, 4, 4) and f["MAP_z_dlas"][0, 0, 0] == 29
    assert f["MAP_inds"][0, 3, 3] == 76 and f["all_exceptions"][0] == 77
    assert f["all_exceptions"][2] == 3 * ncol - 1
    with pytest.raises(ValueError):
        summary_to_fields_multi(t, 3)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, counts, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo = sum(counts[:rank])
        local = torch.arange(lo * NCOL, (lo + counts[rank]) * NCOL, dtype=torch.float64).reshape(-1, NCOL)
        full = gather_summaries(local, counts)          # sizes known
        full2 = gather_summaries(local)                 # sizes exchanged first
        np.save(os.path.join(out_dir, f"r{rank}.npy"), full.numpy())
        assert torch.equal(full, full2)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("counts", [[3, 3], [5, 2], [4, 0]])
def test_gather_summaries_gloo_world2(tmp_path, counts):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, counts, str(tmp_path)), nprocs=world, join=True)
    total = sum(counts)
    want = np.arange(total * NCOL, dtype=np.float64).reshape(total, NCOL)
    for r in range(world):  # every rank ends up with the full table, in quasar order
        np.testing.assert_array_equal(np.load(tmp_path / f"r{r}.npy"), want)


def _sharded_worker(rank, world, port, sizes, out_dir):
    """The sharding skeleton of process_qsos_sharded with a stand-in for the GPU sweep: quasar q's
    summary row is q + column/100, so the gathered table shows who swept what."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bounds = shard_bounds(sizes, world)
        swept = []

        def sweep_block(lo, hi):
            swept.append((lo, hi))
            q = torch.arange(lo, hi, dtype=torch.float64)[:, None]
            return q + torch.arange(NCOL, dtype=torch.float64)[None, :] / 100, {"rows": hi - lo}

        table, block, local = run_sharded(bounds, rank, NCOL, sweep_block)
        assert block == bounds[rank]
        if block[1] > block[0]:
            assert swept == [block] and local == {"rows": block[1] - block[0]}
        else:  # empty shard: no sweep, but the collective was still reached
            assert swept == [] and local is None
        np.save(os.path.join(out_dir, f"s{rank}.npy"), table.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [[700, 300, 900, 500, 800], [640]])
def test_run_sharded_gloo_world2(tmp_path, sizes):
    """N > 1 path of both sharded drivers, including fewer quasars than ranks (one rank gets an
    empty block and must not deadlock the all-gather)."""
    world = 2
    mp.spawn(_sharded_worker, args=(world, _free_port(), sizes, str(tmp_path)), nprocs=world, join=True)
    nq = len(sizes)
    want = np.arange(nq, dtype=np.float64)[:, None] + np.arange(NCOL)[None, :] / 100
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"s{r}.npy"), want)


def _failing_worker(rank, world, port, out_dir):
    """One rank's sweep raises: no rank may be left waiting in the all-gather."""
    import datetime
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from gp_dla_detection_amd.distributed import ShardFailure
    t0 = time.perf_counter()
    outcome = "no exception"
    try:
        bounds = shard_bounds([500, 400, 600, 500], world)

        def sweep_block(lo, hi):
            if rank == 1:
                raise ValueError("rank 1 cannot read its block")
            return torch.zeros((hi - lo, NCOL), dtype=torch.float64), None

        try:
            run_sharded(bounds, rank, NCOL, sweep_block)
        except ShardFailure:
            outcome = "ShardFailure"
        except ValueError as e:
            outcome = f"ValueError: {e}"
        with open(os.path.join(out_dir, f"f{rank}.txt"), "w") as f:
            f.write(f"{outcome}\n{time.perf_counter() - t0}\n")
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_ends_the_sharded_run_on_every_rank(tmp_path):
    """distributed.agree_on_failure: the rank that raised re-raises its own exception, the other rank
    raises ShardFailure at once -- nobody waits for the collective timeout (120 s here)."""
    world = 2
    mp.spawn(_failing_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [open(tmp_path / f"f{r}.txt").read().split("\n") for r in range(world)]
    assert got[0][0] == "ShardFailure"
    assert got[1][0] == "ValueError: rank 1 cannot read its block"
    assert all(float(g[1]) < 30.0 for g in got)


def test_agree_on_failure_without_a_process_group():
    from gp_dla_detection_amd.distributed import agree_on_failure
    agree_on_failure(None)
    with pytest.raises(KeyError):
        agree_on_failure(KeyError("x"))
