"""The multi-GPU path on CPU: shard balancing, and the summary all-gather under gloo with
world_size 2 (the same code path RCCL runs on the GPUs; only the backend differs)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gp_dla_detection_amd.distributed import (SUMMARY_COLUMNS, gather_summaries, shard_bounds,
                                              summary_to_fields)


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
    t = np.arange(24, dtype=np.float64).reshape(2, 12)
    f = summary_to_fields(t)
    assert f["model_posteriors"].shape == (2, 2)
    assert f["min_z_dlas"][1] == 12 and f["p_dlas"][0] == 11
    assert len(SUMMARY_COLUMNS) == 12


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, counts, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo = sum(counts[:rank])
        local = torch.arange(lo * 12, (lo + counts[rank]) * 12, dtype=torch.float64).reshape(-1, 12)
        full = gather_summaries(local, counts)          # sizes known
        full2 = gather_summaries(local)                 # sizes exchanged first
        np.save(os.path.join(out_dir, f"r{rank}.npy"), full.numpy())
        assert torch.equal(full, full2)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("counts", [[3, 3], [5, 2]])
def test_gather_summaries_gloo_world2(tmp_path, counts):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, counts, str(tmp_path)), nprocs=world, join=True)
    total = sum(counts)
    want = np.arange(total * 12, dtype=np.float64).reshape(total, 12)
    for r in range(world):  # every rank ends up with the full table, in quasar order
        np.testing.assert_array_equal(np.load(tmp_path / f"r{r}.npy"), want)
