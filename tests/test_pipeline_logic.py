"""api.run_pipeline on CPU with a stand-in context: the three-stage schedule (upload i+1 / sweep i /
download i-1 over re-filled batch slots), its ordering guarantees and its error paths need no GPU."""
import threading
import time

import numpy as np
import pytest

from gp_dla_detection_amd import api


class FakeBatch:
    def __init__(self, log, slot):
        self.log, self.slot, self.block, self.closed = log, slot, None, False

    def reload(self, block):
        assert not self.closed
        self.log.append(("reload", self.slot, block))
        self.block = block

    def close(self):
        self.closed = True
        self.log.append(("close", self.slot))


class FakeCtx:
    def __init__(self, log):
        self.log, self.made = log, 0

    def upload(self, block):
        b = FakeBatch(self.log, self.made)
        self.made += 1
        b.block = block
        self.log.append(("upload", b.slot, block))
        return b


@pytest.mark.parametrize("slots", [1, 2, 3, 5])
def test_schedule_and_slot_reuse(slots):
    log, lock = [], threading.Lock()
    n = 7
    downloaded = []

    def inputs(i):
        return (i,)

    def process(i, batch):
        assert batch.block == i  # the slot holds THIS block when its sweep is launched
        with lock:
            log.append(("process", batch.slot, i))

    def download(i, batch):
        time.sleep(0.002)
        assert batch.block == i  # not yet re-filled: the slot is re-used only after its download
        downloaded.append(i)

    ctx = FakeCtx(log)
    api.run_pipeline(ctx, n, inputs, process, download, slots)
    assert downloaded == list(range(n))                       # downloads in order
    assert [e[2] for e in log if e[0] == "process"] == list(range(n))   # sweeps in order
    made = min(slots, n)
    assert ctx.made == made and sorted(e[1] for e in log if e[0] == "close") == list(range(made))
    for e in log:  # block i lives in slot i % slots
        if e[0] in ("upload", "reload"):
            assert e[1] == e[2] % made


def test_warm_runs_on_the_download_thread_before_the_first_download():
    order = []
    api.run_pipeline(FakeCtx([]), 3, lambda i: (i,), lambda i, b: None,
                     lambda i, b: order.append(("download", i, threading.current_thread().name)), 2,
                     warm=lambda: order.append(("warm", threading.current_thread().name)))
    assert order[0][0] == "warm" and order[0][1] == order[1][2]
    assert [o[1] for o in order[1:]] == [0, 1, 2]


@pytest.mark.parametrize("where", ["inputs", "process", "download"])
def test_an_error_in_any_stage_surfaces_and_closes_the_slots(where):
    log = []

    def boom(stage, i):
        if stage == where and i == 2:
            raise ValueError(f"{stage} {i}")

    def inputs(i):
        boom("inputs", i)
        return (i,)

    ctx = FakeCtx(log)
    with pytest.raises(ValueError, match=f"{where} 2"):
        api.run_pipeline(ctx, 6, inputs, lambda i, b: boom("process", i), lambda i, b: boom("download", i), 2)
    assert sum(1 for e in log if e[0] == "close") == ctx.made  # every slot that was made is closed


def test_blocks_and_prefault():
    assert api.batch_blocks(10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert api.batch_blocks(0, 4) == [] and api.batch_blocks(3, 100) == [(0, 3)]
    a = np.empty((7, 1000))
    api.prefault(a)   # touches, never fails on odd sizes
    assert a.reshape(-1)[0] == 0
