#!/usr/bin/env python3
"""Diagnostic (1-GPU box): can RCCL form a communicator of TWO ranks that share cuda:0?  (It is the only
way to exercise the N > 1 all-gather over RCCL without a multi-GPU node.)  Each rank is a child process
started before the parent touches the GPU; prints what happened and never hangs longer than the timeout
of the caller."""
import multiprocessing as mp
import os
import socket
import sys


def rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        x = torch.full((4,), float(rank + 1), device="cuda:0", dtype=torch.float64)
        out = torch.empty((world * 4,), device="cuda:0", dtype=torch.float64)
        dist.all_gather_into_tensor(out, x)
        torch.cuda.synchronize()
        q.put((rank, "ok", out.cpu().tolist()))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put((rank, "error", repr(e)[:400]))


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    q = ctx.Queue()
    procs = [ctx.Process(target=rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(90)
    for p in procs:
        if p.is_alive():
            p.kill()
            print("rank still running after 90 s: killed")
    while not q.empty():
        print(q.get())
    sys.exit(0)
