#!/usr/bin/env python3
"""GPU box: produce the chunk files of a world-2 file-to-file run on the synthetic file set, single-
and multi-DLA, for tests/golden/make_consumer_fixtures.py (which hands them to the reference's own
mat_combine / QSOLoader / DLACatalogue in the build container, where h5py exists).

  python tools/make_consumer_chunks.py gpurun_out/consumer

The two ranks are spawned before this process touches the GPU; they share cuda:0 and talk over gloo
(tests/sharded_worker.run_files_rank, the code path of tests/test_gpu_run_files.py)."""
import multiprocessing as mp
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

NQ, S = 40, 24  # more searched quasars (32) than samples: calc_cddf.py:940 indexes quasars by sample index


def main():
    out = os.path.abspath(sys.argv[1])
    from gp_dla_detection_amd import synthetic
    in_dir = os.path.join(out, "in")
    synthetic.write_file_set(in_dir, num_quasars=NQ, num_samples=S, empty_quasar=None)  # (the reference asserts on NaN rows)
    import sharded_worker
    ctx = mp.get_context("spawn")
    for multi in (False, True):
        d = os.path.join(out, "multi" if multi else "single")
        os.makedirs(d, exist_ok=True)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = [ctx.Process(target=sharded_worker.run_files_rank, args=(r, 2, port, multi, in_dir, d, 4))
                 for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(900)
        codes = [p.exitcode for p in procs]
        print("multi" if multi else "single", "exit codes", codes, sorted(os.listdir(d)), flush=True)
        if codes != [0, 0]:
            return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
