#!/usr/bin/env python3
"""File-to-file rate of the sharded run on one GPU (world 1): -v7.3 inputs (cells deflate-compressed,
as MATLAB writes them) -> gp_dla_detection_amd.run_dr12q.run -> the rank's chunk file.  Prints one
JSON line with the phases."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from gp_dla_detection_amd import run_dr12q, synthetic  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
batch = (int(sys.argv[3]) or None) if len(sys.argv) > 3 else None  # quasars per HBM-resident batch (0 / absent: run_dr12q's)
multi = len(sys.argv) > 4 and sys.argv[4] == "multi"
mk = dict(multi=True, Z_lls=0.31, Z_dla=0.69) if multi else {}
d = tempfile.mkdtemp(prefix="gpdla_files_")
t0 = time.perf_counter()
fs = synthetic.write_file_set(d, num_quasars=nq, num_samples=S, skip_every=10 ** 9, empty_quasar=None)
t_gen = time.perf_counter() - t0
pr = fs["prior"]
fs = dict(paths=fs["paths"])  # the generator's copy of the spectra must not count as the run's memory
import gc  # noqa: E402
import resource  # noqa: E402
gc.collect()


def rss_now_mb():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmRSS:"):
                return int(line.split()[1]) / 1024.0
    return float("nan")


run_dr12q.run(fs["paths"]["preloaded"], fs["paths"]["catalog"], fs["paths"]["learned"], fs["paths"]["samples"],
              d + "/warm", "warm", test_ind=np.arange(64), prior_catalog=pr, device=0, **mk)  # warm-up
rss_before = rss_now_mb()  # interpreter + torch + HIP runtime + the library, after a warm-up run
peak_before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0
# the high-water mark so far belongs to the input generator: sample the resident set during the run
import threading  # noqa: E402
samples_mb, stop = [rss_before], threading.Event()


def sampler():
    while not stop.wait(0.01):
        samples_mb.append(rss_now_mb())


watch = threading.Thread(target=sampler, daemon=True)
watch.start()
t0 = time.perf_counter()
res = run_dr12q.run(fs["paths"]["preloaded"], fs["paths"]["catalog"], fs["paths"]["learned"], fs["paths"]["samples"],
                    d + "/out", "synth", prior_catalog=pr, device=0, max_quasars_per_batch=batch, **mk)
t_run = time.perf_counter() - t0
stop.set()
watch.join()
size = os.path.getsize(res["chunk"])
print(json.dumps(dict(quasars=nq, samples=S, multi=multi, seconds=t_run, evals_per_s=nq * S * (5 if multi else 1) / t_run, quasars_per_s=nq / t_run,
                      chunk_bytes=size, input_bytes=os.path.getsize(fs["paths"]["preloaded"]),
                      generate_inputs_s=t_gen, batch=batch,
                      rss_mb=dict(before_run=rss_before, peak_during_run=max(samples_mb),
                                  run_adds=max(samples_mb) - rss_before, samples=len(samples_mb),
                                  ru_maxrss_before_run=peak_before,
                                  ru_maxrss=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0,
                                  note="VmRSS sampled every 10 ms during the run (MB); before_run = resident set after "
                                       "the warm-up run (interpreter, torch, HIP runtime, library); the input file is "
                                       "memory-mapped, its touched pages count; ru_maxrss includes the input generator"), timings=res["timings"], finite_p_dlas=int(np.isfinite(res["fields"]["p_dlas"]).sum()))))
