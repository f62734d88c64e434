#!/usr/bin/env python3
"""Times the training objective (objective.m over spectrum_loss.m, row N3) on one GPU against the
CPU oracle: a learn_qso_model-sized problem (G = 1217 rest pixels, k = 20).  Prints a JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_dla_detection_amd import training  # noqa: E402
from oracle import oracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--quasars", type=int, default=5000)
ap.add_argument("--pixels", type=int, default=1217)
ap.add_argument("--k", type=int, default=20)
ap.add_argument("--forest-lines", type=int, default=0, help="> 1: the mean-flux model's objective (objective_lyseries.m)")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--no-cpu", action="store_true", help="skip the CPU oracle leg (profiling passes)")
args = ap.parse_args()
rng = np.random.default_rng(0)
nq, G, k = args.quasars, args.pixels, args.k
M = rng.standard_normal((G, k)) * 0.3 * 0.8 ** np.arange(k)
x = np.concatenate([M.ravel(order="F"), rng.uniform(-3, -2, G), [np.log(0.1), np.log(0.0023), np.log(3.65)]])
L1 = 1 + rng.uniform(1.5, 3.0, (nq, G))
NV = 10 ** rng.uniform(-3, -1, (nq, G))
F = rng.standard_normal((nq, k)) @ M.T + np.sqrt(NV) * rng.standard_normal((nq, G))
F[rng.uniform(size=F.shape) < 0.1] = np.nan
t = training.TrainingSet(F, L1, NV)
if args.forest_lines > 1:
    t.set_lyseries(args.forest_lines)
t.objective(x)
t0 = time.perf_counter()
reps = args.reps
for _ in range(reps):
    f, g = t.objective(x)
gpu_s = (time.perf_counter() - t0) / reps
t.close()
if args.no_cpu:
    print(json.dumps({"quasars": nq, "pixels": G, "k": k, "gpu_seconds_per_eval": gpu_s, "forest_lines": args.forest_lines}))
    sys.exit(0)
# CPU oracle on the threads this job may really use: the cgroup quota, not the host's CPU count
# (a GPU box hands a 16-CPU slice of a 256-CPU host to a one-GPU job; 256 OpenMP threads on 16 CPUs
# measure oversubscription, not the oracle -- bench.py:cpu_baseline does the same)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import host_cpu_info  # noqa: E402
info = host_cpu_info()
threads = info["cores_physical"] or info["logical_cpus"]
if info["cpu_quota"]:
    threads = max(1, min(threads, int(info["cpu_quota"] + 1e-9)))
sub = min(nq, 64 * threads)
oracle.objective(x, F[:threads], L1[:threads], NV[:threads], num_threads=threads)  # warms the thread pool
rates = []
for _ in range(3):
    t0 = time.perf_counter()
    f_ref, g_ref = oracle.objective(x, F[:sub], L1[:sub], NV[:sub], num_threads=threads)
    rates.append(sub / (time.perf_counter() - t0))
cpu_rate = float(np.median(rates))
# algorithmic flops per quasar: B (n k^2) + K^-1 M (2 n k^2) + the rest O(n k)
flops = nq * (0.9 * G) * (3 * k * k + 10 * k) * 1.0
print(json.dumps({"metric": "training objective evaluations (value + gradient)", "quasars": nq, "pixels": G, "k": k,
                  "gpu_seconds_per_eval": gpu_s, "gpu_quasars_per_s": nq / gpu_s,
                  "cpu_oracle_quasars_per_s": cpu_rate, "cpu_threads": threads, "cpu_quota": info["cpu_quota"],
                  "cpu_sample": f"{sub} quasars, median of 3",
                  "gpu_gflops_algorithmic": flops / gpu_s / 1e9, "forest_lines": args.forest_lines}))
