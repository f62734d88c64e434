#!/usr/bin/env python3
"""bench.py -- throughput of the GP/DLA inference sweep on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  N > 1 works both ways: typed as above, this process starts the N ranks itself (child processes,
  one per GPU, spawned BEFORE anything here touches the GPU; rendezvous on 127.0.0.1) and exits
  with their status; under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...
  bench.py --gpus N ...` (WORLD_SIZE already set) it is one of the ranks.

A "step" is one pass of the hot path (gpdla_batch_process: selection + interpolation, null
evidence, S-sample Voigt/low-rank sweep, evidence + posteriors + MAP) over one HBM-resident batch
of synthetic quasars, followed -- when N > 1 -- by the RCCL all-gather of the posterior table.

Workloads (--workload):
  configs1    (default, the headline) BASELINE.json configs[1]: 1000 synthetic spectra x n = 1500
              pixels x k = 20 x S = 10 000 DLA samples, fp64, per GPU.
  dr12q-mix   the real shape of a DR12Q run: every spectrum distinct, on the BOSS 1e-4-dex pixel
              grid (283 .. 1250 pixels in the modelled range, set by z_qso), 5 % of the pixels
              masked, z_qso spread like the catalogue's.  Same metric; the roofline numerator sums
              the per-quasar flops F(n_kept, k).
  dr12q-shard BASELINE.json configs[2] as a strong-scaling run: --total-spectra (default 162 861)
              distinct quasars of the dr12q-mix shape are cut into N blocks balanced by pixel count
              (distributed.shard_bounds); every rank holds its block resident in HBM (batches of at
              most 32 768 quasars), sweeps it and copies the posterior rows into its table, and the
              15-column table of the WHOLE run is all-gathered once per step.  "scaling": "strong".
Weak scaling (configs1, dr12q-mix): every rank sweeps its own --spectra quasars.
value = sample log-likelihood evaluations per second over all ranks.  Rank 0 prints ONE JSON line; it
carries every rank's kernel / gather / step milliseconds and set-up seconds (`per_rank`).  At N = 1
the configs1 line also carries riders measured AFTER the timed region and never part of `value`:
config.dr12q_mix and config.k40 (one kernel-timed launch each), config.pcie_c (host arrays in, host
arrays out through the one-shot C entry gpdla_process_batch: the PCIe-inclusive rate a C / MEX caller
gets) and, with --pcie, the same through the Python surface.

Environment: HSA_ENABLE_IPC_MODE_LEGACY=0 is set (if unset) before torch is imported, in both launch
forms -- this pool's host driver only supports dmabuf IPC, and without it RCCL's intra-node
transport fails with `hipIpcGetMemHandle: invalid argument`.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 2048 flop / 64 cycles x 2.4 GHz
FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32, MI355X_MICROARCH.md (study variant only)


def algorithmic_flops(n, k: int):
    """n k (k+3) + k^3/3 flops per log-likelihood evaluation."""
    return n * k * (k + 3) + k ** 3 / 3.0


def lib_sha256() -> str:
    from gp_dla_detection_amd import _lib
    with open(_lib.lib_path(), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def pmc_traffic(args):
    """HBM bytes per k_sweep launch from the committed rocprofv3 --pmc summary (separate counter
    passes, FETCH_SIZE doubled per the gfx950 rule; profiles/pmc_latest.json).  PMC collection
    cannot run inside this process, so the figure is reported only when it was measured on the
    same workload AND on the very libgpdla.so that is loaded now (sha256 recorded by
    tools/pmc_to_json.py); otherwise null."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            p = json.load(f)
        c = p["config"]
        same = (c.get("workload", "configs1"), c["spectra"], c["k"], c["dla_samples"], c.get("num_lines", 3)) == (
            args.workload, args.spectra, args.k, args.samples, args.num_lines)
        if same and p.get("lib_sha256") == lib_sha256():
            return float(p["hbm_bytes_per_launch"]), p.get("tag")
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def host_cpu_info() -> dict:
    """Logical CPUs this process may run on, physical cores among them, and the cgroup CPU quota
    (a GPU box hands one GPU's share of a large host to the job: the quota, not nproc, bounds what
    OpenMP can use)."""
    affinity = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count()))
    info = {"logical_cpus": len(affinity), "cpu_model": None, "cores_physical": None, "cpu_quota": None}
    try:
        cores, model, cur = set(), None, {}
        with open("/proc/cpuinfo") as f:
            for line in f.read().splitlines() + [""]:
                if not line.strip():
                    if "processor" in cur and int(cur["processor"]) in set(affinity):
                        cores.add((cur.get("physical id", "0"), cur.get("core id", cur["processor"])))
                    cur = {}
                    continue
                key, _, val = line.partition(":")
                cur[key.strip()] = val.strip()
                if key.strip() == "model name":
                    model = val.strip()
        info["cores_physical"], info["cpu_model"] = len(cores) or None, model
    except OSError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    info["cpu_quota"] = int(txt[0]) / int(txt[1])
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        info["cpu_quota"] = q / int(f.read())
            break
        except (OSError, ValueError, IndexError):
            continue
    return info


def cpu_baseline(model, samples, spectra, seconds_target=6.0, repeats=3):
    """The CPU oracle (literal as-written restatement of the reference path, OpenMP over samples
    like the reference's parfor, process_qsos.m:185) timed on this host: the timing build of the
    same source (-O3 -march=native, compiled here; same operations in the same order as the
    checker build, verified below), on a bounded sample of the same workload -- whole quasars
    (all S samples each) -- `repeats` times; the median rate is reported.  Threads = physical
    cores available to this job (capped by the cgroup CPU quota)."""
    from oracle import oracle
    info = host_cpu_info()
    threads = info["cores_physical"] or info["logical_cpus"]
    if info["cpu_quota"]:
        threads = max(1, min(threads, int(math.floor(info["cpu_quota"] + 1e-9))))
    S = samples["offset_samples"].size
    lib = oracle.load_timing()

    def run(sp, count, nthreads, use=lib):
        t0 = time.perf_counter()
        r = oracle.process_spectrum(model, samples["offset_samples"][:count], samples["nhi_samples"][:count],
                                    sp["wavelengths"], sp["flux"], sp["noise_variance"],
                                    sp["pixel_mask"], sp["z_qso"], num_threads=nthreads, lib=use)
        return time.perf_counter() - t0, r

    # the timing build computes what the checker build computes
    _, a = run(spectra[0], 32, 1)
    _, b = run(spectra[0], 32, 1, use=oracle.load())
    same = float(np.nanmax(np.abs(a["sample_log_likelihoods_dla"] - b["sample_log_likelihoods_dla"])))
    t_probe, _ = run(spectra[0], min(S, 8 * threads), threads)  # also warms the thread pool
    rate_guess = min(S, 8 * threads) / t_probe
    per_repeat = max(1, min(len(spectra), int(round(seconds_target * rate_guess / S))))
    rates, spent_all, checked = [], 0.0, {}
    for r in range(repeats):
        spent = 0.0
        for q in range(per_repeat):
            idx = (r * per_repeat + q) % len(spectra)
            dt, res = run(spectra[idx], S, threads)
            spent += dt
            checked[idx] = res  # the oracle's numbers for this quasar: compared with the GPU's by main()
        rates.append(per_repeat * S / spent)
        spent_all += spent
    one_count = min(S, 256)
    t1 = min(run(spectra[0], one_count, 1)[0] for _ in range(3))
    single = one_count / t1
    value = float(np.median(rates))
    n_desc = sorted({int(np.asarray(s["wavelengths"]).size) for s in spectra[:per_repeat * repeats]})
    return checked, dict(value=value, unit="evals/s", cores=int(threads), kind="port",
                cores_physical=info["cores_physical"], logical_cpus=info["logical_cpus"],
                cpu_quota=info["cpu_quota"], cpu_model=info["cpu_model"], threads=int(threads),
                repeats=repeats, rates=[float(x) for x in rates], single_core_value=single,
                parallel_efficiency=value / (threads * single),
                build="gcc -O3 -march=native -fopenmp -ffp-contract=off (oracle/Makefile `timing`); "
                      f"max |delta| vs the checker build {same:.1e}",
                host=(f"{threads} threads: the CPU quota of this job -- a {info['cpu_quota'] or threads:g}-CPU slice of a "
                      f"{info['logical_cpus']}-logical-CPU host" if info["cpu_quota"] else f"{threads} threads"),
                sample=f"{repeats} x {per_repeat} quasar(s) ({n_desc[0]}..{n_desc[-1]} stored pixels) x all {S} "
                       f"samples, {spent_all:.1f} s in total, OpenMP over samples on {threads} threads; "
                       f"single_core_value from {one_count} samples on 1 thread (best of 3)")


def pcie_c_leg(model, samples, spectra, lp, params, device, S, resident_rate, n_pc=2048, repeats=3):
    """gpdla_process_batch -- host CSR arrays in, host result arrays out, everything between inside the
    library (its own upload / sweep / download threads) -- called through ctypes the way a C or MEX
    consumer calls it: ONE call for n_pc quasars, default batching.  The arrays are built before the
    clock starts (a caller has them: preloaded_qsos.mat); the result arrays are fresh (untouched pages)
    for every repeat, as a caller's would be.  Median of `repeats`."""
    import ctypes as C

    import gp_dla_detection_amd as gp
    from gp_dla_detection_amd import _lib, api
    lib = _lib.load()
    many = [spectra[i % len(spectra)] for i in range(n_pc)]
    csr = api.spectra_to_csr(many)
    lp_no, lp_dla = np.resize(lp[0], n_pc), np.resize(lp[1], n_pc)
    cfg = api._config(params)
    keep = []
    m, s_ = api._model_struct(model, keep), api._samples_struct(samples, keep)
    sp = api._spectra_struct(csr, lp_no, lp_dla, None, keep)
    small = api.process_qsos(model, samples, many[:64], log_priors=(lp_no[:64], lp_dla[:64]), device=device,
                             params=params)  # warm-up of the host paths (and a reference for the check below)
    times = []
    for _ in range(repeats):
        out = gp.Batch.empty_results(n_pc, S, True)
        r = api._result_struct(_lib.Results, out)
        t0 = time.perf_counter()
        _lib.check(lib.gpdla_process_batch(C.byref(m), C.byref(s_), C.byref(sp), C.byref(cfg), C.byref(r), int(device)))
        times.append(time.perf_counter() - t0)
    np.testing.assert_array_equal(out["sample_log_likelihoods_dla"][:64], small["sample_log_likelihoods_dla"])
    assert not np.isnan(out["log_likelihoods_dla"]).any()
    dt = float(np.median(times))
    rate = n_pc * S / dt
    return {"what": f"gpdla_process_batch through ctypes: {n_pc} quasars as CSR host arrays in, "
                    f"{out['sample_log_likelihoods_dla'].nbytes / 1e6:.0f} MB of results out, one call (upload / sweep / "
                    "download pipelined inside the library); never `value`",
            "evals_per_s": rate, "seconds": times, "over_resident": rate / resident_rate,
            "batch_quasars": api.default_batch_size(n_pc, int(np.diff(csr["offsets"]).max()), int(m.k), S, 3)}


def self_launch(n: int, argv) -> int:
    """`python bench.py --gpus N` typed directly (no WORLD_SIZE in the environment): run the N
    ranks as child processes of this one, which has made no GPU call and makes none -- a process
    that has initialised the GPU must neither fork nor exec on this pool.  Each child gets the
    environment torch.distributed.run would give it (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR =
    127.0.0.1, a free MASTER_PORT); rank 0's child prints the JSON line on the inherited stdout.
    If a rank dies the others are ended (by their exact PIDs) instead of waiting in a collective.
    Returns the worst exit status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), GROUP_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # as main() does under torch.distributed.run
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    worst = 0
    alive = list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0:
                worst = worst or rc
                for q in alive:  # the others would wait for this rank forever
                    q.terminate()
    return worst


def launch_check():
    """--launch-check: what a rank does to prove the launch worked, with no GPU involved: rendezvous
    over gloo, all-reduce the ranks, rank 0 prints one JSON line (tests/test_bench_launch.py)."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(t)
    local = torch.tensor([float(os.environ["LOCAL_RANK"])], dtype=torch.float64)
    dist.all_reduce(local)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "rank_sum": t.item(),
                          "local_rank_sum": local.item(),
                          "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def job_cpus(world: int) -> int:
    """CPUs one rank may use for host-side set-up (cgroup quota / affinity, split over the ranks)."""
    info = host_cpu_info()
    n = info["logical_cpus"]
    if info["cpu_quota"]:
        n = min(n, int(info["cpu_quota"]))
    return max(1, min(16, n // max(1, world)))


SHARD_BATCH = 32768  # quasars per HBM-resident batch of the dr12q-shard workload


def main():
    t_process = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["configs1", "dr12q-mix", "dr12q-shard"], default="configs1")
    ap.add_argument("--spectra", type=int, default=1000, help="quasars per GPU per step (configs1, dr12q-mix)")
    ap.add_argument("--total-spectra", type=int, default=162861,
                    help="dr12q-shard: quasars of the whole run, sharded over the GPUs (README.md:115)")
    ap.add_argument("--pixels", type=int, default=1500, help="configs1 only")
    ap.add_argument("--mask-runs", action="store_true",
                    help="dr12q-mix: the 5 %% of masked pixels in contiguous runs of 4..12 (sky lines, bad columns) "
                         "instead of independent pixels")
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--contraction", choices=["f64", "f32"], default="f64",
                    help="f32: BASELINE config 5's study variant (fp32 matrix-core contraction, fp64 "
                         "everything else); not parity-grade, reports its max-abs delta vs f64")
    ap.add_argument("--num-lines", type=int, default=3,
                    help="Lyman-series members in the Voigt profile (set_parameters.m:63: 3; voigt.c:16 allows 31); "
                         "a diagnostic: the headline is quoted at 3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mix-rider", action="store_true",
                    help="configs1 at N = 1 also times ONE launch of the dr12q-mix shape after the timed "
                         "region (config.dr12q_mix, never value), and one of 256 quasars at k = 40 (config.k40); "
                         "this skips both")
    ap.add_argument("--pcie", action="store_true",
                    help="also time the one-shot host-buffer entry point (H2D + sweep + D2H); "
                         "reported as config.pcie_inclusive_evals_per_s, never as value")
    ap.add_argument("--pcie-c", action="store_true", help=argparse.SUPPRESS)  # (kept for old command lines: now the default at N = 1)
    ap.add_argument("--no-pcie-c", action="store_true",
                    help="the N = 1 configs1 line also times gpdla_process_batch itself -- the one-shot C entry a MEX "
                         "gateway binds -- on 2048 quasars handed over as CSR host arrays through ctypes, after the "
                         "timed region (config.pcie_c, never value; about two seconds); this skips it")
    ap.add_argument("--timeout-min", type=float, default=10.0, help="process-group timeout (N > 1)")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))  # nothing above touched the GPU
    # this pool's host driver only supports dmabuf IPC (see the module docstring); set before torch /
    # RCCL load, identically for the self-launched ranks and for ranks torch.distributed.run started
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.launch_check:
        return launch_check()

    import datetime

    import torch
    import torch.distributed as dist

    import gp_dla_detection_amd as gp
    from gp_dla_detection_amd import synthetic
    from gp_dla_detection_amd.distributed import gather_summaries, shard_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world} in the environment")
    # GPDLA_BENCH_REHEARSAL=1: rehearse the multi-process path on a box with ONE GPU (every rank on
    # cuda:0, gloo instead of RCCL, the gather staged through host memory).  Never used by the driver.
    rehearsal = os.environ.get("GPDLA_BENCH_REHEARSAL") == "1"
    shard = args.workload == "dr12q-shard"

    model = synthetic.make_model(args.k)
    samples = synthetic.make_samples(args.samples)
    bounds = None
    if shard:
        # the whole run's blocks from the pixel counts alone (what a real run reads from dataset
        # headers); a rank then makes only its own block -- before it touches the GPU, because the
        # generator's worker processes must not descend from a process that holds a HIP context
        z_all = synthetic.sample_dr12q_redshifts(args.total_spectra)
        bounds = shard_bounds(synthetic.boss_pixel_counts(z_all), world)
        lo, hi = bounds[rank]
        spectra = synthetic.make_dr12q_mix_parallel(lo, hi - lo, args.k, job_cpus(world))
        n_kept = synthetic.kept_pixel_counts(spectra)
    elif args.workload == "configs1":
        # every spectrum distinct, on every rank (as in dr12q-mix: one method for both workloads)
        spectra = synthetic.make_spectra(args.spectra, args.pixels, model, first_index=args.spectra * rank)
        n_kept = np.full(args.spectra, args.pixels)
    else:
        spectra = synthetic.make_dr12q_mix(args.spectra, model, first_index=args.spectra * rank,
                                           mask_runs=args.mask_runs)
        n_kept = synthetic.kept_pixel_counts(spectra)
    nloc = len(spectra)
    cat = synthetic.make_prior_catalog()
    z = np.array([s["z_qso"] for s in spectra])
    lp = gp.dla_existence_prior(cat["z_qsos"], cat["dla_ind"], z) if nloc else (np.zeros(0), np.zeros(0))

    if rehearsal:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:  # device_count() does not initialise the GPU
        raise SystemExit(f"rank {rank}: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    if world > 1:
        timeout = datetime.timedelta(minutes=args.timeout_min)
        if rehearsal:
            dist.init_process_group("gloo", timeout=timeout)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=timeout)

    stream = torch.cuda.Stream()
    if args.num_lines != 3:  # a diagnostic line: the oracle leg and the riders are quoted at three lines
        args.no_cpu_baseline = args.no_mix_rider = args.no_pcie_c = True
    params = gp.Parameters(contraction_precision=1 if args.contraction == "f32" else 0, num_lines=args.num_lines)
    ctx = gp.Context(local_rank, params=params, stream=stream)
    ctx.set_model(model)
    ctx.set_samples(samples)
    # inputs resident in HBM before the timed region: one batch, or (dr12q-shard) batches of SHARD_BATCH
    cuts = [(b0, min(b0 + SHARD_BATCH, nloc)) for b0 in range(0, nloc, SHARD_BATCH)] if shard else [(0, nloc)]
    batches = [ctx.upload(spectra[b0:b1], lp[0][b0:b1], lp[1][b0:b1]) for b0, b1 in cuts if b1 > b0]
    cuts = [c for c in cuts if c[1] > c[0]]
    batch = batches[0] if batches else None
    ctx.set_timing(True)
    counts = [b[1] - b[0] for b in bounds] if shard else [args.spectra] * world
    gather_dev = "cpu" if rehearsal else f"cuda:{local_rank}"
    table = torch.empty((nloc, 15), dtype=torch.float64, device=f"cuda:{local_rank}") if shard else None
    if world > 1:
        # set-up, like the upload above: the first collective creates the RCCL communicator (seconds);
        # it must not fall into the timed region when the driver asks for --warmup 0
        with torch.cuda.stream(stream):
            probe = torch.zeros(8, dtype=torch.float64, device=gather_dev)
            dist.all_reduce(probe)
        torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_process

    gather_events = []

    def step():
        """One pass of the hot path over this rank's resident quasars + the gather of the table.
        Returns the sweep kernels' milliseconds (hipEvent pairs inside the library; reading them
        waits for the sweep)."""
        ms = 0.0
        with torch.cuda.stream(stream):
            for (b0, b1), bt in zip(cuts, batches):
                bt.process()
                ms += ctx.last_sweep_ms()
                if shard:
                    table[b0:b1].copy_(bt.summary_tensor())
            if world > 1:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                gather_summaries(table if shard else batch.summary_tensor(), counts)
                e1.record(stream)
                gather_events.append((e0, e1))
        return ms

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    gather_events.clear()
    # timed region: exactly K steps; the sweep kernel of every step is bracketed by a hipEvent pair
    # recorded on the launch stream inside the library (read back per step: an event wait only)
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kernel_ms.append(step())
    fence()
    elapsed_local = elapsed = time.perf_counter() - t0
    sweep_ms = float(np.mean(kernel_ms))
    gather_ms = float(np.mean([a.elapsed_time(b) for a, b in gather_events])) if gather_events else 0.0
    per_rank = None
    if world > 1:
        mine = torch.tensor([elapsed_local / args.steps * 1e3, sweep_ms, gather_ms, setup_s, float(nloc),
                             float(np.sum(n_kept))], dtype=torch.float64, device=gather_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        every = torch.stack(every).cpu().numpy()
        elapsed = float(every[:, 0].max()) * args.steps / 1e3  # the slowest rank's clock
        per_rank = {"step_ms": every[:, 0].tolist(), "kernel_ms": every[:, 1].tolist(),
                    "gather_ms": every[:, 2].tolist(), "setup_s": every[:, 3].tolist(),
                    "quasars": [int(v) for v in every[:, 4]], "kept_pixels": [int(v) for v in every[:, 5]]}

    quasars_per_step = (args.total_spectra if shard else args.spectra * world)
    total_evals = quasars_per_step * args.samples * args.steps
    value = total_evals / elapsed
    flops = float(np.sum(algorithmic_flops(n_kept.astype(np.float64), args.k))) * args.samples
    achieved = flops / (sweep_ms * 1e-3) / 1e12 if sweep_ms > 0 else 0.0
    peak = FP64_MFMA_PEAK_TFLOPS if args.contraction == "f64" else FP32_MFMA_PEAK_TFLOPS

    out = None
    if rank == 0:
        traffic, traffic_tag = pmc_traffic(args)
        workload = {"configs1": "BASELINE configs[1]: synthetic spectra, fused Voigt + low-rank log-evidence sweep, "
                                "HBM-resident",
                    "dr12q-mix": "dr12q-mix: distinct synthetic spectra on the BOSS pixel grid (DR12Q length mix, "
                                 "5 % masked), fused Voigt + low-rank log-evidence sweep, HBM-resident",
                    "dr12q-shard": f"BASELINE configs[2]: {args.total_spectra} distinct synthetic spectra of the DR12Q "
                                   "length mix (BOSS pixel grid, 5 % masked) sharded over the GPUs in blocks balanced "
                                   "by pixel count, HBM-resident; all-gather of the whole run's posterior table"
                    }[args.workload]
        out = {
            "metric": "sample log-likelihoods/sec (n~1500,k=20)",
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if shard else "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.contraction == "f64" else "f32 contraction, f64 elsewhere",
            "data": "synthetic",
            "config": {"workload": workload,
                       **({"total_spectra": args.total_spectra, "spectra_rank0": nloc} if shard
                          else {"spectra_per_gpu": args.spectra}),
                       "pixels": args.pixels if args.workload == "configs1" else
                       {"kept_min": int(n_kept.min()), "kept_mean": float(n_kept.mean()),
                        "kept_max": int(n_kept.max())},
                       "k": args.k, "dla_samples": args.samples, "num_lines": args.num_lines,
                       "parallelism": f"spectra sharded over {world} GPU(s), "
                                      f"{'gloo (REHEARSAL on one GPU)' if rehearsal else 'RCCL'} all-gather of "
                                      "the 15-column posterior table",
                       "setup_s": setup_s},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_unit": "bytes/launch (rocprofv3 PMC, "
                         f"profiles/pmc_latest.json{', tag ' + traffic_tag if traffic_tag else ''}; null unless "
                         "measured on this libgpdla.so and workload)",
                         # (GPDLA_EXPANDED_RECORDS only means something to libgpdla_legacy.so, loaded through GPDLA_LIB_PATH by the A/B tools)
                         "kernel": (("k_sweep_slim<3>" if args.num_lines == 3 else "k_sweep_slim<0>") if args.k <= 20 and args.contraction == "f64"
                                    and not os.environ.get("GPDLA_EXPANDED_RECORDS") else
                                    ("k_sweep_split" if os.environ.get("GPDLA_EXPANDED_RECORDS") else "k_sweep_split_slim")
                                    if args.k > 20 and args.contraction == "f64" else "k_sweep"),
                         "kernel_ms": sweep_ms,
                         "flops_per_launch": flops,
                         "flops_per_eval": algorithmic_flops(float(n_kept.mean()), args.k)},
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
            out["roofline"]["rank"] = 0  # the roofline object is rank 0's kernel on rank 0's quasars
        if rehearsal:
            out["rehearsal"] = True  # ranks time-slice ONE GPU over gloo: not a scaling point
            out["backend"] = "gloo"
        if args.contraction == "f32":
            nchk = min(args.spectra, 4)
            lpc = (lp[0][:nchk], lp[1][:nchk])
            ref = gp.process_qsos(model, samples, spectra[:nchk], log_priors=lpc, device=local_rank)
            got = gp.process_qsos(model, samples, spectra[:nchk], log_priors=lpc, device=local_rank,
                                  params=params)
            out["config"]["max_abs_delta_vs_f64"] = {
                "sample_log_likelihoods_dla": float(np.nanmax(np.abs(
                    got["sample_log_likelihoods_dla"] - ref["sample_log_likelihoods_dla"]))),
                "log_likelihoods_dla": float(np.nanmax(np.abs(
                    got["log_likelihoods_dla"] - ref["log_likelihoods_dla"]))),
                "p_dlas": float(np.nanmax(np.abs(got["p_dlas"] - ref["p_dlas"]))),
                "quasars_checked": nchk}
        if args.pcie and world == 1:
            # host buffers in, host buffers out, through the script surface (api.process_qsos: bounded
            # batches, uploads and downloads overlapped with the sweeps); never `value`
            n_pc = 2048
            many = [spectra[i % len(spectra)] for i in range(n_pc)]
            lp_pc = (np.resize(lp[0], n_pc), np.resize(lp[1], n_pc))
            gp.process_qsos(model, samples, many[:64], log_priors=(lp_pc[0][:64], lp_pc[1][:64]),
                            device=local_rank, params=params)  # warm-up (first-touch of the host paths)
            t0 = time.perf_counter()
            res_pc = gp.process_qsos(model, samples, many, log_priors=lp_pc, device=local_rank, params=params)
            dt = time.perf_counter() - t0
            assert res_pc["sample_log_likelihoods_dla"].shape == (n_pc, args.samples)
            out["config"]["pcie_inclusive_evals_per_s"] = n_pc * args.samples / dt
            out["config"]["pcie_inclusive_sample"] = (f"{n_pc} quasars through process_qsos (host arrays in, "
                                                      f"{res_pc['sample_log_likelihoods_dla'].nbytes / 1e6:.0f} MB of "
                                                      "results out; upload / sweep / download pipelined)")
            out["config"]["pcie_inclusive_over_resident"] = out["config"]["pcie_inclusive_evals_per_s"] / value
        if world == 1 and args.workload == "configs1" and (args.pcie_c or (not args.no_pcie_c and args.contraction == "f64"
                                                                          and args.k <= 20 and not args.no_mix_rider)):
            try:  # a rider: its failure must not cost the headline line
                out["config"]["pcie_c"] = pcie_c_leg(model, samples, spectra, lp, params, local_rank, args.samples, value,
                                                     repeats=3 if args.pcie_c else 2)
            except Exception as e:  # noqa: BLE001
                out["config"]["pcie_c"] = {"error": f"{type(e).__name__}: {e}"}
        if (world == 1 and args.workload == "configs1" and args.contraction == "f64" and args.k <= 20
                and not args.no_mix_rider):
            # the production shape beside the headline: ONE launch of 1000 quasars of the DR12Q length
            # mix (after a warm-up launch), timed by the same hipEvent pair; never `value`
            mix = synthetic.make_dr12q_mix(1000, model)
            zm = np.array([s_["z_qso"] for s_ in mix])
            lpm = gp.dla_existence_prior(cat["z_qsos"], cat["dla_ind"], zm)
            bm = ctx.upload(mix, lpm[0], lpm[1])
            with torch.cuda.stream(stream):
                bm.process()
                bm.process()
                ms_mix = ctx.last_sweep_ms()
            kept = synthetic.kept_pixel_counts(mix)
            fl = float(np.sum(algorithmic_flops(kept.astype(np.float64), args.k))) * args.samples
            out["config"]["dr12q_mix"] = {
                "what": "one k_sweep_slim launch over 1000 distinct quasars of the DR12Q length mix (BOSS grid, 5 % "
                        "masked), after the timed region; kernel-timed, never `value`",
                "kernel_ms": ms_mix, "evals_per_s": 1000 * args.samples / (ms_mix * 1e-3),
                "achieved_tflops": fl / (ms_mix * 1e-3) / 1e12, "frac": fl / (ms_mix * 1e-3) / 1e12 / peak,
                "kept_pixels_mean": float(kept.mean())}
            bm.close()
            # ... and the 20 < k <= 40 class (BASELINE configs[3]/[4] territory): ONE launch of 256 quasars
            # x 1500 px at k = 40 through a second context, timed the same way; never `value`
            model40 = synthetic.make_model(40)
            sp40 = synthetic.make_spectra(256, args.pixels, model40, first_index=7000)
            z40 = np.array([s_["z_qso"] for s_ in sp40])
            lp40 = gp.dla_existence_prior(cat["z_qsos"], cat["dla_ind"], z40)
            ctx40 = gp.Context(local_rank, params=params, stream=stream)
            ctx40.set_model(model40)
            ctx40.set_samples(samples)
            ctx40.set_timing(True)
            b40 = ctx40.upload(sp40, lp40[0], lp40[1])
            with torch.cuda.stream(stream):
                b40.process()
                b40.process()
                ms40 = ctx40.last_sweep_ms()
            fl40 = 256 * algorithmic_flops(float(args.pixels), 40) * args.samples
            out["config"]["k40"] = {
                "what": f"one k_sweep_split_slim launch over 256 distinct quasars x {args.pixels} px at k = 40, after the "
                        "timed region; kernel-timed, never `value`",
                "kernel_ms": ms40, "evals_per_s": 256 * args.samples / (ms40 * 1e-3),
                "achieved_tflops": fl40 / (ms40 * 1e-3) / 1e12, "frac": fl40 / (ms40 * 1e-3) / 1e12 / peak}
            b40.close()
            ctx40.close()
        if world == 1 and not args.no_cpu_baseline and not shard:
            checked, out["cpu_baseline"] = cpu_baseline(model, samples, spectra)
            # the quasars the oracle has just swept (all S samples each), against the timed GPU batch:
            # the "max-abs delta" half of BASELINE.json's metric, measured on the bench workload itself
            table_s, summ = batch.samples_tensor(), batch.summary_tensor()
            worst = 0.0
            for idx, ref in checked.items():
                got = table_s[idx].cpu().numpy()
                row = summ[idx].cpu().numpy()
                worst = max(worst, float(np.nanmax(np.abs(got - ref["sample_log_likelihoods_dla"]))),
                            abs(float(row[4]) - ref["log_likelihood_no_dla"]),
                            abs(float(row[5]) - ref["log_likelihood_dla"]))
            out["parity"] = {"max_abs_delta_vs_oracle": worst, "tolerance": 1e-8, "quasars": len(checked),
                             "entries": len(checked) * (args.samples + 2),
                             "what": "sample_log_likelihoods_dla, log_likelihoods_no_dla and log_likelihoods_dla of "
                                     "the timed batch against the CPU oracle (the restatement of "
                                     "process_qsos.m / log_mvnpdf_low_rank.m / voigt.c; no MATLAB exists here)"}
            if args.contraction == "f64" and not (worst < 1e-8):
                raise SystemExit(f"parity check failed: max |delta| vs the oracle = {worst}")
        print(json.dumps(out), flush=True)
    for bt in batches:
        bt.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
