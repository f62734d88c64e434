#!/usr/bin/env python3
"""bench.py -- throughput of the GP/DLA inference sweep on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (gpdla_batch_process: selection + interpolation, null
evidence, S-sample Voigt/low-rank sweep, evidence + posteriors) over one HBM-resident batch of
synthetic quasars, followed -- when N > 1 -- by the RCCL all-gather of the posterior table.
Workload = BASELINE.json configs[1]: 1000 synthetic spectra x n = 1500 pixels x k = 20 x
S = 10 000 DLA samples, fp64, per GPU (weak scaling: every rank sweeps its own 1000 quasars, as a
DR12Q run would shard its 162 861).  value = sample log-likelihood evaluations per second over
all ranks.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 2048 flop / 64 cycles x 2.4 GHz
FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32, MI355X_MICROARCH.md (study variant only)


def algorithmic_flops(n: int, k: int) -> float:
    """n k (k+3) + k^3/3 flops per log-likelihood evaluation."""
    return n * k * (k + 3) + k ** 3 / 3.0


def pmc_traffic(args):
    """HBM bytes per k_sweep launch from the committed rocprofv3 --pmc summary (separate counter
    passes, FETCH_SIZE doubled per the gfx950 rule; profiles/pmc_latest.json).  PMC collection
    cannot run inside this process, so the figure is reported only when it was measured on the
    same workload shape; otherwise null."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            p = json.load(f)
        c = p["config"]
        if (c["spectra"], c["pixels"], c["k"], c["dla_samples"]) == (args.spectra, args.pixels, args.k,
                                                                     args.samples):
            return float(p["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(model, samples, spectra, seconds_target=12.0):
    """The CPU oracle (literal as-written restatement of the reference path, OpenMP over samples
    like the reference's parfor) timed on this host: a bounded sample of the same workload --
    whole quasars (all S samples each) until about `seconds_target` seconds have been spent."""
    from oracle import oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    S = samples["offset_samples"].size

    def run(sp, count, threads):
        t0 = time.perf_counter()
        oracle.process_spectrum(model, samples["offset_samples"][:count], samples["nhi_samples"][:count],
                                sp["wavelengths"], sp["flux"], sp["noise_variance"],
                                sp["pixel_mask"], sp["z_qso"], num_threads=threads)
        return time.perf_counter() - t0

    run(spectra[0], min(S, 4 * cores), cores)  # warm up the thread pool
    done, spent = 0, 0.0
    while spent < seconds_target and done < len(spectra):
        spent += run(spectra[done], S, cores)
        done += 1
    one_count = min(S, 64)
    t1 = run(spectra[0], one_count, 1)
    return dict(value=done * S / spent, unit="evals/s", cores=int(cores), kind="port",
                single_core_value=one_count / t1,
                sample=f"{done} quasar(s) n={spectra[0]['wavelengths'].size - 4} x all {S} samples, "
                       f"{spent:.1f} s, OpenMP over samples on {cores} threads; "
                       f"single_core_value from {one_count} samples on 1 thread")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spectra", type=int, default=1000, help="quasars per GPU per step")
    ap.add_argument("--pixels", type=int, default=1500)
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--contraction", choices=["f64", "f32"], default="f64",
                    help="f32: BASELINE config 5's study variant (fp32 matrix-core contraction, fp64 "
                         "everything else); not parity-grade, reports its max-abs delta vs f64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pcie", action="store_true",
                    help="also time the one-shot host-buffer entry point (H2D + sweep + D2H); "
                         "reported as config.pcie_inclusive_evals_per_s, never as value")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import gp_dla_detection_amd as gp
    from gp_dla_detection_amd import synthetic
    from gp_dla_detection_amd.distributed import gather_summaries

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # GPDLA_BENCH_REHEARSAL=1: rehearse the multi-process path on a box with ONE GPU (every rank on
    # cuda:0, gloo instead of RCCL, the gather staged through host memory).  Never used by the driver.
    rehearsal = os.environ.get("GPDLA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    model = synthetic.make_model(args.k)
    samples = synthetic.make_samples(args.samples)
    # distinct spectra per rank; only a few distinct realisations are generated and tiled (the
    # sweep's cost does not depend on the flux values)
    distinct = min(args.spectra, 16)
    base = synthetic.make_spectra(distinct, args.pixels, model, first_index=1000 * rank)
    spectra = [base[i % distinct] for i in range(args.spectra)]
    cat = synthetic.make_prior_catalog()
    z = np.array([s["z_qso"] for s in spectra])
    lp = gp.dla_existence_prior(cat["z_qsos"], cat["dla_ind"], z)

    stream = torch.cuda.Stream()
    params = gp.Parameters(contraction_precision=1 if args.contraction == "f32" else 0)
    ctx = gp.Context(local_rank, params=params, stream=stream)
    ctx.set_model(model)
    ctx.set_samples(samples)
    batch = ctx.upload(spectra, lp[0], lp[1])  # inputs resident in HBM before the timed region
    ctx.set_timing(True)
    counts = [args.spectra] * world

    def step():
        with torch.cuda.stream(stream):
            batch.process()
            if world > 1:
                table = batch.summary_tensor()
                gather_summaries(table.cpu() if rehearsal else table, counts)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # timed region: exactly K steps; the sweep kernel of every step is bracketed by a hipEvent pair
    # recorded on the launch stream inside the library (read back per step: an event wait only)
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(ctx.last_sweep_ms())
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    sweep_ms = float(np.mean(kernel_ms))

    evals_per_step = args.spectra * args.samples
    total_evals = evals_per_step * world * args.steps
    value = total_evals / elapsed
    flops = algorithmic_flops(args.pixels, args.k) * evals_per_step
    achieved = flops / (sweep_ms * 1e-3) / 1e12
    peak = FP64_MFMA_PEAK_TFLOPS if args.contraction == "f64" else FP32_MFMA_PEAK_TFLOPS

    out = None
    if rank == 0:
        out = {
            "metric": "sample log-likelihoods/sec (n~1500,k=20)",
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.contraction == "f64" else "f32 contraction, f64 elsewhere",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic spectra, fused Voigt + low-rank "
                                   "log-evidence sweep, HBM-resident",
                       "spectra_per_gpu": args.spectra, "pixels": args.pixels, "k": args.k,
                       "dla_samples": args.samples, "num_lines": 3,
                       "parallelism": f"spectra sharded over {world} GPU(s), RCCL all-gather of "
                                      "the 12-column posterior table"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": pmc_traffic(args), "traffic_unit": "bytes/launch (rocprofv3 PMC, "
                         "profiles/pmc_latest.json)", "kernel": "k_sweep", "kernel_ms": sweep_ms,
                         "flops_per_eval": algorithmic_flops(args.pixels, args.k)},
        }
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
            n_pc = min(args.spectra, 128)
            t0 = time.perf_counter()
            gp.process_qsos(model, samples, spectra[:n_pc], log_priors=(lp[0][:n_pc], lp[1][:n_pc]),
                            device=local_rank)
            out["config"]["pcie_inclusive_evals_per_s"] = n_pc * args.samples / (time.perf_counter() - t0)
            out["config"]["pcie_inclusive_sample"] = f"{n_pc} quasars through gpdla_process_batch (host buffers)"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, samples, spectra)
        print(json.dumps(out), flush=True)
    batch.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
