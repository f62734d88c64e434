"""File-to-file sharded run: BASELINE config 3 (DR12Q on the 8 GPUs of a node) as a pipeline.

The reference processes a catalogue by running ``process_qsos`` on disjoint ``test_ind`` slices as
separate batch jobs, each saving its own ``processed_qsos_*.mat``, and recombining the chunks
afterwards with ``mat_combine`` (CDDF_analysis/sbatch_reunion.py:13-63).  Here the jobs are the
ranks of one ``torch.distributed`` process group, one per GPU:

1. every rank opens the ``-v7.3`` inputs of process_qsos.m:30-61 (catalogue, learned model, DLA
   samples, preloaded spectra) and takes the run's ``test_ind`` selection (:52-61);
2. the selected quasars are split into contiguous blocks balanced by pixel count
   (:func:`distributed.shard_bounds`; the counts come from dataset headers, no spectrum is read);
3. a rank sweeps its block in bounded HBM-resident batches through :func:`api.run_pipeline` -- the
   spectra of batch i+1 are read from the file and uploaded, and the results of batch i-1
   downloaded, while batch i is swept;
4. each rank writes its own chunk ``processed_qsos_<name>_<lo>-<hi>.mat`` (single-DLA variables of
   process_qsos.m:236-250, or the multi-DLA list of multi :498-523) with the chunk's own
   ``test_ind`` -- a file the reference's ``mat_combine`` recombines unchanged
   (tests/golden/make_consumer_fixtures.py does exactly that); the per-sample tables, the only
   large variables, are written batch by batch while the next batch is swept
   (:class:`io.ProcessedStreamWriter`: one column of HDF5 chunks per batch);
5. the per-quasar posterior rows (15 or 78 fp64) are all-gathered over RCCL, so every rank -- and
   rank 0's ``*_summary.mat`` -- holds the whole run's posterior table; the per-sample tables stay
   in the chunks, as the reference keeps them per job.

Launch: ``python -m torch.distributed.run --nproc-per-node 8 -m gp_dla_detection_amd.run_dr12q ...``
or call :func:`run` from a process whose default group is initialised (world 1 needs no group).
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from . import _lib, io
from .api import (Batch, Context, batch_blocks, default_batch_size, dla_existence_prior,
                  dla_existence_prior_multi, run_pipeline)
from .distributed import (_world, agree_on_failure, gather_summaries, shard_bounds, summary_to_fields,
                          summary_to_fields_multi)
from .parameters import MultiParameters, Parameters


def select_test_ind(catalog: dict, test_ind=None) -> np.ndarray:
    """0-based catalogue indices of the run: ``test_ind`` as a boolean mask / index array, or the
    reference's default selection ``catalog.filter_flags == 0`` (README.md:296)."""
    n = np.asarray(catalog["z_qsos"]).size
    if test_ind is None:
        flags = catalog.get("filter_flags")
        return np.arange(n) if flags is None else np.flatnonzero(np.asarray(flags).reshape(-1) == 0)
    t = np.asarray(test_ind).reshape(-1)
    if t.dtype == bool:
        if t.size != n:
            raise ValueError(f"test_ind has {t.size} entries for {n} catalogue quasars")
        return np.flatnonzero(t)
    return t.astype(np.int64)


def ramped_blocks(n: int, batch: int, ramp: int = 8):
    """Batches of a pipelined run: ``batch`` quasars each, but a small first one and a small last one
    (``batch // ramp``) -- the first batch's read + upload and the last batch's download + write are
    the two stretches of a run that nothing overlaps.  Every batch but the last starts and ends on
    a multiple of the returned grid (the chunk width of :class:`io.ProcessedStreamWriter`).
    Returns ``([(lo, hi), ...], grid)``."""
    batch = max(1, int(batch))
    small = batch // ramp
    if small < 16 or n <= batch:  # nothing to ramp: plain fixed-size batches
        return batch_blocks(n, batch), batch
    batch = (batch // small) * small
    blocks, lo = [(0, small)], small
    while n - lo > batch + small:
        blocks.append((lo, lo + batch))
        lo += batch
    mid = ((n - lo - small) // small) * small  # what is left: a batch on the grid, then the small last one
    if mid > 0:
        blocks.append((lo, lo + mid))
        lo += mid
    blocks.append((lo, n))
    return blocks, small


#: host bytes one batch's per-sample tables may take in a multi-DLA run (they are 6.5 times the
#: single-DLA table per quasar): the batch size is capped so that the staging buffer stays this small
MULTI_STAGING_BYTES = 512 << 20


def _staging(rows: int, S: int, md: int) -> dict:
    """ONE batch's worth of host arrays for the downloads of a run (per-sample tables included),
    page-locked where torch can provide that (a device-to-host copy into pageable memory is staged
    through a bounce buffer by the runtime and runs at less than half the rate)."""
    out = Batch.empty_results_multi(rows, md, S) if md else Batch.empty_results(rows, S)
    try:
        import torch
        for name in ("sample_log_likelihoods_dla", "sample_log_likelihoods_lls", "base_sample_inds"):
            if name in out and out[name].size:
                t = torch.empty(out[name].nbytes, dtype=torch.uint8, pin_memory=True)
                out[name] = t.numpy().view(out[name].dtype).reshape(out[name].shape)
                out.setdefault("_pinned", []).append(t)  # keeps the allocation alive
    except Exception:  # no pinned memory (no GPU runtime): pageable arrays work, slower
        pass
    return out


def run(preloaded_file: str, catalog_file: str, learned_file: str, samples_file: str, out_dir: str,
        test_set_name: str = "dr12q", test_ind=None, prior_catalog: dict | None = None,
        multi: bool = False, params: Parameters | None = None, Z_lls: float | None = None,
        Z_dla: float | None = None, device: int | None = None, max_quasars_per_batch: int | None = None,
        pipeline_slots: int = 3, run_metadata: dict | None = None, write_summary: bool = True) -> dict:
    """One rank of the sharded file-to-file run (see the module docstring).

    ``prior_catalog``: ``{"z_qsos", "dla_ind"}`` of the training release (``api.prepare_prior``);
    multi-DLA runs also need ``Z_lls`` / ``Z_dla`` (set_lls_parameters.m:59-71).
    Host memory is bounded by ONE batch: each batch's results are downloaded into a page-locked
    staging buffer, its per-sample tables are handed to the chunk writer from there, and only the
    per-quasar variables (a few hundred bytes per quasar) are kept for the end of the file -- the
    reference pre-fills the whole run's tables (process_qsos.m:74-82) and saves them at the end
    (:250), and its recombination script is known for the 150 GB that costs
    (CDDF_analysis/sbatch_reunion.py:6-7).
    A rank that fails does not leave the others waiting: every rank reports before the gather
    (:func:`distributed.agree_on_failure`) and every rank raises; the failing rank removes its partial
    chunk file, the finished chunks of the others stay (each is a complete file for its block, as
    the chunks of the reference's independent jobs are, CDDF_analysis/sbatch_reunion.py:13-27).
    Returns ``dict(fields=<posterior variables of ALL quasars of the run>, block=(lo, hi),
    chunk=<path of this rank's chunk file or None>, selected=<catalogue indices of the run>,
    timings=<seconds: setup_s, pipeline_s, save_s, total_s>)``."""
    import time

    import torch

    t_start = time.perf_counter()
    timings = {}
    world, rank = _world()
    p = params or (MultiParameters() if multi else Parameters())
    md = p.max_dlas if multi else 0
    if device is None:
        device = torch.cuda.current_device()
    state = dict(chunk=None, writer=None)

    def local_work():
        """Everything this rank does on its own: returns (sel, z_all, bounds, table)."""
        if prior_catalog is None:
            raise ValueError("need prior_catalog (z_qsos, dla_ind of the training release)")
        catalog = io.load_catalog(catalog_file)
        z_all = np.asarray(catalog["z_qsos"], dtype=np.float64).reshape(-1)
        sel = select_test_ind(catalog, test_ind)
        model = io.load_learned_model(learned_file)
        samples = io.load_dla_samples(samples_file)
        S, k = samples["offset_samples"].size, model["M"].shape[1]
        os.makedirs(out_dir, exist_ok=True)
        reader = io.PreloadedReader(preloaded_file)
        try:
            if reader.num_quasars != z_all.size:
                raise ValueError(f"{preloaded_file} holds {reader.num_quasars} spectra, the catalogue {z_all.size}")
            counts = reader.pixel_counts(sel)
            bounds = shard_bounds(counts, world)
            lo, hi = bounds[rank]
            nloc = hi - lo
            z_sel = z_all[sel]
            if multi:
                if Z_lls is None or Z_dla is None:
                    raise ValueError("a multi-DLA run needs Z_lls and Z_dla (set_lls_parameters.m:59-71)")
                lp_no, lp_lls, lp_dla = dla_existence_prior_multi(prior_catalog["z_qsos"], prior_catalog["dla_ind"],
                                                                  z_sel[lo:hi], Z_lls, Z_dla, p)
                ncol = _lib.summary_cols_multi(md)
            else:
                lp_no, lp_dla = dla_existence_prior(prior_catalog["z_qsos"], prior_catalog["dla_ind"], z_sel[lo:hi], p)
                lp_lls = None
                ncol = _lib.SUMMARY_COLS
            # the rank's posterior table lives on the GPU: each batch's rows are copied into it behind
            # its sweep (same stream), and the whole table is what the RCCL all-gather reads
            stream = torch.cuda.Stream(device=device)
            table = torch.empty((nloc, ncol), dtype=torch.float64, device=f"cuda:{device}")
            if nloc:
                per_batch = max_quasars_per_batch
                if per_batch is None:
                    per_batch = default_batch_size(nloc, int(counts[lo:hi].max()), k, S, pipeline_slots,
                                                   multi_models=(md + 1) if multi else 0)
                    if multi:  # keep the staging buffer of a batch small (see MULTI_STAGING_BYTES)
                        per_q = S * (8 * md + 8 + 4 * max(md - 1, 0))
                        per_batch = max(64, min(per_batch, MULTI_STAGING_BYTES // per_q))
                blocks, grid = ramped_blocks(nloc, per_batch)
                # per-quasar variables of the whole block (small); per-sample tables of ONE batch
                small = (Batch.empty_results_multi(nloc, md, S, with_samples=False) if multi
                         else Batch.empty_results(nloc, S, with_samples=False))
                stage = _staging(max(b1 - b0 for b0, b1 in blocks), S, md)
                ctx = Context(device, p, stream=stream)
                # the chunk file is open from the start: the download thread transposes each batch's
                # per-sample tables into MATLAB's order and writes them while the next batch is swept
                state["chunk"] = io.chunk_filename(out_dir, test_set_name, lo, hi, multi)
                writer = state["writer"] = io.ProcessedStreamWriter(state["chunk"], nloc, S, grid, md)
                copied = [None] * len(blocks)

                def inputs(i):  # runs on the upload thread: file reads overlap the sweep in flight
                    b0, b1 = blocks[i]
                    spectra = reader.read_csr(sel[lo + b0:lo + b1], z_all)  # flat arrays, native reader (csrc/h5cells.c)
                    args = (spectra, lp_no[b0:b1], lp_dla[b0:b1])
                    return args + ((lp_lls[b0:b1],) if multi else ())

                def process(i, batch):
                    b0, b1 = blocks[i]
                    with torch.cuda.stream(stream):
                        if multi:
                            ctx.set_first_quasar_index(p.first_quasar_index + lo + b0)
                            batch.process_multi()
                        else:
                            batch.process()
                        table[b0:b1].copy_(batch.summary_tensor())
                        # (the slot must not be re-filled before this copy has run: see distributed.py)
                        copied[i] = torch.cuda.Event()
                        copied[i].record(stream)

                def download(i, batch):  # the one download thread: the staging buffer is its own
                    b0, b1 = blocks[i]
                    n = b1 - b0
                    (batch.download_multi(True, stage, 0) if multi else batch.download(True, stage, 0))
                    for name, dst in small.items():
                        dst[b0:b1] = stage[name][:n]
                    writer.append(b0, {k_: stage[k_][:n] for k_ in writer.streamed})
                    copied[i].synchronize()

                try:
                    ctx.set_model(model)
                    ctx.set_samples(samples)
                    timings["setup_s"] = time.perf_counter() - t_start
                    run_pipeline(ctx, len(blocks), inputs, process, download, pipeline_slots)
                    stream.synchronize()
                    timings["pipeline_s"] = time.perf_counter() - t_start - timings["setup_s"]
                finally:
                    ctx.close()
                t_save = time.perf_counter()
                mask = np.zeros(z_all.size, dtype=bool)
                mask[sel[lo:hi]] = True
                meta = dict(test_set_name=test_set_name, **(run_metadata or {}))
                small.update(num_lines=p.num_lines, prior_z_qso_increase=p.prior_z_qso_increase, max_z_cut=p.max_z_cut)
                if multi:
                    small.update(k=k, min_z_cut=p.min_z_cut, num_dla_samples=S)
                writer.finish(small, test_ind=mask, **meta)
                state["writer"] = None
                timings["save_s"] = time.perf_counter() - t_save
        finally:
            reader.close()
        return sel, z_all, bounds, table, stream

    def discard_chunk():
        if state["writer"] is not None:
            state["writer"].abort()
        if state["chunk"] and os.path.exists(state["chunk"]):
            os.remove(state["chunk"])  # a partial chunk file must not be mistaken for a finished one

    error, result = None, None
    try:
        result = local_work()
    except Exception as e:
        error = e
    except BaseException:
        # KeyboardInterrupt / SystemExit in the middle of the pipeline: the partial chunk file must not
        # survive looking like a finished one.  (The other ranks then leave their all-reduce by the
        # process-group timeout: an interrupt is not a failure to agree on.)
        discard_chunk()
        raise
    try:
        agree_on_failure(error, what="file-to-file run")  # raises on EVERY rank if any rank failed
    except BaseException:
        if error is not None:  # (a finished chunk of a rank that did not fail stays: it is complete)
            discard_chunk()
        raise
    sel, z_all, bounds, table, stream = result
    timings.setdefault("save_s", 0.0)
    with torch.cuda.stream(stream):
        gathered = gather_summaries(table, [b[1] - b[0] for b in bounds])
    stream.synchronize()
    fields = summary_to_fields_multi(gathered, md) if multi else summary_to_fields(gathered)
    if write_summary and rank == 0:
        mask = np.zeros(z_all.size, dtype=bool)
        mask[sel] = True
        stem = f"processed_qsos_multi_meanflux{test_set_name}" if multi else f"processed_qsos_{test_set_name}"
        io.savemat73(os.path.join(out_dir, stem + "_summary.mat"),
                     dict(test_ind=mask.reshape(-1, 1), **{k_: v for k_, v in fields.items()}))
    timings["total_s"] = time.perf_counter() - t_start
    return dict(fields=fields, block=bounds[rank], chunk=state["chunk"], selected=sel, timings=timings)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--preloaded", required=True, help="preloaded_qsos.mat (-v7.3)")
    ap.add_argument("--catalog", required=True, help="catalog.mat")
    ap.add_argument("--learned", required=True, help="learned_qso_model_*.mat")
    ap.add_argument("--samples", required=True, help="dla_samples.mat")
    ap.add_argument("--prior", required=True,
                    help=".mat/.npz with the training release's z_qsos and dla_ind (after api.prepare_prior)")
    ap.add_argument("--out", required=True, help="output directory for the chunk files")
    ap.add_argument("--name", default="dr12q", help="test_set_name")
    ap.add_argument("--multi", action="store_true", help="multi-DLA driver (process_qsos_multiple_dlas_meanflux)")
    ap.add_argument("--z-lls", type=float, default=None)
    ap.add_argument("--z-dla", type=float, default=None)
    ap.add_argument("--max-dlas", type=int, default=4)
    ap.add_argument("--batch", type=int, default=None, help="quasars per HBM-resident batch")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--timeout-min", type=float, default=30.0,
                    help="process-group timeout: how long a rank waits in a collective for a rank that died "
                         "without reporting (a rank that RAISES is agreed on at once)")
    args = ap.parse_args(argv)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import datetime
        kw = dict(device_id=torch.device("cuda", local_rank)) if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, timeout=datetime.timedelta(minutes=args.timeout_min), **kw)
    try:
        if args.prior.endswith(".npz"):
            pr = np.load(args.prior)
            prior = dict(z_qsos=pr["z_qsos"], dla_ind=pr["dla_ind"])
        else:
            m = io._load_mat(args.prior, ("z_qsos", "dla_ind"))
            prior = dict(z_qsos=np.asarray(m["z_qsos"]).reshape(-1), dla_ind=np.asarray(m["dla_ind"]).reshape(-1) != 0)
        params = MultiParameters(max_dlas=args.max_dlas) if args.multi else Parameters()
        res = run(args.preloaded, args.catalog, args.learned, args.samples, args.out, args.name,
                  prior_catalog=prior, multi=args.multi, params=params, Z_lls=args.z_lls, Z_dla=args.z_dla,
                  device=local_rank, max_quasars_per_batch=args.batch)
        print(f"rank {int(os.environ.get('RANK', '0'))}: quasars [{res['block'][0]}, {res['block'][1]}) -> {res['chunk']}",
              flush=True)
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
