"""Multi-GPU driver: shard quasars across ranks, one RCCL all-gather of the posterior table.

The reference's only multi-node mechanism is "run process_qsos on disjoint ``test_ind`` slices as
separate batch jobs, then concatenate the .mat chunks along the quasar axis"
(CDDF_analysis/sbatch_reunion.py:13-63; every key is concatenated, the multi-DLA ones included,
:29-55).  Quasars are independent (serial outer loop with no carried state, process_qsos.m:88),
so the MI355X-native equivalent is: one process per GPU, each sweeping a contiguous block of
quasars balanced by pixel count, no communication during compute, and one ``all_gather`` of the
small per-quasar summary row over RCCL/xGMI at the end:

* single-DLA: 15 fp64 = 120 B per quasar (19.5 MB for DR12Q's 162 861 quasars);
* multi-DLA: ``14 + 4 max_dlas + 3 max_dlas^2`` fp64 = 78 for max_dlas = 4 (624 B; 102 MB).

The per-sample tables (``sample_log_likelihoods_dla``: 80 KB per quasar and model,
``base_sample_inds``) are NOT gathered: each rank keeps its shard, as the reference keeps per-job
chunk files.  A rank only needs its own block of spectra: pass a loader instead of the full list.
"""
from __future__ import annotations

import numpy as np

SUMMARY_COLUMNS = ("min_z_dlas", "max_z_dlas", "log_priors_no_dla", "log_priors_dla",
                   "log_likelihoods_no_dla", "log_likelihoods_dla", "log_posteriors_no_dla",
                   "log_posteriors_dla", "model_posteriors_no_dla", "model_posteriors_dla",
                   "p_no_dlas", "p_dlas", "MAP_inds", "MAP_z_dlas", "MAP_log_nhis")


def shard_bounds(pixel_counts, world_size: int):
    """Contiguous blocks [lo, hi) per rank, balanced by the sum of pixel counts (the sweep's cost
    is proportional to n per quasar), every rank non-empty when there are enough quasars."""
    counts = np.asarray(pixel_counts, dtype=np.float64)
    nq = counts.size
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    cum = np.concatenate([[0.0], np.cumsum(counts)])
    targets = cum[-1] * np.arange(1, world_size) / world_size
    cuts = np.searchsorted(cum, targets, side="left")
    cuts = np.clip(cuts, 0, nq)
    edges = np.concatenate([[0], cuts, [nq]]).astype(np.int64)
    if nq >= world_size:  # every rank gets at least one quasar
        for r in range(1, world_size):
            edges[r] = min(max(edges[r], edges[r - 1] + 1), nq - (world_size - r))
    edges = np.maximum.accumulate(edges)
    return [(int(edges[r]), int(edges[r + 1])) for r in range(world_size)]


def gather_summaries(local_table, counts=None, group=None):
    """All-gather the per-quasar summary rows of every rank into the full [nq_total, ncol] table,
    in rank (= quasar) order.  ``local_table``: torch tensor [nq_local, ncol] on this rank's device
    (``Batch.summary_tensor()``); a rank whose block is empty passes a [0, ncol] tensor.  With the
    gloo backend (CPU rehearsal of the multi-process path) device tensors are staged through host
    memory.  ``counts``: rows per rank if already known (skips the size exchange)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_table
    if dist.get_backend(group) == "gloo" and local_table.is_cuda:
        local_table = local_table.cpu()
    world = dist.get_world_size(group)
    ncol = local_table.shape[1]
    if counts is None:
        mine = torch.tensor([local_table.shape[0]], dtype=torch.int64, device=local_table.device)
        sizes = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(sizes, mine, group=group)
        counts = [int(s.item()) for s in sizes]
    if len(set(counts)) == 1:
        out = torch.empty((world * counts[0], ncol), dtype=local_table.dtype, device=local_table.device)
        if counts[0]:
            dist.all_gather_into_tensor(out, local_table.contiguous(), group=group)
        return out
    width = max(counts)
    padded = torch.zeros((width, ncol), dtype=local_table.dtype, device=local_table.device)
    padded[: local_table.shape[0]] = local_table
    out = torch.empty((world * width, ncol), dtype=local_table.dtype, device=local_table.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * width: r * width + counts[r]] for r in range(world)], dim=0)


class ShardFailure(RuntimeError):
    """Some rank of a sharded run failed; raised on EVERY rank (see :func:`agree_on_failure`)."""


def agree_on_failure(error: BaseException | None, group=None, what: str = "sharded run") -> None:
    """Every rank calls this once its local work is over -- successfully (``error`` None) or not --
    and BEFORE the collective that needs everybody's results: one all-reduce (MAX) of a failure flag.
    If any rank failed, every rank raises -- the failing one its own exception, the others
    :class:`ShardFailure` -- instead of waiting in the gather for a rank that will never arrive
    (the reference's batch jobs are independent processes, CDDF_analysis/sbatch_reunion.py:13-27:
    there a failed job simply leaves its chunk missing; ranks of one process group must agree).
    A rank that dies without reaching this point is covered by the group's ``timeout``."""
    import torch
    import torch.distributed as dist

    failed = error is not None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        on_gpu = dist.get_backend(group) != "gloo" and torch.cuda.is_available()
        flag = torch.tensor([1.0 if failed else 0.0], dtype=torch.float64,
                            device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        anyone = bool(flag.item() > 0)
    else:
        anyone = failed
    if failed:
        raise error
    if anyone:
        raise ShardFailure(f"{what}: another rank failed; this rank's results are discarded")


def summary_to_fields(table) -> dict:
    """Split a gathered [nq, 15] table into the reference's output variables
    (process_qsos.m:236-244) plus the MAP columns of generate_ascii_catalog.m:73-80."""
    t = table.detach().cpu().numpy() if hasattr(table, "detach") else np.asarray(table)
    out = {name: t[:, i].copy() for i, name in enumerate(SUMMARY_COLUMNS)}
    out["model_posteriors"] = np.stack([out.pop("model_posteriors_no_dla"),
                                        out.pop("model_posteriors_dla")], axis=1)
    return out


def summary_to_fields_multi(table, max_dlas: int) -> dict:
    """Split a gathered multi-DLA table (layout: GPDLA_SUMMARY_COLS_MULTI in include/gpdla.h) into
    the variables process_qsos_multiple_dlas_meanflux.m:498-510 saves (per-sample arrays aside)."""
    t = table.detach().cpu().numpy() if hasattr(table, "detach") else np.asarray(table)
    md, nq = int(max_dlas), t.shape[0]
    if t.shape[1] != 14 + 4 * md + 3 * md * md:
        raise ValueError(f"table has {t.shape[1]} columns, max_dlas = {md} needs {14 + 4 * md + 3 * md * md}")
    pos = [0]

    def take(n, shape=None):
        a = t[:, pos[0]: pos[0] + n].copy()
        pos[0] += n
        return a[:, 0] if shape is None else a.reshape((nq,) + shape)

    out = {}
    out["min_z_dlas"], out["max_z_dlas"] = take(1), take(1)
    out["log_priors_no_dla"], out["log_priors_lls"] = take(1), take(1)
    out["log_priors_dla"] = take(md, (md,))
    out["log_likelihoods_no_dla"], out["log_likelihoods_lls"] = take(1), take(1)
    out["log_likelihoods_dla"] = take(md, (md,))
    out["log_posteriors_no_dla"], out["log_posteriors_lls"] = take(1), take(1)
    out["log_posteriors_dla"] = take(md, (md,))
    out["model_posteriors"] = take(2 + md, (2 + md,))
    out["p_no_dlas"], out["p_lls"], out["p_dlas"] = take(1), take(1), take(1)
    for name in ("MAP_z_dlas", "MAP_log_nhis", "MAP_inds"):
        out[name] = take(md * md, (md, md))
    out["all_exceptions"] = take(1)
    return out


def _world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def _block(spectra, lo, hi):
    """This rank's spectra: a slice of the full list, or whatever the loader returns for [lo, hi)."""
    if callable(spectra):
        block = list(spectra(lo, hi))
        if len(block) != hi - lo:
            raise ValueError(f"loader returned {len(block)} spectra for block [{lo}, {hi})")
        return block
    return list(spectra[lo:hi])


def _pixel_counts(spectra, pixel_counts):
    if pixel_counts is not None:
        return np.asarray(pixel_counts)
    if callable(spectra):
        raise ValueError("a spectra loader needs pixel_counts (one entry per quasar)")
    return np.array([np.asarray(s["wavelengths"]).size for s in spectra])


def run_sharded(bounds, rank, ncol, sweep_block, group=None):
    """The sharding skeleton shared by both drivers: ``sweep_block(lo, hi)`` returns this rank's
    (summary tensor [hi-lo, ncol], local per-sample results); a rank with an empty block skips
    the sweep and contributes a [0, ncol] table, so the collective is reached by every rank.
    Returns (gathered table, (lo, hi), local results or None)."""
    import torch
    lo, hi = bounds[rank]
    error = None
    table, local = torch.empty((0, ncol), dtype=torch.float64), None
    if hi > lo:
        try:
            table, local = sweep_block(lo, hi)
        except Exception as e:  # agreed on below: no rank is left waiting in the gather
            error = e
    if len(bounds) > 1:
        agree_on_failure(error, group)
    elif error is not None:
        raise error
    counts = [b[1] - b[0] for b in bounds]
    if len(bounds) > 1 and hi == lo:  # put the empty table where the other ranks' tables live
        import torch.distributed as dist
        if dist.get_backend(group) != "gloo" and torch.cuda.is_available():
            table = table.to(torch.device("cuda", torch.cuda.current_device()))
    return gather_summaries(table, counts, group), (lo, hi), local


def sweep_block_pipelined(model, samples, block_spectra, priors, params, device, multi, first_index=0,
                          base_sample_inds=None, max_quasars_per_batch=None, pipeline_slots=3):
    """One rank's block of a sharded run, swept in bounded HBM-resident batches through
    api.run_pipeline (upload i+1 / sweep i / download i-1).  Each batch's summary rows are copied,
    behind its sweep on the same stream, into the rank's device table -- what the RCCL all-gather
    then reads.  Returns (table [n, ncol] on the device, host results of the block)."""
    import torch

    from . import _lib
    from .api import Batch, Context, batch_blocks, default_batch_size, prefault, run_pipeline

    n = len(block_spectra)
    S = np.asarray(samples["offset_samples"]).size
    k = np.asarray(model["M"]).shape[1]
    md = params.max_dlas if multi else 0
    ncol = _lib.summary_cols_multi(md) if multi else _lib.SUMMARY_COLS
    stream = torch.cuda.Stream(device=device)
    table = torch.empty((n, ncol), dtype=torch.float64, device=f"cuda:{device}")
    if max_quasars_per_batch is None:
        longest = max(np.asarray(s["wavelengths"]).size for s in block_spectra)
        max_quasars_per_batch = default_batch_size(n, longest, k, S, pipeline_slots, multi_models=(md + 1) if multi else 0)
    blocks = batch_blocks(n, max_quasars_per_batch)
    local = Batch.empty_results_multi(n, md, S) if multi else Batch.empty_results(n, S)
    ctx = Context(device, params, stream=stream)

    def inputs(i):
        b0, b1 = blocks[i]
        return (block_spectra[b0:b1],) + tuple(np.asarray(p)[b0:b1] for p in priors)

    copied = [None] * len(blocks)

    def process(i, batch):
        b0, b1 = blocks[i]
        with torch.cuda.stream(stream):
            if multi:
                ctx.set_first_quasar_index(first_index + b0)
                batch.process_multi(None if base_sample_inds is None else np.asarray(base_sample_inds)[b0:b1])
            else:
                batch.process()
            table[b0:b1].copy_(batch.summary_tensor())
            # the copy reads the batch's summary rows AFTER the library recorded the batch's own
            # "done" event: the slot must not be re-filled before it has run (download waits for it)
            copied[i] = torch.cuda.Event()
            copied[i].record(stream)

    def download(i, batch):
        b0 = blocks[i][0]
        (batch.download_multi(True, local, b0) if multi else batch.download(True, local, b0))
        copied[i].synchronize()

    try:
        ctx.set_model(model)
        ctx.set_samples(samples)
        run_pipeline(ctx, len(blocks), inputs, process, download, pipeline_slots,
                     warm=lambda: prefault(local["sample_log_likelihoods_dla"]))
        stream.synchronize()
    finally:
        ctx.close()
    return table, local


def process_qsos_sharded(model, samples, spectra, log_priors, params=None, device=None,
                         pixel_counts=None, max_quasars_per_batch=None):
    """process_qsos over every quasar with the work split across the ranks of the default process
    group.  ``spectra``: the full list (every rank passes the same one), or a loader
    ``f(lo, hi) -> list`` together with ``pixel_counts`` so that a rank reads only its block.
    ``log_priors = (no_dla, dla)`` for ALL quasars.  A rank sweeps its block in bounded, pipelined
    batches (:func:`sweep_block_pipelined`).  Returns (gathered summary fields for all quasars,
    this rank's (lo, hi) block, this rank's sample_log_likelihoods_dla [hi-lo, S] or None for an
    empty block)."""
    import torch

    from . import _lib
    from .parameters import Parameters

    world, rank = _world()
    if device is None:
        device = torch.cuda.current_device()
    p = params or Parameters()
    bounds = shard_bounds(_pixel_counts(spectra, pixel_counts), world)
    lp_no, lp_dla = (np.asarray(x, dtype=np.float64) for x in log_priors)

    def sweep_block(lo, hi):
        table, local = sweep_block_pipelined(model, samples, _block(spectra, lo, hi), (lp_no[lo:hi], lp_dla[lo:hi]),
                                             p, device, False, max_quasars_per_batch=max_quasars_per_batch)
        return table, local["sample_log_likelihoods_dla"]

    table, block, local = run_sharded(bounds, rank, _lib.SUMMARY_COLS, sweep_block)
    return summary_to_fields(table), block, local


def process_qsos_multiple_dlas_meanflux_sharded(model, samples, spectra, log_priors, params=None,
                                                device=None, pixel_counts=None,
                                                base_sample_inds=None, max_quasars_per_batch=None):
    """The multi-DLA driver split across ranks.  ``log_priors = (no_dla [nq], lls [nq], dla [nq,
    max_dlas])`` for ALL quasars.  Each rank's batches are told the global index of their first quasar
    (``first_quasar_index``), which keys the Philox stream of the weighted resampling (multi
    :467-472), so the indices drawn -- and with them every result -- equal those of an unsharded
    run.  ``base_sample_inds`` (optional, [nq, max_dlas-1, S] for ALL quasars) replays supplied
    indices instead.  Returns (gathered summary fields, (lo, hi), this rank's per-sample results:
    dict with sample_log_likelihoods_dla [hi-lo, max_dlas, S], sample_log_likelihoods_lls,
    base_sample_inds; None for an empty block)."""
    import torch

    from . import _lib
    from .parameters import MultiParameters

    p = params or MultiParameters()
    world, rank = _world()
    if device is None:
        device = torch.cuda.current_device()
    bounds = shard_bounds(_pixel_counts(spectra, pixel_counts), world)
    lp_no, lp_lls, lp_dla = (np.asarray(x, dtype=np.float64) for x in log_priors)
    lp_dla = lp_dla.reshape(lp_no.size, p.max_dlas)

    def sweep_block(lo, hi):
        table, res = sweep_block_pipelined(
            model, samples, _block(spectra, lo, hi), (lp_no[lo:hi], lp_dla[lo:hi], lp_lls[lo:hi]), p, device, True,
            first_index=p.first_quasar_index + lo,
            base_sample_inds=None if base_sample_inds is None else np.asarray(base_sample_inds)[lo:hi],
            max_quasars_per_batch=max_quasars_per_batch)
        return table, {key: res[key] for key in ("sample_log_likelihoods_dla", "sample_log_likelihoods_lls",
                                                 "base_sample_inds")}

    table, block, local = run_sharded(bounds, rank, _lib.summary_cols_multi(p.max_dlas), sweep_block)
    return summary_to_fields_multi(table, p.max_dlas), block, local
