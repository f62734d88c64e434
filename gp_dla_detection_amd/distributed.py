"""Multi-GPU driver: shard quasars across ranks, one RCCL all-gather of the posterior table.

The reference's only multi-node mechanism is "run process_qsos on disjoint ``test_ind`` slices as
separate batch jobs, then concatenate the .mat chunks along the quasar axis"
(CDDF_analysis/sbatch_reunion.py:13-63).  Quasars are independent (serial outer loop with no
carried state, process_qsos.m:88), so the MI355X-native equivalent is: one process per GPU, each
sweeping a contiguous block of quasars balanced by pixel count, no communication during compute,
and one ``all_gather`` of the small per-quasar summary row (12 fp64 = 96 B; 15.6 MB for DR12Q's
162 861 quasars) over RCCL/xGMI at the end.  ``sample_log_likelihoods_dla`` (80 KB per quasar) is
NOT gathered: each rank keeps its shard, as the reference keeps per-job chunks.
"""
from __future__ import annotations

import numpy as np

SUMMARY_COLUMNS = ("min_z_dlas", "max_z_dlas", "log_priors_no_dla", "log_priors_dla",
                   "log_likelihoods_no_dla", "log_likelihoods_dla", "log_posteriors_no_dla",
                   "log_posteriors_dla", "model_posteriors_no_dla", "model_posteriors_dla",
                   "p_no_dlas", "p_dlas")


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
    """All-gather the per-quasar summary rows of every rank into the full [nq_total, 12] table, in
    rank (= quasar) order.  ``local_table``: torch tensor [nq_local, 12] on this rank's device
    (``Batch.summary_tensor()``); works on CPU tensors with gloo as well.  ``counts``: rows per
    rank if already known (skips the size exchange)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_table
    world = dist.get_world_size(group)
    ncol = local_table.shape[1]
    if counts is None:
        mine = torch.tensor([local_table.shape[0]], dtype=torch.int64, device=local_table.device)
        sizes = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(sizes, mine, group=group)
        counts = [int(s.item()) for s in sizes]
    if len(set(counts)) == 1:
        out = torch.empty((world * counts[0], ncol), dtype=local_table.dtype, device=local_table.device)
        dist.all_gather_into_tensor(out, local_table.contiguous(), group=group)
        return out
    width = max(counts)
    padded = torch.zeros((width, ncol), dtype=local_table.dtype, device=local_table.device)
    padded[: local_table.shape[0]] = local_table
    out = torch.empty((world * width, ncol), dtype=local_table.dtype, device=local_table.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * width: r * width + counts[r]] for r in range(world)], dim=0)


def summary_to_fields(table) -> dict:
    """Split a gathered [nq, 12] table into the reference's output variables
    (process_qsos.m:236-244)."""
    t = table.detach().cpu().numpy() if hasattr(table, "detach") else np.asarray(table)
    out = {name: t[:, i].copy() for i, name in enumerate(SUMMARY_COLUMNS)}
    out["model_posteriors"] = np.stack([out.pop("model_posteriors_no_dla"),
                                        out.pop("model_posteriors_dla")], axis=1)
    return out


def process_qsos_sharded(model, samples, spectra, log_priors, params=None, device=None):
    """process_qsos over every quasar of ``spectra`` with the work split across the ranks of the
    default process group.  Every rank passes the same full list; returns (gathered summary
    fields, this rank's (lo, hi) block, this rank's sample_log_likelihoods_dla)."""
    import torch
    import torch.distributed as dist

    from .api import Context

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if device is None:
        device = torch.cuda.current_device()
    sizes = [np.asarray(s["wavelengths"]).size for s in spectra]
    bounds = shard_bounds(sizes, world)
    lo, hi = bounds[rank]
    ctx = Context(device, params)
    try:
        ctx.set_model(model)
        ctx.set_samples(samples)
        batch = ctx.upload(spectra[lo:hi], np.asarray(log_priors[0])[lo:hi],
                           np.asarray(log_priors[1])[lo:hi])
        try:
            batch.process()
            ctx.synchronize()
            table = gather_summaries(batch.summary_tensor(), [b[1] - b[0] for b in bounds])
            fields = summary_to_fields(table)
            local = batch.download()["sample_log_likelihoods_dla"]
        finally:
            batch.close()
    finally:
        ctx.close()
    return fields, (lo, hi), local
