"""Multi-GPU sharding of a sweep of independent GP fits.

A single fit is never split across GPUs (its factor panel is latency-bound and lives in one GPU's
HBM); a batch of windows -- one per Monte-Carlo trajectory / terrain segment -- is cut into
contiguous per-rank shards with no data-path collective.  The only exchange is one all-gather of
per-fit summaries (log marginal likelihood, max predictive sigma, status): a few KB, latency-bound,
RCCL over xGMI on GPUs ("nccl" backend) or gloo in the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

SUMMARY_FIELDS = ("logml", "max_sigma", "info")


def shard_range(n_fits: int, rank: int, world: int):
    """Contiguous block partition: ranks < n_fits % world get one extra fit.  Returns (start, stop)."""
    if not (0 <= rank < world) or n_fits < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_fits, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_summaries(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """local: (n_local, len(SUMMARY_FIELDS)) float64 on this rank's device.  Returns the (n_total, F)
    table in global fit order on every rank.  Shards may differ by one row, so rows are padded to the
    largest shard for the all_gather and trimmed afterwards."""
    if not (dist.is_available() and dist.is_initialized()):
        return local          # no process group: one rank owns everything (a world of ONE rank still runs the collective)
    world = dist.get_world_size()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    nmax = max(b - a for a, b in sizes)
    pad = torch.zeros((nmax, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([parts[r][: b - a] for r, (a, b) in enumerate(sizes)], 0)


def ensemble_stats(table: torch.Tensor):
    """Ensemble mean / variance of the gathered log marginal likelihoods and the failure count."""
    ok = table[:, 2] == 0
    lm = table[ok, 0]
    return {"n": int(table.shape[0]), "n_failed": int((~ok).sum()), "logml_mean": float(lm.mean()) if lm.numel() else 0.0,
            "logml_var": float(lm.var(unbiased=False)) if lm.numel() else 0.0,
            "max_sigma": float(table[ok, 1].max()) if lm.numel() else 0.0}
