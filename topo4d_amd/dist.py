"""
Multi-GPU driver for the render hot path: one process per GPU (torchrun), units sharded, losses gathered.

The reference is single-process / single-GPU; this is the one parallel axis the hot path has:
every (frame, view) render reads the same replicated Gaussians and its own camera, so units are independent
(SURVEY.md §8e).  Rank r of W takes units {u : u mod W == r}.  No data-path collective exists; the only exchange
is an all_gather of the per-view scalar photometric losses (a few floats per rank) — RCCL on GPU ("nccl" backend
is RCCL on ROCm), gloo in the CPU tests.  A process group of ONE rank takes the no-collective fast path like no process
group at all, unless T4D_FORCE_COLLECTIVES=1 is set: that is how the RCCL calls are exercised on a one-GPU test box
(tests/test_gpu_multirank.py).
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch


def _group_active() -> bool:
    """True when a collective has to run: a process group exists and has more than one rank (or T4D_FORCE_COLLECTIVES=1)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return False
    return dist.get_world_size() > 1 or os.environ.get("T4D_FORCE_COLLECTIVES") == "1"


def shard_units(n_units: int, rank: int, world: int) -> List[int]:
    """Round-robin shard: units rank, rank+world, ... (24 views over 8 ranks -> 3 views each)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_units, world))


def shard_sizes(n_units: int, world: int) -> List[int]:
    return [len(range(r, n_units, world)) for r in range(world)]


def gather_losses(local: torch.Tensor, n_units: int = None) -> torch.Tensor:
    """all_gather of per-unit scalar losses.  `local` is this rank's 1-D tensor (round-robin shard of n_units
    units; ranks may hold different counts).  Returns the losses of ALL units in unit order, on every rank."""
    import torch.distributed as dist
    if not _group_active():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    if local.is_cuda and dist.get_backend() != "nccl":        # gloo dry runs: stage through the host
        return gather_losses(local.cpu(), n_units).to(local.device)
    if n_units is None:                       # equal shards
        out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        # rank-major -> unit order (unit u lives on rank u % world at position u // world)
        return out.view(world, -1).t().reshape(-1)
    sizes = shard_sizes(n_units, world)
    m = max(sizes)
    padded = torch.zeros(m, dtype=local.dtype, device=local.device)
    padded[: local.numel()] = local
    out = torch.empty(world * m, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    out = out.view(world, m)
    res = torch.empty(n_units, dtype=local.dtype, device=local.device)
    for r in range(world):
        res[r::world] = out[r, : sizes[r]]
    return res


def gather_losses_async(local: torch.Tensor, out: torch.Tensor):
    """Equal-shard loss gather that does NOT block the compute stream: returns (out, work).  The collective is ordered
    after the work already enqueued on the current stream; kernels enqueued afterwards overlap with it.  `out` must hold
    world*local.numel() elements in rank-major order (out.view(world,-1).t() is unit order) and, like `local`, must not be
    rewritten before `work.wait()`."""
    import torch.distributed as dist
    if not _group_active():
        out[: local.numel()].copy_(local)
        return out, None
    if local.is_cuda and dist.get_backend() != "nccl":        # gloo dry runs: synchronous, through the host
        tmp = torch.empty(out.numel(), dtype=local.dtype)
        dist.all_gather_into_tensor(tmp, local.cpu().contiguous())
        out.copy_(tmp)
        return out, None
    work = dist.all_gather_into_tensor(out, local.contiguous(), async_op=True)
    return out, work


def all_reduce_grads(grads: Sequence[torch.Tensor]) -> None:
    """Optional data-parallel training step: sum the (already view-summed) parameter gradients over ranks.
    Changes the optimisation schedule versus train.py:661-673 (one Adam step per view) — see DESIGN.md."""
    import torch.distributed as dist
    if not _group_active():
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    o = 0
    for g in grads:
        g.copy_(flat[o: o + g.numel()].view_as(g))
        o += g.numel()


def band_bounds(n_rows: int, rank: int, world: int):
    """Contiguous row band [begin, end) of rank `rank` when `n_rows` image rows are split over `world` ranks (the first
    n_rows % world ranks take one row more): the unit the texture bake (BASELINE config 5) shards by."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_rows, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def gather_bands(band: torch.Tensor, n_rows: int) -> torch.Tensor:
    """all_gather of per-rank row bands ([rows_r, ...], band_bounds order) into the full [n_rows, ...] tensor on every rank.
    Bands may differ by one row; they are padded to the tallest for the collective."""
    import torch.distributed as dist
    if not _group_active():
        return band
    world = dist.get_world_size()
    if band.is_cuda and dist.get_backend() != "nccl":         # gloo dry runs: stage through the host
        return gather_bands(band.cpu(), n_rows).to(band.device)
    tallest = -(-n_rows // world)
    padded = torch.zeros((tallest,) + tuple(band.shape[1:]), dtype=band.dtype, device=band.device)
    padded[: band.shape[0]] = band
    out = torch.empty((world * tallest,) + tuple(band.shape[1:]), dtype=band.dtype, device=band.device)
    dist.all_gather_into_tensor(out, padded.contiguous())
    parts = []
    for r in range(world):
        b0, b1 = band_bounds(n_rows, r, world)
        parts.append(out[r * tallest: r * tallest + (b1 - b0)])
    return torch.cat(parts, 0)
