"""Data parallelism for the hot path: one process per GPU, clips sharded over ranks, RCCL over xGMI.

The reference relies on Lightning's DDPStrategy (train_COSKAD.py:75-78): DistributedSampler sharding,
bucketed gradient all-reduce, rank-0 buffer broadcast.  Here the model is 0.96 MB, so the whole gradient
is ONE flat fp32 buffer and ONE all-reduce per step (latency-bound, 2*(W-1)/W*0.96 MB on the wire), and
the running centre statistics (19 floats) are all-reduced when the centre is refreshed -- which makes the
W-GPU centre equal the single-process one (the reference lets rank 0's centre win).

These helpers are device-agnostic (`nccl` = RCCL on the GPUs, `gloo` in the CPU tests).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

Tensor = torch.Tensor


def world_size(group=None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def rank(group=None) -> int:
    return dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0


def shard_indices(n: int, rank_: int, world: int) -> torch.Tensor:
    """DistributedSampler(shuffle=False) semantics: rank r takes clips r, r+W, r+2W, ...; the tail is padded
    by wrapping so every rank gets ceil(n/W) clips."""
    per = (n + world - 1) // world
    idx = torch.arange(rank_, rank_ + per * world, world)
    return idx % n


def allreduce_mean_(flat_grad: Tensor, group=None) -> Tensor:
    """In-place mean over ranks of the flat gradient buffer (DDP semantics: SUM then / W)."""
    w = world_size(group)
    if w > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
        flat_grad.div_(w)
    return flat_grad


def allreduce_grads_mean_(params, group=None) -> None:
    """Mean over ranks of the `.grad` of every parameter as ONE flat bucket (one collective per step instead of one
    per parameter tensor: the model is ~1 MB, the cost is latency).  The bucket is laid out over EVERY parameter that
    requires a gradient -- zeros where a rank has none (an unused branch on a ragged shard) -- so that its length and
    the meaning of every slot are the same on all ranks; a parameter unused on this rank receives the others' mean."""
    w = world_size(group)
    if w == 1:
        return
    ps = [p for p in params if p.requires_grad]
    if not ps:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(w)
    off = 0
    for p in ps:
        k = p.numel()
        if p.grad is None:
            p.grad = flat[off:off + k].view(p.shape).clone()
        else:
            p.grad.copy_(flat[off:off + k].view(p.shape))
        off += k


def allreduce_sum_(stats: Tensor, group=None) -> Tensor:
    """In-place sum over ranks of additive statistics (centre sums / counts, gyromidpoint sums)."""
    if world_size(group) > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats


def broadcast_(t: Tensor, src: int = 0, group=None) -> Tensor:
    if world_size(group) > 1:
        dist.broadcast(t, src=src, group=group)
    return t


def broadcast_module_(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Rank `src`'s parameters and buffers to every rank -- what Lightning's DDP wrap does before the first step
    (train_COSKAD.py:75-78): without it every process would start from its own random initialisation and only the
    gradients would be shared.  Floating-point tensors travel as ONE flat buffer, integer buffers
    (`num_batches_tracked`) as another."""
    if world_size(group) == 1:
        return
    with torch.no_grad():
        ts = [p for p in module.parameters()] + [b for b in module.buffers()]
        for is_float in (True, False):
            sel = [t for t in ts if t.is_floating_point() == is_float and t.numel() > 0]
            if not sel:
                continue
            dt = torch.float32 if is_float else torch.int64
            flat = torch.cat([t.detach().reshape(-1).to(dt) for t in sel])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in sel:
                k = t.numel()
                t.copy_(flat[off:off + k].view(t.shape).to(t.dtype))
                off += k


def broadcast_buffers_(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Rank `src`'s buffers (BatchNorm running statistics, centre, inverse covariance) to every rank -- DDP's
    `broadcast_buffers=True`, which Lightning's DDPStrategy keeps on."""
    if world_size(group) == 1:
        return
    with torch.no_grad():
        for is_float in (True, False):
            sel = [b for b in module.buffers() if b.is_floating_point() == is_float and b.numel() > 0]
            if not sel:
                continue
            dt = torch.float32 if is_float else torch.int64
            flat = torch.cat([b.detach().reshape(-1).to(dt) for b in sel])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for b in sel:
                k = b.numel()
                b.copy_(flat[off:off + k].view(b.shape).to(b.dtype))
                off += k


def dedupe_rows(keys: Tensor):
    """Indices of the first occurrence of every distinct row of `keys` [N, K] (int64), in first-occurrence order:
    removes the wrap-padding duplicates that equal-length validation shards carry after a gather."""
    uniq, inv = torch.unique(keys, dim=0, return_inverse=True)
    first = torch.full((uniq.shape[0],), keys.shape[0], dtype=torch.int64, device=keys.device)
    first = first.scatter_reduce(0, inv, torch.arange(keys.shape[0], device=keys.device), reduce="amin", include_self=True)
    return torch.sort(first).values


def gather_rows(t: Tensor, group=None) -> Tensor:
    """all_gather of per-rank row blocks [N_r, ...] -> [sum N_r, ...] in rank order (validation: latents + metadata
    to score on every rank).  Ranks may hold different row counts (an r::W split of the last batch): blocks are
    padded to the largest count for the collective and trimmed afterwards."""
    w = world_size(group)
    if w == 1:
        return t
    home = t.device
    if t.is_cuda and dist.get_backend(group) == "gloo":
        t = t.cpu()                      # gloo's all_gather takes host tensors only (single-GPU rehearsals of the N-rank flow)
    out = _gather_rows(t, w, group)
    return out.to(home)


def _gather_rows(t: Tensor, w: int, group=None) -> Tensor:
    n = torch.tensor([t.shape[0]], device=t.device, dtype=torch.int64)
    counts = [torch.empty_like(n) for _ in range(w)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    pad = t.contiguous()
    if t.shape[0] < m:
        pad = torch.cat([pad, pad.new_zeros((m - t.shape[0],) + tuple(t.shape[1:]))], 0)
    out = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)
