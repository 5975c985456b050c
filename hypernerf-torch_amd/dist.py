"""Data parallelism over rays: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm) over xGMI.

The reference trains with Lightning `ddp_sharded` (train.py:225-229): rays are sharded by the sampler and
gradients reduced.  Here rays are sharded contiguously, the model is replicated (6 MB) and one step performs
exactly one all-reduce of a single flat fp32 gradient bucket (parameters that received no gradient — e.g.
`nerf_embed` when GLO tables are shared — contribute zeros) and, for evaluation, one all-gather of pixels.
Both messages are latency-bound on xGMI, so they are single collectives, not per-parameter buckets.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of n rays owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rays(rays: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_range(rays.shape[0], rank, world)
    return rays[lo:hi]


class GradBucket:
    """One flat fp32 buffer holding every parameter gradient of a module."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += p.numel()
        self.numel = n
        self.flat: Optional[torch.Tensor] = None

    def _ensure(self, device):
        if self.flat is None or self.flat.device != device:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)

    def all_reduce_mean(self, group=None):
        """Average gradients across ranks with a single all-reduce; missing grads count as zero."""
        if not self.params:
            return
        from .arena import ParamArena
        arena = ParamArena.lookup(self.params)
        if arena is not None and len(arena[0].params) == len(self.params) and \
                all(a is b for a, b in zip(arena[0].params, self.params)):
            arena[0].all_reduce_mean(group)      # the gradients already are one flat buffer: reduce it in place
            return
        self._ensure(self.params[0].device)
        self.flat.zero_()
        for p, o in zip(self.params, self.offsets):
            if p.grad is not None:
                self.flat[o:o + p.numel()].copy_(p.grad.reshape(-1))
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))
        for p, o in zip(self.params, self.offsets):
            g = self.flat[o:o + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)


def all_gather_pixels(x: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate equally sized per-rank pixel blocks (B, C) -> (world*B, C) with one all-gather."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return x
    world = dist.get_world_size(group)
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out
