"""Data parallelism over rays: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm) over xGMI.

The reference trains with Lightning `ddp_sharded` (train.py:225-229): rays are sharded by the sampler and
gradients reduced.  Here rays are sharded contiguously, the model is replicated (6 MB) and one step performs
exactly one all-reduce of a single flat fp32 gradient bucket (parameters that received no gradient — e.g.
`nerf_embed` when GLO tables are shared — contribute zeros) and, for evaluation, one all-gather of pixels.
Both messages are latency-bound on xGMI, so they are single collectives, not per-parameter buckets.
"""
from __future__ import annotations

import contextlib
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


class GradSync:
    """The gradient all-reduce of one training step, overlapped with the tail of backward (SURVEY.md §5 'overlap with
    the tail of backward'; what Lightning's DDP bucketing does for the reference, train.py:225-229).

    The last kernel of a backward pass is the batched weight-gradient launch, which produces every dW at once — there
    is nothing behind it to overlap with.  So that launch is split in two (machine.WGRAD_SPLIT_OFFSET): the jobs of
    the template networks (the tail of the gradient buffer, ~85 % of its bytes) run at the end of backward; the jobs
    of the warp field and the hyper sheet are held back and run WHILE the first bucket is being all-reduced on RCCL's
    stream; their (small, latency-bound) all-reduce follows.

        sync = GradSync(arena, model)              # picks the split from the model's parameter layout
        with sync.splitting():
            loss = ...; loss.backward()            # bucket 0 launched at the end of backward, bucket 1 held
        sync.reduce(functional.flush_held_wgrads)  # or the replay of a graph that captured that call
        optimizer.step()                           # ArenaAdam(grad_scale=1/world)

    Without a usable split (no `nerf_mlps_*` block at the tail of the arena) or with overlap=False it is the single
    in-place all-reduce of the whole buffer.

    MEASURED COST (one MI355X, config 2, tools/held_probe.py): the single batched weight-gradient launch takes
    0.72-0.75 ms; split at the template networks it takes 0.65 + 0.28 ms, split at the fine-level template 0.48 +
    0.47 ms — each launch pays its own ramp and tail (jobs stream ~5 MB each, ~0.2 ms), so splitting costs ~0.2 ms,
    which is more than a 6 MB all-reduce over xGMI is expected to take.  TrainStep and bench.py therefore default to
    the single all-reduce; the overlap is there for topologies where the measured all-reduce exceeds that."""

    def __init__(self, arena, model: Optional[torch.nn.Module] = None, group=None, overlap: bool = True,
                 tail_prefix: str = "nerf_mlps_"):
        self.arena, self.group = arena, group
        self.split = self._template_offset(arena, model, tail_prefix) if (overlap and model is not None) else None

    @staticmethod
    def _template_offset(arena, model, tail_prefix: str = "nerf_mlps_") -> Optional[int]:
        """Offset (floats) where the parameters named `tail_prefix*` (default: the template networks) start, if they
        are exactly the tail of the arena."""
        tail = [p for name, p in model.named_parameters() if name.startswith(tail_prefix) and p.requires_grad]
        offs = [arena.attached(p) for p in tail]
        if not tail or any(o is None for o in offs):
            return None
        split = min(offs)
        ids = {id(p) for p in tail}
        for p, o in zip(arena.params, arena.offsets):
            if (o >= split) != (id(p) in ids):
                return None
        return split if 0 < split < arena.numel else None

    @contextlib.contextmanager
    def splitting(self):
        """Wrap the forward+backward passes of the step: backward passes that END inside hold their bucket-1 jobs
        (machine.WGRAD_SPLIT_OFFSET is a process-wide switch, so it is only set for the duration)."""
        from . import machine
        old = machine.WGRAD_SPLIT_OFFSET
        machine.WGRAD_SPLIT_OFFSET = self.split
        try:
            yield self
        finally:
            machine.WGRAD_SPLIT_OFFSET = old

    def reduce(self, run_held=None, force: bool = False):
        """Call after forward+backward.  `run_held` launches the held weight-gradient jobs (eagerly or as a graph
        replay); it is called even when there is nothing to overlap with, so the gradients are always complete.
        `force`: issue the all-reduce in a one-rank group too (ParamArena.all_reduce_sum)."""
        if self.split is None:
            if run_held is not None:
                run_held()
            self.arena.all_reduce_sum(self.group, force=force)
            return
        if run_held is None:
            from . import functional
            if functional.held_wgrads():
                raise RuntimeError("GradSync.reduce: weight-gradient jobs are being held back (splitting() was "
                                   "active during backward) but no `run_held` was given to launch them")
        g = self.arena.grad
        h0 = dist.all_reduce(g[self.split:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if run_held is not None:
            run_held()
        h1 = dist.all_reduce(g[:self.split], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        h0.wait()
        h1.wait()


def collective_capturable(group=None) -> bool:
    """True when a collective on `group` can be recorded into a HIP graph: RCCL ("nccl") enqueues a kernel on the
    current stream, which torch captures like any other launch; gloo stages through host memory (a device-to-host copy
    and a host wait: illegal inside a capture)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return str(dist.get_backend(group)).lower() == "nccl"
