"""Flat parameter / gradient arena.

The hot path owns ~100 small parameter tensors (6 MB).  Left to autograd, every backward pass allocates and
zero-fills a gradient per program call, AccumulateGrad adds the coarse- and fine-level contributions of the
shared modules tensor by tensor, the optimizer walks a tensor list and data parallelism has to gather the
gradients into a bucket first: ~80 tiny launches per step next to ~20 real ones.  `ParamArena` lays all
parameters out in ONE fp32 buffer and all gradients in another (each `p.data` / `p.grad` becomes a view), so that

* the weight-gradient kernel and the embedding backward accumulate straight into `arena.grad`
  (functional._ProgramFn / _EmbedFn detect the arena and return no per-parameter gradients),
* `arena.zero_grad()` is one fill, the optimizer steps one tensor (`arena.flat_param`),
* `arena.all_reduce_mean()` is one RCCL all-reduce on the gradient buffer itself — no bucket copies.

`p.grad` keeps PyTorch's meaning (sum of everything backpropagated since the last zero).  Calling
`optimizer.zero_grad(set_to_none=True)` on the individual parameters detaches them from the arena; the kernels
then fall back to returning gradients through autograd.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

_ALIGN = 4  # floats: every tensor starts on a 16-byte boundary


class ParamArena:
    def __init__(self, params: Iterable[torch.nn.Parameter]):
        seen, plist = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                plist.append(p)
        if not plist:
            raise ValueError("ParamArena needs at least one trainable parameter")
        dev = plist[0].device
        for p in plist:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("ParamArena holds fp32 parameters of one device")
        self.params: List[torch.nn.Parameter] = plist
        self.offsets: List[int] = []
        n = 0
        for p in plist:
            self.offsets.append(n)
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = n
        self.data = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(plist, self.offsets):
                self.data[o:o + p.numel()].copy_(p.data.reshape(-1))
                old_grad = p.grad
                p.data = self.data[o:o + p.numel()].view(p.shape)
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
                if old_grad is not None:
                    p.grad.copy_(old_grad)
                p._hn_arena = (self, o)
        # what the optimizer steps: one tensor whose gradient is the gradient buffer
        self.flat_param = torch.nn.Parameter(self.data)
        self.flat_param.grad = self.grad
        self._bumps = 0

    # ---- queries used by the kernels' autograd wrappers -----------------------------------------
    def version(self) -> int:
        """Changes whenever the parameter values were updated through the arena (in-place optimizer step)."""
        return self.flat_param._version + self.data._version + self._bumps

    def bump(self):
        """Called by updates torch cannot see (the HIP Adam kernel writes the buffer directly)."""
        self._bumps += 1

    def attached(self, p: torch.nn.Parameter) -> Optional[int]:
        """Offset (floats) of p's gradient inside `grad`, or None if p.grad is no longer the arena view."""
        tag = getattr(p, "_hn_arena", None)
        if tag is None or tag[0] is not self or p.grad is None:
            return None
        if p.grad.data_ptr() != self.grad.data_ptr() + 4 * tag[1]:
            return None
        return tag[1]

    @staticmethod
    def lookup(params: Sequence[torch.nn.Parameter]) -> Optional[Tuple["ParamArena", List[int]]]:
        """(arena, gradient offsets) if EVERY parameter is attached to one arena, else None."""
        if not params:
            return None
        tag = getattr(params[0], "_hn_arena", None)
        if tag is None:
            return None
        arena = tag[0]
        offs = []
        for p in params:
            o = arena.attached(p)
            if o is None:
                return None
            offs.append(o)
        return arena, offs

    # ---- training-loop side -----------------------------------------------------------------------
    def _complete_grad(self):
        """An optimizer may be holding this arena's reduce launch back (optim.ArenaAdam(fuse_reduce=True)): whoever needs
        the complete gradient buffer first runs it."""
        from . import machine
        machine.flush_pending_reduce(self.grad)

    def zero_grad(self):
        self._complete_grad()
        self.grad.zero_()

    def all_reduce_sum(self, group=None, force: bool = False):
        """SUM the gradient buffer across ranks: one all-reduce, in place.  The 1/world of the mean is applied by the
        consumer (ArenaAdam's grad_scale: no separate division launch).  `force`: issue the collective even in a
        one-rank group (RCCL runs it as a copy) — what lets a one-GPU box exercise the captured all-reduce."""
        if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
            self._complete_grad()
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)

    def all_reduce_mean(self, group=None):
        """Average the gradient buffer across ranks: one all-reduce, in place (for optimizers without grad_scale)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            self._complete_grad()
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
            self.grad.div_(dist.get_world_size(group))
