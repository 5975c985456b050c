"""Fused Adam on a ParamArena: one HIP kernel updates every parameter of the model and clears the
gradient buffer for the next step (`hn_adam_step`).  Same update rule and defaults as torch.optim.Adam, which is what
the reference's `get_optimizer` builds (utils/__init__.py:23-41), and `MultiStepLR` — its 'steplr' scheduler
(utils/__init__.py:43-46).

Everything a captured launch depends on lives ON THE DEVICE: the step counter and the hyper-parameters
[lr, beta1, beta2, eps, weight_decay, grad_scale].  `param_groups[0]` stays the user-facing source of truth (torch
schedulers write `lr` there); `sync_hyper()` uploads it when it changed — `step()` does that itself when it runs
eagerly, and whoever replays a graph that contains the step calls it before the replay (TrainStep does).
"""
from __future__ import annotations

import ctypes as C
from bisect import bisect_right
from typing import Sequence

import torch

from . import _lib as L
from .arena import ParamArena


class ArenaAdam:
    def __init__(self, arena: ParamArena, lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, zero_grad: bool = True, grad_scale: float = 1.0):
        L.require_gpu(arena.data)
        self.arena = arena
        self.param_groups = [{"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay}]
        self.grad_scale = float(grad_scale)        # 1 / world size when the gradients arrive SUM-all-reduced
        self.zero_grad_in_step = zero_grad
        dev = arena.data.device
        self.exp_avg = torch.zeros_like(arena.data)
        self.exp_avg_sq = torch.zeros_like(arena.data)
        self._step_words = torch.zeros(2, dtype=torch.float32, device=dev)      # [updates done, ticket counter]
        self.step_count = self._step_words[:1]
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        self._uploaded = None
        self.sync_hyper()

    def _hyper_values(self):
        g = self.param_groups[0]
        return (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                float(g["weight_decay"]), float(self.grad_scale), 0.0, 0.0)

    def sync_hyper(self) -> bool:
        """Upload the hyper-parameters if they changed since the last upload (one 32-byte copy; never while a
        stream capture is in progress: a captured copy would freeze the values into the graph)."""
        vals = self._hyper_values()
        if vals == self._uploaded:
            return False
        if torch.cuda.is_current_stream_capturing():
            raise L.HnError("ArenaAdam: hyper-parameters changed inside a stream capture; call sync_hyper() before")
        self.hyper.copy_(torch.tensor(vals, dtype=torch.float32))
        self._uploaded = vals
        return True

    @torch.no_grad()
    def step(self):
        L.load()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        a = self.arena
        L.launch("hn_adam_step", L.ptr(a.data), L.ptr(a.grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                 C.c_longlong(a.numel), L.ptr(self.hyper), L.ptr(self.step_count),
                 C.c_int(int(self.zero_grad_in_step)), L.stream_handle())
        a.bump()

    def zero_grad(self, set_to_none: bool = False):
        self.arena.zero_grad()

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count,
                "param_groups": self.param_groups}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"]); self.step_count.copy_(sd["step"])
        self.param_groups = sd["param_groups"]
        self.sync_hyper()


class MultiStepLR:
    """torch.optim.lr_scheduler.MultiStepLR for ArenaAdam (the reference's 'steplr', utils/__init__.py:43-46):
    lr = base_lr * gamma ** (number of milestones <= epoch).  `step()` is called once per epoch, as Lightning does
    with the scheduler `configure_optimizers` returns (train.py:128-131); it only writes `param_groups[0]['lr']`,
    the device copy follows at the next `sync_hyper()`."""

    def __init__(self, optimizer: ArenaAdam, milestones: Sequence[int], gamma: float = 0.1, last_epoch: int = 0):
        self.optimizer = optimizer
        self.milestones = sorted(int(m) for m in milestones)
        self.gamma = float(gamma)
        self.base_lr = float(optimizer.param_groups[0]["lr"])
        self.last_epoch = int(last_epoch)
        self._apply()

    def _apply(self):
        self.optimizer.param_groups[0]["lr"] = self.base_lr * self.gamma ** bisect_right(self.milestones, self.last_epoch)

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [self.optimizer.param_groups[0]["lr"]]

    def state_dict(self):
        return {"milestones": self.milestones, "gamma": self.gamma, "base_lr": self.base_lr,
                "last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.milestones, self.gamma = list(sd["milestones"]), float(sd["gamma"])
        self.base_lr, self.last_epoch = float(sd["base_lr"]), int(sd["last_epoch"])
        self._apply()
