"""Fused Adam on a ParamArena: one HIP kernel updates every parameter of the model and clears the
gradient buffer for the next step (`hn_adam_step`).  Same update rule and defaults as torch.optim.Adam, which is what
the reference's `get_optimizer` builds (utils/__init__.py).  The step counter is a device scalar, so `step()` can be
captured in a HIP graph."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .arena import ParamArena


class ArenaAdam:
    def __init__(self, arena: ParamArena, lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, zero_grad: bool = True):
        L.require_gpu(arena.data)
        self.arena = arena
        self.param_groups = [{"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay}]
        self.zero_grad_in_step = zero_grad
        self.exp_avg = torch.zeros_like(arena.data)
        self.exp_avg_sq = torch.zeros_like(arena.data)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=arena.data.device)

    @torch.no_grad()
    def step(self):
        L.load()
        g = self.param_groups[0]
        a = self.arena
        L.launch("hn_adam_step", L.ptr(a.data), L.ptr(a.grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                 C.c_longlong(a.numel), C.c_float(g["lr"]), C.c_float(g["betas"][0]), C.c_float(g["betas"][1]),
                 C.c_float(g["eps"]), C.c_float(g["weight_decay"]), L.ptr(self.step_count),
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
