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


# A/B switch of the fused reduce + Adam launch for the callers that turn it on (TrainStep, bench.py): HN_FUSE_REDUCE=0
FUSE_REDUCE = __import__("os").environ.get("HN_FUSE_REDUCE", "0") != "0"


class ArenaAdam:
    def __init__(self, arena: ParamArena, lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, zero_grad: bool = True, grad_scale: float = 1.0, fuse_reduce: bool = False):
        L.require_gpu(arena.data)
        self.arena = arena
        # fuse_reduce: consume the reduce launch of the batched weight gradient (machine.PendingReduce) — `step()` then
        # completes the gradient and applies the update in ONE launch (hn_mlp_wgrad_reduce_adam).  Between backward() and
        # step() the gradient buffer is then INCOMPLETE; `finish_gradients()` (or any ParamArena collective / zero_grad)
        # completes it with the plain reduce.  Single-GPU training steps: training.TrainStep and bench.py switch it on.
        self.fuse_reduce = bool(fuse_reduce)
        if self.fuse_reduce:
            import weakref
            from . import machine
            machine.REDUCE_CONSUMERS[machine._uid(arena.grad)] = weakref.ref(self)
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
        from . import machine
        pend = machine.take_pending_reduce(a.grad)
        if pend is not None:
            rest = pend.rest_table(a.grad) if self.fuse_reduce else None
            if rest is None:            # not a partition of this arena (or fusion off): the two-launch form
                pend.plain()
                pend = None
        if pend is not None:
            f = L.HnAdamFuse()
            f.params, f.grads, f.exp_avg, f.exp_avg_sq = a.data.data_ptr(), a.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
            f.n, f.hyper, f.step = a.numel, self.hyper.data_ptr(), self.step_count.data_ptr()
            f.rest, f.n_rest, f.zero_grad = rest[0].data_ptr(), rest[1], int(self.zero_grad_in_step)
            pend.fused(f)
        else:
            L.launch("hn_adam_step", L.ptr(a.data), L.ptr(a.grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                     C.c_longlong(a.numel), L.ptr(self.hyper), L.ptr(self.step_count),
                     C.c_int(int(self.zero_grad_in_step)), L.stream_handle())
        a.bump()

    def finish_gradients(self):
        """Complete the gradient buffer before step() (fuse_reduce holds the reduce launch back until then)."""
        from . import machine
        machine.flush_pending_reduce(self.arena.grad)

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


# ------------------------------------------------------------------------------------------------------------------
# the rest of the reference's `get_scheduler` (utils/__init__.py:43-59): 'cosine', 'poly', and the GradualWarmup wrapper
# (utils/warmup_scheduler.py) it puts around any of them when warmup_epochs > 0.  Pure host arithmetic on
# `optimizer.param_groups[0]['lr']` (any object with that attribute: ArenaAdam or a torch optimizer), stepped once per
# epoch; the device copy follows at the next `sync_hyper()`.  Pinned by tests/golden/g17_lr_schedules.npz, recorded
# from the reference's own get_scheduler (incl. what torch's chainable schedulers make of the warm-up hand-over).
# ------------------------------------------------------------------------------------------------------------------
class _EpochScheduler:
    def __init__(self, optimizer):
        self.optimizer = optimizer
        self.base_lr = float(optimizer.param_groups[0]["lr"])
        self.last_epoch = 0

    @property
    def lr(self) -> float:
        return float(self.optimizer.param_groups[0]["lr"])

    def _set(self, lr: float):
        self.optimizer.param_groups[0]["lr"] = float(lr)

    def get_lr(self) -> float:          # the chainable form: next lr from last_epoch and the CURRENT lr
        raise NotImplementedError

    def step(self):
        self.last_epoch += 1
        self._set(self.get_lr())

    def get_last_lr(self):
        return [self.lr]

    def state_dict(self):
        """Everything needed to resume mid-schedule: the scheduler's own fields, the learning rate the optimizer holds
        right now (the chainable forms compute the next rate FROM it) and — GradualWarmup — the wrapped scheduler's
        state (the reference's _LRScheduler.state_dict keeps `after_scheduler` inside its __dict__ the same way)."""
        sd = {k: v for k, v in self.__dict__.items() if k not in ("optimizer", "after")}
        sd["lr"] = self.lr
        after = getattr(self, "after", None)
        if after is not None:
            sd["after"] = after.state_dict()
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        lr, after = sd.pop("lr", None), sd.pop("after", None)
        self.__dict__.update(sd)
        if after is not None and getattr(self, "after", None) is not None:
            self.after.load_state_dict(after)
        if lr is not None:
            self._set(lr)           # the optimizer holds the restored rate, not the one it was constructed with


class StepLR(_EpochScheduler):
    """'steplr' in torch's chainable form (lr *= gamma at every milestone): what MultiStepLR above computes in closed
    form; this one can follow a warm-up that has changed the optimizer's lr."""

    def __init__(self, optimizer, milestones: Sequence[int], gamma: float = 0.1):
        super().__init__(optimizer)
        self.milestones = sorted(int(m) for m in milestones)
        self.gamma = float(gamma)

    def get_lr(self):
        return self.lr * self.gamma ** self.milestones.count(self.last_epoch)


class CosineAnnealingLR(_EpochScheduler):
    """'cosine' (utils/__init__.py:47-48: T_max = num_epochs, eta_min = 1e-8), torch's recursion."""

    def __init__(self, optimizer, T_max: int, eta_min: float = 1e-8):
        super().__init__(optimizer)
        self.T_max, self.eta_min = int(T_max), float(eta_min)

    def get_lr(self):
        import math
        e, T = self.last_epoch, self.T_max
        if (e - 1 - T) % (2 * T) == 0:
            return self.lr + (self.base_lr - self.eta_min) * (1 - math.cos(math.pi / T)) / 2
        return ((1 + math.cos(math.pi * e / T)) / (1 + math.cos(math.pi * (e - 1) / T)) * (self.lr - self.eta_min)
                + self.eta_min)


class PolyLR(_EpochScheduler):
    """'poly' as the reference states it (utils/__init__.py:49-52): lr = base_lr * (1 - epoch / num_epochs) ** poly_exp.
    Upstream raises NameError here (`LambdaLR` is never imported), so there is nothing to pin it to."""

    def __init__(self, optimizer, num_epochs: int, poly_exp: float = 0.9):
        super().__init__(optimizer)
        self.num_epochs, self.poly_exp = int(num_epochs), float(poly_exp)

    def get_lr(self):
        return self.base_lr * max(0.0, 1.0 - self.last_epoch / self.num_epochs) ** self.poly_exp


class GradualWarmup(_EpochScheduler):
    """GradualWarmupScheduler (utils/warmup_scheduler.py:4-66): lr ramps linearly from base_lr to base_lr * multiplier
    over `total_epoch` epochs, then `after` takes over with its base_lr scaled by the multiplier — including the
    hand-over epoch, where the reference asks the wrapped scheduler for a learning rate before ever stepping it."""

    def __init__(self, optimizer, multiplier: float, total_epoch: int, after: _EpochScheduler = None):
        if multiplier < 1.0:
            raise ValueError('multiplier should be greater thant or equal to 1.')
        super().__init__(optimizer)
        self.multiplier, self.total_epoch, self.after, self.finished = float(multiplier), int(total_epoch), after, False

    def get_lr(self):
        if self.last_epoch > self.total_epoch:
            if self.after is not None:
                if not self.finished:
                    self.after.base_lr = self.base_lr * self.multiplier
                    self.finished = True
                return self.after.get_lr()
            return self.base_lr * self.multiplier
        return self.base_lr * ((self.multiplier - 1.0) * self.last_epoch / self.total_epoch + 1.0)

    def step(self):
        if self.finished and self.after is not None:
            self.after.step()
        else:
            super().step()


def get_scheduler(hparams, optimizer):
    """The reference's get_scheduler (utils/__init__.py:43-59) for ArenaAdam: hparams needs `lr_scheduler` and, per
    kind, decay_step / decay_gamma | num_epochs | num_epochs / poly_exp, plus warmup_epochs / warmup_multiplier."""
    kind = hparams.lr_scheduler
    warm = getattr(hparams, "warmup_epochs", 0) > 0 and getattr(hparams, "optimizer", "adam") not in ("radam", "ranger")
    if kind == 'steplr':
        sch = (StepLR if warm else MultiStepLR)(optimizer, hparams.decay_step, hparams.decay_gamma)
    elif kind == 'cosine':
        sch = CosineAnnealingLR(optimizer, hparams.num_epochs, 1e-8)
    elif kind == 'poly':
        sch = PolyLR(optimizer, hparams.num_epochs, hparams.poly_exp)
    else:
        raise ValueError('scheduler not recognized!')
    if warm:
        sch = GradualWarmup(optimizer, hparams.warmup_multiplier, hparams.warmup_epochs, sch)
    return sch
