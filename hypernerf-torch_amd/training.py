"""The training step of the reference's Lightning system as one replayable GPU program
(reference: train.py:96-114 `forward`, 146-163 `training_step`; optimizer from utils.get_optimizer).

`TrainStep(model, lr=...)` owns what NeRFSystem owns around the model — loss, optimizer — laid out the MI355X way:
parameters and gradients in a `ParamArena`, `ArenaAdam` (one kernel for all parameters), the whole step (prepare_ray_dict -> model ->
MSE -> backward -> fused HIP Adam) captured once into a HIP graph and replayed on fixed input buffers; with
torch.distributed initialised, rays are expected pre-sharded per rank and the gradient buffer is all-reduced in
place between the captured forward+backward and the optimizer step.  `step(rays, rgbs)` returns the same log the
reference's training_step records: {'train/loss', 'train/psnr', 'lr'} (device scalars, no host sync).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist

from .arena import ParamArena
from .optim import ArenaAdam
from .graphs import GraphedStep
from .hypernerf import model_utils
from .losses import MSELoss, psnr

_EXTRA = {'nerf_alpha': None, 'warp_alpha': None, 'hyper_alpha': None, 'hyper_sheet_alpha': None}


class TrainStep:
    def __init__(self, model: torch.nn.Module, lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, use_graph: bool = True, group=None):
        self.model = model
        self.arena = ParamArena(model.parameters())
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.group = group
        self.use_graph = use_graph
        # hn_adam_step: one launch for all parameters, clears the gradient buffer on the way out, graph-capturable
        # (on one GPU it is part of the captured step; with N>1 it follows the gradient all-reduce)
        self.optimizer = ArenaAdam(self.arena, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, zero_grad=True)
        self.loss_fn = MSELoss()
        self._graph: Optional[GraphedStep] = None
        self._rays = self._rgbs = None
        self._log: Dict[str, torch.Tensor] = {}

    # ---- the step body (what gets captured) ---------------------------------------------------
    def _forward_backward(self):
        results = self.model(model_utils.prepare_ray_dict(self._rays), dict(_EXTRA))
        loss = self.loss_fn(results, self._rgbs)
        typ = 'fine' if 'fine' in results else 'coarse'
        loss.backward()          # into arena.grad, left zeroed by the previous optimizer step
        with torch.no_grad():
            self._log = {'train/loss': loss.detach(), 'train/psnr': psnr(results[typ]['rgb'].detach(), self._rgbs)}

    def _whole(self):
        self._forward_backward()
        self.optimizer.step()

    # ---- public -----------------------------------------------------------------------------------
    def step(self, rays: torch.Tensor, rgbs: torch.Tensor) -> Dict[str, torch.Tensor]:
        """rays (B, 8|9), rgbs (B, 3) on the GPU; B must stay the same from call to call when graphs are on."""
        if self._rays is None or self._rays.shape != rays.shape:
            self._rays, self._rgbs = rays.clone(), rgbs.clone()
            self._graph = None
        else:
            self._rays.copy_(rays)
            self._rgbs.copy_(rgbs)
        if not self.use_graph:
            self._forward_backward()
            if self.world > 1:
                self.arena.all_reduce_mean(self.group)
            self.optimizer.step()
        elif self.world == 1:
            if self._graph is None:     # first call: 2 eager warm-up steps (real steps), the capture, then the replay
                self._graph = GraphedStep(self._whole, warmup=2)
            self._graph()
        else:
            if self._graph is None:
                self._graph = GraphedStep(self._forward_backward, warmup=2)
            self._graph()
            self.arena.all_reduce_mean(self.group)
            self.optimizer.step()
        log = {k: v.clone() for k, v in self._log.items()}     # graph outputs are overwritten by the next replay
        log['lr'] = self.optimizer.param_groups[0]['lr']
        return log
