"""The training step of the reference's Lightning system as one replayable GPU program
(reference: train.py:96-114 `forward` with its chunk loop, 146-163 `training_step`, 116-131 `configure_optimizers`;
optimizer / scheduler from utils.get_optimizer / get_scheduler, utils/__init__.py:23-59).

`TrainStep(model, lr=...)` owns what NeRFSystem owns around the model — loss, optimizer, LR schedule — laid out the
MI355X way: parameters and gradients in a `ParamArena`, `ArenaAdam` (one kernel for all parameters, hyper-parameters
and step counter on the device), the whole step (prepare_ray_dict -> model per chunk -> MSE -> backward -> Adam)
captured once into a HIP graph and replayed on fixed input buffers.  With torch.distributed initialised the initial
parameters are broadcast from rank 0 (what Lightning's DDP wrapper does), rays are expected pre-sharded per rank,
the gradient buffer is SUM-all-reduced in place between the captured forward+backward and the optimizer step —
optionally in two buckets, the first in flight while the weight gradients of the second are still being computed
(dist.GradSync, `overlap_grad_sync=True`) — and the 1/world of the mean is folded into the Adam kernel.
`step(rays, rgbs)` returns the log the reference's training_step records: {'train/loss', 'train/psnr', 'lr'}
(device scalars, no host sync).
"""
from __future__ import annotations

import contextlib
from typing import Dict, Optional, Sequence

import torch
import torch.distributed as dist

from . import functional as F
from . import machine
from .arena import ParamArena
from .dist import GradSync, collective_capturable
from .graphs import GraphedStep
from .hypernerf import model_utils
from .losses import MSELoss, psnr
from .optim import ArenaAdam, MultiStepLR, get_scheduler

_EXTRA = {'nerf_alpha': None, 'warp_alpha': None, 'hyper_alpha': None, 'hyper_sheet_alpha': None}


class TrainStep:
    def __init__(self, model: torch.nn.Module, lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, use_graph: bool = True, group=None, chunk: int = 32 * 1024,
                 decay_step: Optional[Sequence[int]] = None, decay_gamma: float = 0.1, overlap_grad_sync: bool = False,
                 hparams=None, force_dp: bool = False, capture_collective: bool = True):
        self.model = model
        self.arena = ParamArena(model.parameters())
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.group = group
        self.use_graph = use_graph
        self.chunk = int(chunk)          # rays per model call (train.py:108-111); the default exceeds any batch size
        self.sync: Optional[GradSync] = None
        # force_dp: take the data-parallel code path (broadcast, gradient all-reduce, Adam behind it) in a ONE-rank
        # group too — a one-GPU box can then run the captured RCCL all-reduce (tests, bench.py --force-dp)
        self.dp = self.world > 1 or (bool(force_dp) and dist.is_available() and dist.is_initialized())
        self.capture_collective = bool(capture_collective)
        self.dp_graph = None             # how the N>1 step runs: "one graph" | "three pieces (<why>)"
        if self.dp:
            # replicas must start identical (Lightning DDP broadcasts module state from rank 0 at wrap time)
            dist.broadcast(self.arena.data, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            self.arena.bump()
            # one all-reduce of the flat gradient buffer, or (overlap_grad_sync=True) two buckets with the first in
            # flight while the second is still being computed — off by default, see dist.GradSync
            self.sync = GradSync(self.arena, model, group, overlap=overlap_grad_sync)
        # hn_adam_step: one launch for all parameters, clears the gradient buffer on the way out, graph-capturable
        # (on one GPU it is part of the captured step; with N>1 it follows the gradient all-reduce and scales the
        # summed gradient by 1/world itself)
        # Without a collective between backward and the optimizer (one GPU) the launch that completes the gradient also
        # applies the update (hn_mlp_wgrad_reduce_adam: no gradient write-back, no second pass over the arena)
        from . import optim as _optim
        self.optimizer = ArenaAdam(self.arena, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, zero_grad=True,
                                   grad_scale=1.0 / self.world, fuse_reduce=(not self.dp) and _optim.FUSE_REDUCE)
        # 'steplr' of the reference (utils/__init__.py:43-46): stepped once per epoch by the caller (`epoch_end`)
        self.scheduler = MultiStepLR(self.optimizer, decay_step, decay_gamma) if decay_step else None
        # any scheduler of the reference's get_scheduler (utils/__init__.py:43-59): `hparams` carries lr_scheduler
        # ('steplr' | 'cosine' | 'poly') and its arguments, warmup_epochs / warmup_multiplier for the warm-up wrapper
        if hparams is not None and getattr(hparams, "lr_scheduler", None):
            self.scheduler = get_scheduler(hparams, self.optimizer)
        self.loss_fn = MSELoss()
        self._graph: Optional[GraphedStep] = None
        self._rays = self._rgbs = None
        self._rng: Optional[Dict[str, torch.Tensor]] = None    # fixed buffers of injected random draws (tests)
        self._log: Dict[str, torch.Tensor] = {}

    # ---- the step body (what gets captured) ---------------------------------------------------
    def _forward_backward(self):
        with (self.sync.splitting() if self.sync is not None else contextlib.nullcontext()):
            self._forward_backward_body()

    def _forward_backward_body(self):
        b = self._rays.shape[0]
        loss_sum, psnr_in = None, []
        # the reference renders `chunk` rays at a time and concatenates the results before the loss
        # (train.py:108-114); mean((rgb-gt)^2) over the batch = sum over chunks of (rays in chunk / B) x chunk mean,
        # so every chunk is back-propagated on its own (its activations are freed before the next chunk runs)
        for i in range(0, b, self.chunk):
            rays, rgbs = self._rays[i:i + self.chunk], self._rgbs[i:i + self.chunk]
            kw = {} if self._rng is None else {'rng': {k: v[i:i + self.chunk] for k, v in self._rng.items()}}
            results = self.model(model_utils.prepare_ray_dict(rays), dict(_EXTRA), **kw)
            w = rays.shape[0] / b
            loss = self.loss_fn(results, rgbs)
            F.backward(loss, w)          # into arena.grad (zeroed by the previous Adam launch); cached root gradient = w
            if i + self.chunk < b:
                F.flush_held_wgrads()        # only the LAST chunk's held jobs overlap with the all-reduce (a held
                                             # job keeps its chunk's activation stash alive)
            typ = 'fine' if 'fine' in results else 'coarse'
            with torch.no_grad():
                loss_sum = loss.detach() * w if loss_sum is None else loss_sum + loss.detach() * w
                psnr_in.append(results[typ]['rgb'].detach())
        with torch.no_grad():
            pred = psnr_in[0] if len(psnr_in) == 1 else torch.cat(psnr_in, 0)
            self._log = {'train/loss': loss_sum, 'train/psnr': psnr(pred, self._rgbs)}

    def _whole(self):
        self._forward_backward()
        self.optimizer.step()

    def _whole_dp(self):
        """forward + backward | ONE in-place SUM all-reduce of the gradient buffer | Adam, as one capturable program
        (RCCL's kernel is enqueued on the capturing stream like any launch): one graph replay per step, no host in
        the loop between backward and the optimizer (reference: Lightning DDP's reduce inside backward,
        train.py:224-229)."""
        self._forward_backward()
        F.flush_held_wgrads()
        self.arena.all_reduce_sum(self.group, force=True)
        self.optimizer.step()

    def _snapshot(self):
        o = self.optimizer
        return [t.clone() for t in (self.arena.data, self.arena.grad, o.exp_avg, o.exp_avg_sq, o.step_count)]

    def _restore(self, snap):
        o = self.optimizer
        with torch.no_grad():
            for dst, src in zip((self.arena.data, self.arena.grad, o.exp_avg, o.exp_avg_sq, o.step_count), snap):
                dst.copy_(src)
        self.arena.bump()

    def _capture(self, fn):
        """Warm-up runs + capture execute `fn` for real; parameters, gradient buffer and optimizer state are put back
        afterwards so that the first step() applies exactly ONE update (and, with N>1, all-reduces ONE gradient)."""
        snap = self._snapshot()
        try:
            g = GraphedStep(fn, warmup=2)
        finally:
            self._restore(snap)
        return g

    def _capture_data_parallel(self):
        """Two graphs: forward + backward up to the first weight-gradient bucket | the held bucket (None when the
        model offers no split).  The warm-up runs execute both; the capture of the first leaves exactly one pass's
        held jobs behind, which the capture of the second consumes."""
        snap = self._snapshot()
        F.flush_held_wgrads()
        g1 = GraphedStep(self._forward_backward, warmup=2, mutates_params=False,
                         warmup_fn=lambda: (self._forward_backward(), F.flush_held_wgrads()))
        g2 = (GraphedStep(F.flush_held_wgrads, warmup=0, pool=g1.graph.pool(), mutates_params=False)
              if F.held_wgrads() else None)
        self._restore(snap)
        return g1, g2

    def _capture_dp(self):
        """The N>1 step as ONE graph when the collective can be captured (RCCL, single all-reduce); otherwise — gloo,
        the two-bucket overlap, or a capture that raised — graph | eager all-reduce | eager Adam, in this process
        (never a re-exec)."""
        why = None
        if self.sync.split is not None:
            why = "two overlapped buckets"
        elif not self.capture_collective:
            why = "capture_collective=False"
        elif not collective_capturable(self.group):
            why = f"backend {dist.get_backend(self.group)} stages through the host"
        else:
            g = None
            try:
                g = self._capture(self._whole_dp)
            except Exception as e:      # noqa: BLE001 (whatever the runtime says about capturing the collective)
                why = f"capturing the all-reduce raised {type(e).__name__}: {str(e)[:120]}"
                # a capture that died part-way may have queued weight-gradient shares whose stashes belong to the
                # abandoned graph's pool: they must never be launched
                F.drop_pending_wgrads()
            # every rank takes the SAME form of the step (a rank replaying one graph next to a rank issuing an eager
            # all-reduce would still match collective for collective, but time and behave differently): agree on it
            ok = torch.tensor([1.0 if g is not None else 0.0], device=self.arena.data.device)
            if self.world > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if float(ok.item()) > 0.5:
                self.dp_graph = "one graph: forward + backward + all-reduce + Adam"
                return g
            if g is not None:
                why = "another rank could not capture the all-reduce"
                g = None
        self.dp_graph = f"three pieces: graph | eager all-reduce | Adam ({why})"
        return self._capture_data_parallel()

    # ---- public -----------------------------------------------------------------------------------
    def step(self, rays: torch.Tensor, rgbs: torch.Tensor, rng: Optional[Dict[str, torch.Tensor]] = None
             ) -> Dict[str, torch.Tensor]:
        """rays (B, 8|9), rgbs (B, 3) on the GPU; B must stay the same from call to call when graphs are on.
        `rng` optionally supplies the random draws of NerfModel.forward ('t_rand', 'u', 'noise_coarse', 'noise_fine',
        one row per ray) instead of torch's generator: parity runs against the CPU oracle share the draws this way."""
        if self._rays is None or self._rays.shape != rays.shape or (rng is None) != (self._rng is None):
            self._rays, self._rgbs = rays.clone(), rgbs.clone()
            self._rng = None if rng is None else {k: v.clone() for k, v in rng.items()}
            self._graph = None
        else:
            self._rays.copy_(rays)
            self._rgbs.copy_(rgbs)
            if rng is not None:
                for k, v in rng.items():
                    self._rng[k].copy_(v)
        # what a captured graph froze besides the shapes: the precision mode and whether the model trains / evaluates
        state = (F.get_precision(), self.model.training)
        if getattr(self, "_graph_state", None) != state:
            self._graph_state = state
            self._graph = None
        self.optimizer.sync_hyper()      # a replayed Adam launch reads lr & co. from device memory
        if not self.use_graph:
            self._forward_backward()
            if self.dp:
                self.sync.reduce(F.flush_held_wgrads, force=True)
            self.optimizer.step()
        elif not self.dp:
            if self._graph is None:
                self._graph = self._capture(self._whole)
            self._graph()
        else:
            if self._graph is None:
                self._graph = self._capture_dp()
            if isinstance(self._graph, GraphedStep):
                self._graph()               # forward + backward + all-reduce + Adam: ONE replay
            else:
                fwd_bwd, held = self._graph
                fwd_bwd()
                self.sync.reduce(held, force=True)   # all-reduce(bucket 0) || held weight gradients, then all-reduce(bucket 1)
                self.optimizer.step()
        # a replay updates the parameters without running any Python: tell the weight packers (an eval forward
        # after this must repack — see machine.MlpRunner.pack)
        self.arena.bump()
        machine.note_parameters_changed()
        log = {k: v.clone() for k, v in self._log.items()}     # graph outputs are overwritten by the next replay
        log['lr'] = self.optimizer.param_groups[0]['lr']
        return log

    def epoch_end(self):
        """Advance the LR schedule by one epoch (Lightning steps the scheduler of configure_optimizers per epoch)."""
        if self.scheduler is not None:
            self.scheduler.step()
