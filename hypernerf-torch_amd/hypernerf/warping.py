"""MI355X drop-in for the reference's `hypernerf/warping.py`: TranslationField, SE3Field.

Constructor signatures, attribute / sub-module names (state_dict keys) and initialisers follow the
reference (hypernerf/warping.py:28-272); `forward` runs the fused HIP MLP machine.
"""
from __future__ import annotations

import functools
from functools import partial
from typing import Any, Dict, Iterable, Optional

import torch
import torch.nn as nn

from .. import functional as F
from ..machine import AuxSpec, GradIn, Layer, OutSpec, Program, copy_features, posenc_features, posenc_jax_features
from . import model_utils, modules


class TranslationField(nn.Module):
    """Warp field predicting a per-point translation (reference: hypernerf/warping.py:28-125)."""

    def __init__(self, in_ch, min_deg=0, max_deg=8, in_ch_embed: int = 8, use_posenc_identity=True,
                 skips: list = None, depth=6, hidden_channels=128, activation=None, norm=None, hidden_init=None,
                 output_init=None):
        super().__init__()
        self.min_deg, self.max_deg = min_deg, max_deg
        self.use_posenc_identity = use_posenc_identity
        self.skips = [4, ] if skips is None else skips
        self.depth = depth
        self.embed_dim = in_ch_embed
        self.hidden_channels = hidden_channels
        self.activation = nn.ReLU() if activation is None else activation
        self.norm = norm
        self.hidden_init = nn.init.xavier_normal_ if hidden_init is None else hidden_init
        self.output_init = functools.partial(nn.init.uniform_, b=1e-4) if output_init is None else output_init
        self.n_freq = 10  # hard-coded in the reference (warping.py:74); min_deg/max_deg are inert
        self.in_ch_pts = in_ch
        self.in_ch = model_utils.get_posenc_ch_orig(in_ch, self.n_freq) + in_ch_embed
        self.out_ch = 3
        self.mlp = modules.MLP(in_ch=self.in_ch, out_ch=self.out_ch, depth=self.depth, width=self.hidden_channels,
                               hidden_activation=self.activation, hidden_norm=self.norm,
                               hidden_init=self.hidden_init, output_init=self.output_init, skips=self.skips)
        self._calls = {}

    def input_aux(self, pts_src: int, embed_src: int, pts_grad: bool, embed_grad: bool) -> AuxSpec:
        return AuxSpec(posenc_features(pts_src, range(self.in_ch_pts), self.n_freq, pts_grad) +
                       copy_features(embed_src, range(self.embed_dim), embed_grad))

    def _call(self, per_ray_embed: bool, pts_grad: bool, embed_grad: bool) -> F.ProgramCall:
        key = (per_ray_embed, pts_grad, embed_grad)
        call = self._calls.get(key)
        if call is None:
            # pts_grad: d(p + delta(p)) / dp = the output gradient itself (the residual, added to the source's gradient
            # by functional._ProgramFn.backward) + the encoder's share through the MLP (reference: autograd through
            # warping.py:90-96; the render path never asks for it — sample points carry no gradient)
            layers = modules.mlp_layers(self.mlp, "mlp", self.input_aux(0, 1, pts_grad, embed_grad), None,
                                        OutSpec(0, 0, "none", residual=(0, 0)), GradIn(4, 0))
            call = F.ProgramCall(Program(layers, name="TranslationField"), [False, per_ray_embed], [3], [("g", 0)])
            self._calls[key] = call
        return call

    def warp(self, points: torch.Tensor, metadata: torch.Tensor, extra_params) -> torch.Tensor:
        """points (..., 3); metadata = encoded embedding, broadcast per point (..., E) as the reference passes
        it, or per ray (B, E) for points (B, S, 3)."""
        lead = points.shape[:-1]
        ge = torch.is_grad_enabled()
        per_ray = metadata.dim() == points.dim() - 1
        s = points.shape[-2] if per_ray else 1
        call = self._call(per_ray, bool(points.requires_grad and ge), bool(metadata.requires_grad and ge))
        m = metadata if per_ray else metadata.reshape(-1, metadata.shape[-1])
        (y,) = F.run_program(call, [points.reshape(-1, 3), m], s)
        return y.view(*lead, 3)

    def forward(self, points: torch.Tensor, metadata: torch.Tensor, extra_params, return_jacobian: bool = False):
        out = {'warped_points': self.warp(points, metadata, extra_params)}
        if return_jacobian:
            raise NotImplementedError
        return out


class SE3Field(nn.Module):
    """SE(3) warp field (reference: hypernerf/warping.py:128-272) — BASELINE config 5.

    The reference class is never instantiated by its model (`models.py:234` hard-codes TranslationField) and its
    own `warp` only runs for a single point and then returns ones because of layout bugs in rigid_body
    (SURVEY.md §8a-19), so there is no reference output to pin: this implements what the code states it computes
    — posenc (model_utils.py:255-274, quirks included) -> trunk MLP -> w_net / v_net -> exp_se3 -> R p + t, per
    point, the metadata embedding ignored as upstream does (warping.py:223-224) — and is checked against the
    oracle's restatement of the same formulas ("parity unpinned").  Use it by assigning
    `model.warp_field = SE3Field(in_ch=3)`.

    Kernels: encoder, trunk and both heads are ONE program of the HIP MLP machine (the two heads' first layers as one
    row-stacked 128 -> 256 layer, their logit layers as windows on its halves); the exponential map and the rigid
    transform are `hn_se3_apply_*`.  No library GEMM is involved.
    """

    def __init__(self, in_ch=1, out_ch=1):
        super().__init__()
        self.out_ch = out_ch
        self.min_deg: int = 0
        self.max_deg: int = 8
        self.use_posenc_identity: bool = False
        self.activation = torch.nn.ReLU()
        self.norm: Optional[Any] = None
        self.skips: Iterable[int] = (4,)
        self.trunk_depth: int = 6
        self.trunk_width: int = 128
        self.rotation_depth: int = 0
        self.rotation_width: int = 128
        self.pivot_depth: int = 0
        self.pivot_width: int = 128
        self.translation_depth: int = 0
        self.translation_width: int = 128
        self.default_init = nn.init.xavier_normal_
        self.rotation_init = partial(nn.init.uniform_, b=1e-4)
        self.translation_init = partial(nn.init.uniform_, b=1e-4)
        self.in_ch_pts = in_ch
        self.in_ch = model_utils.get_posenc_ch(in_ch, min_deg=self.min_deg, max_deg=self.max_deg,
                                               use_identity=self.use_posenc_identity, alpha=None)
        self.trunk = modules.MLP(in_ch=self.in_ch, out_ch=self.trunk_width, depth=self.trunk_depth,
                                 width=self.trunk_width, hidden_activation=self.activation, hidden_norm=self.norm,
                                 hidden_init=self.default_init, skips=self.skips)
        self.w_net = modules.MLP(in_ch=self.trunk_width, out_ch=3, depth=self.rotation_depth,
                                 width=self.rotation_width, hidden_activation=self.activation,
                                 hidden_norm=self.norm, hidden_init=self.default_init,
                                 output_init=self.rotation_init, output_channels=3)
        self.v_net = modules.MLP(in_ch=self.trunk_width, out_ch=3, depth=self.translation_depth,
                                 width=self.translation_width, hidden_activation=self.activation,
                                 hidden_norm=self.norm, hidden_init=self.default_init,
                                 output_init=self.translation_init, output_channels=3)
        self._calls = {}

    def _field_call(self, pts_grad: bool) -> F.ProgramCall:
        """ONE program for posenc -> trunk -> both heads (warping.py:212-225): the first Linear of `w_net` and of
        `v_net` read the same trunk output, so they run as one 128 -> 256 layer whose matrix is the two stacked by
        rows; each 128 -> 3 logit layer then reads its own half of that activation (a window).  Output (P, 6) =
        [w | v]."""
        call = self._calls.get(pts_grad)
        if call is None:
            if len(self.w_net.linears) != 1 or len(self.v_net.linears) != 1:
                raise NotImplementedError("SE3Field heads deeper than the reference's (depth 0) are not implemented")
            aux = AuxSpec(posenc_jax_features(0, range(self.in_ch_pts), self.min_deg, self.max_deg,
                                              self.use_posenc_identity, pts_grad))
            layers = modules.mlp_layers(self.trunk, "trunk", aux, None, None, None)
            tw, hw = self.trunk_width, self.rotation_width
            if self.translation_width != hw or hw % 32:
                raise NotImplementedError("SE3Field: rotation and translation heads of different / odd widths")
            w0, v0 = self.w_net.linears[0], self.v_net.linears[0]
            layers.append(Layer("heads.linears.0", [w0.weight, v0.weight], [w0.bias, v0.bias], main=(0, tw), act="relu"))
            wl, vl = self.w_net.logit_layer, self.v_net.logit_layer
            layers.append(Layer("w_net.logit_layer", wl.weight, wl.bias, main=(0, 2 * hw), act="none", commit=False,
                                out=OutSpec(0, 0, "none"), grad_in=GradIn(4, 0)))
            layers.append(Layer("v_net.logit_layer", vl.weight, vl.bias, main=(-hw, 2 * hw), act="none", commit=False,
                                out=OutSpec(0, 3, "none"), grad_in=GradIn(4, 3)))
            call = F.ProgramCall(Program(layers, name="SE3Field"), [False], [6], [("g", 0)])
            self._calls[pts_grad] = call
        return call

    def warp(self, points: torch.Tensor, metadata_embed: torch.Tensor, extra_params: Dict[str, Any]):
        """points (..., 3) -> warped points (..., 3); `metadata_embed` is ignored (reference: warping.py:223-224).
        Differentiable w.r.t. the points as well (through the encoder + trunk and through R p + t)."""
        lead = points.shape[:-1]
        ge = torch.is_grad_enabled()
        flat = points.reshape(-1, self.in_ch_pts)
        pg = bool(points.requires_grad and ge)
        (wv,) = F.run_program(self._field_call(pg), [flat], 1)
        return F.se3_warp(wv, flat if pg else flat.detach()).view(*lead, 3)

    def warp_with_rows(self, points: torch.Tensor, table: torch.Tensor, idx: torch.Tensor):
        """points (B, S, 3) -> (xyz (B, S, 3), warped (B, S, 3 + H)) with warped = [xyz | table[idx[ray]]]: the
        `warped_points` of an axis-aligned-plane level (models.py:533-534, 578-581) written by the exp-map launch
        itself.  Both outputs are differentiable (a loss on `warped` reaches the field through its xyz columns and the
        table through its row columns, as the reference's cat([xyz, rows]) does)."""
        b, s = points.shape[0], points.shape[1]
        flat = points.reshape(-1, self.in_ch_pts)
        pg = bool(points.requires_grad and torch.is_grad_enabled())
        (wv,) = F.run_program(self._field_call(pg), [flat], 1)
        xyz, warped = F.se3_warp(wv, flat if pg else flat.detach(), table, idx, s)
        return xyz.view(b, s, 3), warped.view(b, s, -1)

    def forward(self, points, metadata, extra_params, return_jacobian: bool = False):
        out = {'warped_points': self.warp(points, metadata, extra_params)}
        if return_jacobian:
            raise NotImplementedError
        return out
