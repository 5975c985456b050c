"""MI355X drop-in for the reference's `hypernerf/modules.py`: MLP, GLOEmbed, NerfMLP, HyperSheetMLP.

Same constructor signatures, attribute names, sub-module names (=> identical `state_dict` keys,
nn.Linear (out,in) fp32 layout) and initialisers in the same RNG order as the reference
(hypernerf/modules.py:46-337).  `forward` does not run ATen layers: every module compiles itself
into a program for the MFMA "MLP machine" (hypernerf_torch_amd.machine) and launches the HIP
kernels through the C ABI.  There is no CPU path.
"""
from __future__ import annotations

import functools
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from .. import _lib as L
from .. import functional as F
from ..machine import AuxSpec, Feature, GradIn, Layer, OutSpec, Program, copy_features
from . import model_utils


def _act_name(act, what: str) -> str:
    if act is None or isinstance(act, nn.Identity):
        return "none"
    if isinstance(act, nn.ReLU):
        return "relu"
    if isinstance(act, nn.Sigmoid):
        return "sigmoid"
    raise NotImplementedError(f"{what}: activation {type(act).__name__} is not implemented in the HIP MLP machine "
                              "(reference path uses ReLU / Sigmoid / Identity only)")


def mlp_layers(mlp: "MLP", prefix: str, input_aux: Optional[AuxSpec], main_in: Optional[int],
               out: Optional[OutSpec], grad_in: Optional[GradIn]) -> List[Layer]:
    """Layer list of one reference-style MLP (hypernerf/modules.py:116-127).

    The MLP input is [running activation (main_in features, may be None) | generated features
    (input_aux, may be None)]; skip layers re-append the same input after `linears[i]`, i in skips.
    """
    if _act_name(mlp.hidden_activation, prefix) != "relu":
        raise NotImplementedError(f"{prefix}: hidden activation must be ReLU")
    out_act = _act_name(mlp.output_activation, prefix)
    if main_in is not None and any(i in mlp.skips for i in range(len(mlp.linears) - 1)):
        # a skip would have to re-append the running activation, which no longer exists
        raise NotImplementedError(f"{prefix}: skip connections need generated (aux) inputs only")
    layers: List[Layer] = []
    n_hidden = len(mlp.linears)
    if (n_hidden - 1) in mlp.skips:
        # the reference concatenates the input after EVERY layer in `skips` (modules.py:122-124), also the last hidden
        # one, and then dies in logit_layer (in_features = width, modules.py:104) with a shape RuntimeError
        raise RuntimeError(f"{prefix}: skip after the last hidden layer — logit_layer expects {mlp.width} input "
                           f"features, the concatenation has more (the reference fails the same way: mat1 and mat2 "
                           f"shapes cannot be multiplied)")
    for i, lin in enumerate(mlp.linears):
        if i == 0:
            ly = Layer(f"{prefix}.linears.0", lin.weight, lin.bias,
                       main=(0, main_in) if main_in is not None else None,
                       aux=input_aux, aux_c0=main_in or 0, act="relu")
        elif (i - 1) in mlp.skips:
            ly = Layer(f"{prefix}.linears.{i}", lin.weight, lin.bias, main=(0, mlp.width), aux=input_aux,
                       aux_c0=mlp.width, act="relu")
        else:
            ly = Layer(f"{prefix}.linears.{i}", lin.weight, lin.bias, main=(0, mlp.width), act="relu")
        layers.append(ly)
    lg = mlp.logit_layer
    if out is not None and not out.wide:
        if out_act == "relu":
            raise NotImplementedError(f"{prefix}: ReLU on a narrow output head")
        out = OutSpec(out.dst, out.col, out_act, out.residual, False, out.publish)
        layers.append(Layer(f"{prefix}.logit_layer", lg.weight, lg.bias, main=(0, mlp.width), act="none",
                            commit=False, out=out, grad_in=grad_in))
    else:
        if out_act == "sigmoid":
            raise NotImplementedError(f"{prefix}: sigmoid on a wide output")
        layers.append(Layer(f"{prefix}.logit_layer", lg.weight, lg.bias, main=(0, mlp.width), act=out_act,
                            commit=True, out=out, grad_in=grad_in))
    return layers


class MLP(nn.Module):
    """A multi-layer perceptron (reference: hypernerf/modules.py:46-127)."""

    def __init__(self, in_ch: int, out_ch: int, depth: int = 8, width: int = 256, hidden_init=None,
                 hidden_activation=None, hidden_norm=None, output_init=None, output_channels=0,
                 output_activation=None, use_bias=True, skips=None):
        super().__init__()
        self.in_ch, self.out_ch, self.depth, self.width = in_ch, out_ch, depth, width
        self.hidden_init = nn.init.xavier_uniform_ if hidden_init is None else hidden_init
        self.hidden_activation = nn.ReLU() if hidden_activation is None else hidden_activation
        self.hidden_norm = hidden_norm
        self.output_init = nn.init.xavier_uniform_ if output_init is None else output_init
        self.output_channels = output_channels
        self.output_activation = nn.Identity() if output_activation is None else output_activation
        self.use_bias = use_bias
        self.skips = [4, ] if skips is None else skips
        # linears[0]: in->width; linears[i+1]: width(+in if i in skips)->width, i < depth-1  (:99-101)
        self.linears = nn.ModuleList(
            [nn.Linear(in_ch, width)] +
            [nn.Linear(width + in_ch, width) if i in self.skips else nn.Linear(width, width)
             for i in range(depth - 1)])
        self.logit_layer = nn.Linear(width, out_ch)
        for lin in self.linears:
            self.hidden_init(lin.weight)
        if self.output_init is not None:
            self.output_init(self.logit_layer.weight)
        self._calls = {}
        self._check_machine_limits()

    def _check_machine_limits(self):
        """What the HIP MLP machine does not run is refused HERE, at construction — before a mis-configured model reaches
        the GPU (round 6; the first forward raised before).  The reference's own configurations all pass (ReLU hidden
        layers, Identity / Sigmoid heads, width <= 256: modules.py:62-114, models.py:139-166).  INTEGRATION.md lists these."""
        name = type(self).__name__
        if _act_name(self.hidden_activation, name) != "relu":
            raise NotImplementedError(f"{name}: hidden_activation {type(self.hidden_activation).__name__}: the hidden layers "
                                      "of the HIP MLP machine are ReLU (mask bit + v_max on the bit pattern)")
        out_act = _act_name(self.output_activation, name)           # raises for anything but Identity / ReLU / Sigmoid
        if self.width > 256:
            raise NotImplementedError(f"{name}: width {self.width} > 256 — a wave carries one hidden vector of <= 256 "
                                      "features (64 VGPRs in bf16) through the layers in registers; not a host-side limit")
        if self.width < 1 or self.depth < 0 or self.in_ch < 1 or self.out_ch < 1:
            raise ValueError(f"{name}: in_ch, out_ch, width >= 1 and depth >= 0")
        if self.out_ch > 256:
            raise NotImplementedError(f"{name}: out_ch {self.out_ch} > 256 (one layer = <= 8 tiles of 32 rows)")
        if self.out_ch > 4 and out_act == "sigmoid":
            raise NotImplementedError(f"{name}: Sigmoid on an output of {self.out_ch} > 4 columns (the sigmoid lives in the "
                                      "narrow-head epilogue: rgb / alpha heads)")
        if self.in_ch > 64 * L.HN_AUXG_MAX:
            raise NotImplementedError(f"{name}: in_ch {self.in_ch} > {64 * L.HN_AUXG_MAX} input features of one layer")

    def _call(self, need_input_grad: bool) -> F.ProgramCall:
        call = self._calls.get(need_input_grad)
        if call is None:
            aux = AuxSpec(copy_features(0, range(self.in_ch), need_input_grad))
            out_act = _act_name(self.output_activation, "mlp")
            narrow = self.out_ch <= 4 and out_act != "relu"
            grad_in = GradIn(4, 0, (5, 0) if (narrow and out_act == "sigmoid") else None)
            layers = mlp_layers(self, "mlp", aux, None, OutSpec(0, 0, "none", None, wide=not narrow), grad_in)
            call = F.ProgramCall(Program(layers, name="MLP"), [False], [self.out_ch], [("g", 0), ("y", 0)])
            self._calls[need_input_grad] = call
        return call

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        lead = inputs.shape[:-1]
        call = self._call(bool(inputs.requires_grad and torch.is_grad_enabled()))
        (y,) = F.run_program(call, [inputs.reshape(-1, self.in_ch)], 1)
        return y.view(*lead, self.out_ch)


class GLOEmbed(nn.Module):
    """GLO embedding table (reference: hypernerf/modules.py:131-167)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, embedding_init=None):
        super().__init__()
        self.num_embeddings = num_embeddings
        self.embedding_dim = embedding_dim
        if embedding_init is None:
            embedding_init = functools.partial(nn.init.normal_, std=0.1 / embedding_dim)
        self.embedding_init = embedding_init
        self.embed = nn.Embedding(num_embeddings=num_embeddings, embedding_dim=embedding_dim)
        self.embedding_init(self.embed.weight)

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        if inputs.shape[-1] == 1:
            inputs = torch.squeeze(inputs, dim=-1)
        out = F.embed_lookup(self.embed.weight, inputs)
        res = out.view(*inputs.shape, self.embedding_dim)
        if res.dim() == 2:
            res._hn_embed = out._hn_embed      # the view carries the tag (functional._scatter_embed_grad)
        return res


def nerf_mlp_layers(m: "NerfMLP", prefix: str, input_aux: AuxSpec, alpha_aux: Optional[AuxSpec],
                    rgb_aux: Optional[AuxSpec], dst_rgb: int = 0, dst_alpha: int = 1) -> List[Layer]:
    """Layer list of NerfMLP.forward (hypernerf/modules.py:266-298): trunk -> bottleneck ->
    {alpha head, rgb MLP}.  dst `dst_rgb` = rgb (P,3), dst `dst_alpha` = alpha (P,1); backward sources 4 = d rgb,
    5 = d alpha, 6 = rgb (for sigmoid')."""
    layers = mlp_layers(m.trunk_mlp, f"{prefix}.trunk_mlp", input_aux, None, None, None)
    bw = m.bottleneck_mlp.weight.shape[0]
    layers.append(Layer(f"{prefix}.bottleneck_mlp", m.bottleneck_mlp.weight, m.bottleneck_mlp.bias,
                        main=(0, m.trunk_width), act="none"))
    layers.append(Layer(f"{prefix}.alpha_mlp", m.alpha_mlp.weight, m.alpha_mlp.bias, main=(0, bw), aux=alpha_aux,
                        aux_c0=bw, act="none", commit=False, out=OutSpec(dst_alpha, 0, "none"), grad_in=GradIn(5, 0)))
    rgb_act = _act_name(m.rgb_mlp.output_activation, prefix + ".rgb_mlp")
    layers += mlp_layers(m.rgb_mlp, f"{prefix}.rgb_mlp", rgb_aux, bw, OutSpec(dst_rgb, 0, rgb_act),
                         GradIn(4, 0, (6, 0) if rgb_act == "sigmoid" else None))
    return layers


class NerfMLP(nn.Module):
    """Template NeRF MLP (reference: hypernerf/modules.py:172-298)."""

    def __init__(self, in_ch, trunk_depth=8, trunk_width=256, rgb_branch_depth=1, rgb_branch_width=128,
                 rgb_channels=3, alpha_brach_depth=1, alpha_brach_width=128, alpha_channels=1, skips=None,
                 hidden_activation=None, rgb_activation=None, alpha_condition_dim: int = 8,
                 rgb_condition_dim: int = 39, norm=None):
        super().__init__()
        self.in_ch = in_ch
        self.trunk_depth, self.trunk_width = trunk_depth, trunk_width
        self.rgb_branch_depth, self.rgb_branch_width = rgb_branch_depth, rgb_branch_width
        self.rgb_channels = rgb_channels
        self.alpha_branch_depth, self.alpha_branch_width = alpha_brach_depth, alpha_brach_width
        self.alpha_channels = alpha_channels
        self.alpha_condition_dim, self.rgb_condition_dim = alpha_condition_dim, rgb_condition_dim
        self.condition_density = False
        # the reference overwrites these with the passed value even when None (modules.py:207-217);
        # MLP re-defaults None, so the effective behaviour is: None -> [4] / ReLU / Identity.
        self.skips = skips
        self.hidden_activation = hidden_activation
        self.rgb_activation = rgb_activation
        self.sigma_activation = nn.Identity()
        self.norm = norm
        self.trunk_mlp = MLP(in_ch=in_ch, out_ch=trunk_width, depth=trunk_depth, width=trunk_width,
                             hidden_activation=hidden_activation, skips=skips,
                             output_activation=hidden_activation)
        self.bottleneck_mlp = nn.Linear(trunk_width, trunk_width // 2)
        self.rgb_mlp = MLP(in_ch=rgb_branch_width + rgb_condition_dim, out_ch=rgb_channels,
                           depth=rgb_branch_depth, hidden_activation=hidden_activation,
                           output_activation=rgb_activation, width=rgb_branch_width, skips=skips)
        self.alpha_mlp = nn.Linear(alpha_brach_width + alpha_condition_dim, alpha_channels)
        nn.init.xavier_uniform_(self.alpha_mlp.weight)
        self._calls = {}
        # refused at construction (round 6), not at the first forward: the rgb head is a narrow (<= 4 column) head of the
        # machine — Identity or Sigmoid (models.py:164, 288) —, the trunk feeds the bottleneck at trunk_width // 2
        if _act_name(self.rgb_mlp.output_activation, "NerfMLP.rgb_mlp") == "relu":
            raise NotImplementedError("NerfMLP: rgb_activation ReLU (a ReLU on a <= 4-column head is not implemented in "
                                      "the HIP MLP machine; the reference uses Sigmoid / Identity)")
        if rgb_channels > 4 or alpha_channels > 4:
            raise NotImplementedError("NerfMLP: rgb_channels / alpha_channels > 4 (narrow heads of the HIP MLP machine)")

    def broadcast_condition(self, c, num_samples):
        if c.dim() == 2:
            c = c.unsqueeze(1)
        return c.repeat(1, num_samples, 1)

    def _call(self, x_grad: bool, ac: Optional[int], ac_grad: bool, rc: Optional[int], rc_grad: bool):
        key = (x_grad, ac, ac_grad, rc, rc_grad)
        call = self._calls.get(key)
        if call is None:
            x_aux = AuxSpec(copy_features(0, range(self.in_ch), x_grad))
            a_aux = AuxSpec(copy_features(1, range(ac), ac_grad)) if ac else None
            r_aux = AuxSpec(copy_features(2, range(rc), rc_grad)) if rc else None
            layers = nerf_mlp_layers(self, "nerf", x_aux, a_aux, r_aux)
            call = F.ProgramCall(Program(layers, name="NerfMLP"), [False, True, True],
                                 [self.rgb_channels, self.alpha_channels], [("g", 0), ("g", 1), ("y", 0)])
            self._calls[key] = call
        return call

    def forward(self, x, alpha_condition=None, rgb_condition=None):
        """x: (B,S,in_ch); conditions (B,C) broadcast over S.  Returns {'rgb': (B,S,3), 'alpha': (B,S,1)}."""
        b, s = x.shape[0], x.shape[1]
        ge = torch.is_grad_enabled()

        def prep(c):
            if c is None:
                return None
            if c.dim() == 3:
                if c.shape[1] != 1:
                    raise NotImplementedError("per-sample conditions: pass (B,C) or (B,1,C) as the reference does")
                c = c[:, 0]
            return c

        ac, rc = prep(alpha_condition), prep(rgb_condition)
        call = self._call(bool(x.requires_grad and ge), None if ac is None else ac.shape[-1],
                          bool(ac is not None and ac.requires_grad and ge), None if rc is None else rc.shape[-1],
                          bool(rc is not None and rc.requires_grad and ge))
        rgb, alpha = F.run_program(call, [x.reshape(b * s, -1), ac, rc], s)
        return {"rgb": rgb.view(b, s, -1), "alpha": alpha.view(b, s, -1)}


class HyperSheetMLP(nn.Module):
    """Ambient-dimension slicing MLP (reference: hypernerf/modules.py:302-337)."""

    def __init__(self, in_ch: int = 3, in_ch_embed: int = 8, out_ch: int = 3, depth: int = 6, width: int = 64,
                 min_deg: int = 0, max_deg: int = 1, skips=None, use_residual=False):
        super().__init__()
        self.out_ch, self.depth, self.width = out_ch, depth, width
        self.min_deg, self.max_deg = min_deg, max_deg
        self.in_ch_embed = in_ch_embed
        self.n_freq = 7  # hard-coded in the reference (modules.py:313)
        self.in_ch_pts = in_ch
        self.in_ch = model_utils.get_posenc_ch_orig(in_ch, self.n_freq) + in_ch_embed
        self.skips = [4, ] if skips is None else skips
        self.hidden_init = nn.init.xavier_uniform_
        self.output_init = functools.partial(nn.init.normal_, std=1e-5)
        self.use_residual = use_residual
        self.mlp = MLP(in_ch=self.in_ch, out_ch=self.out_ch, depth=self.depth, hidden_init=self.hidden_init,
                       output_init=self.output_init, width=self.width, skips=self.skips)
        self._calls = {}

    def input_aux(self, pts_src: int, embed_src: int, pts_grad: bool, embed_grad: bool) -> AuxSpec:
        from ..machine import posenc_features
        return AuxSpec(posenc_features(pts_src, range(self.in_ch_pts), self.n_freq, pts_grad) +
                       copy_features(embed_src, range(self.in_ch_embed), embed_grad))

    def _call(self, per_ray_embed: bool, pts_grad: bool, embed_grad: bool):
        key = (per_ray_embed, pts_grad, embed_grad)
        call = self._calls.get(key)
        if call is None:
            wide = self.out_ch > 4
            layers = mlp_layers(self.mlp, "mlp", self.input_aux(0, 1, pts_grad, embed_grad), None,
                                OutSpec(0, 0, "none", None, wide=wide), GradIn(4, 0))
            call = F.ProgramCall(Program(layers, name="HyperSheetMLP"), [False, per_ray_embed], [self.out_ch],
                                 [("g", 0)])
            self._calls[key] = call
        return call

    def forward(self, pts, embed, alpha=None):
        """pts (B,S,3); embed (B,S,E) broadcast per sample (the reference's call shape) or (B,E)."""
        if self.use_residual:
            raise NotImplementedError("use_residual=True is unreachable in the reference model and not implemented")
        lead = pts.shape[:-1]
        ge = torch.is_grad_enabled()
        per_ray = embed.dim() == pts.dim() - 1
        s = pts.shape[-2] if pts.dim() >= 3 else 1
        call = self._call(per_ray, bool(pts.requires_grad and ge), bool(embed.requires_grad and ge))
        e = embed if per_ray else embed.reshape(-1, embed.shape[-1])
        (y,) = F.run_program(call, [pts.reshape(-1, pts.shape[-1]), e], s if per_ray else 1)
        return y.view(*lead, self.out_ch)
