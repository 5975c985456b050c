"""MI355X drop-in for the reference's legacy `models/nerf.py` (nerf_pl): `Embedding`, `NeRF`.

Constructor signatures, attribute names and state_dict keys (`xyz_encoding_{i}.0.*`,
`xyz_encoding_final.*`, `dir_encoding.0.*`, `sigma.*`, `rgb.0.*`) follow models/nerf.py:4-124.
`NeRF.forward` runs the fused HIP MLP machine; `render_rays` additionally fuses both encoders into it.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from .. import _lib as L
from .. import functional as F
from ..machine import AuxSpec, Feature, GradIn, Layer, OutSpec, Program, copy_features


class Embedding(nn.Module):
    """x -> (x, sin(2^k x), cos(2^k x), ...) (reference: models/nerf.py:4-38)."""

    def __init__(self, in_channels, N_freqs, logscale=True):
        super().__init__()
        self.N_freqs = N_freqs
        self.in_channels = in_channels
        self.funcs = [torch.sin, torch.cos]
        self.out_channels = in_channels * (len(self.funcs) * N_freqs + 1)
        if logscale:
            self.freq_bands = 2 ** torch.linspace(0, N_freqs - 1, N_freqs)
        else:
            self.freq_bands = torch.linspace(1, 2 ** (N_freqs - 1), N_freqs)
        self._dev = {}

    def features(self, src: int, need_grad: bool = False) -> List[Feature]:
        out = [Feature(src, c, L.HN_FEAT_ID, 1.0, need_grad) for c in range(self.in_channels)]
        for f in self.freq_bands.tolist():
            out += [Feature(src, c, L.HN_FEAT_SIN, float(f), need_grad) for c in range(self.in_channels)]
            out += [Feature(src, c, L.HN_FEAT_COS, float(f), need_grad) for c in range(self.in_channels)]
        return out

    def forward(self, x):
        key = str(x.device)
        if key not in self._dev:
            self._dev[key] = self.freq_bands.to(x.device, torch.float32).contiguous()
        return F.posenc(x, self._dev[key], identity=True, jax_cos=False)


class NeRF(nn.Module):
    """The nerf_pl 8x256 NeRF MLP (reference: models/nerf.py:41-124)."""

    def __init__(self, D=8, W=256, in_channels_xyz=63, in_channels_dir=27, skips=[4]):
        super().__init__()
        self.D, self.W = D, W
        self.in_channels_xyz, self.in_channels_dir = in_channels_xyz, in_channels_dir
        self.skips = skips
        for i in range(D):
            if i == 0:
                layer = nn.Linear(in_channels_xyz, W)
            elif i in skips:
                layer = nn.Linear(W + in_channels_xyz, W)
            else:
                layer = nn.Linear(W, W)
            setattr(self, f"xyz_encoding_{i + 1}", nn.Sequential(layer, nn.ReLU(True)))
        self.xyz_encoding_final = nn.Linear(W, W)
        self.dir_encoding = nn.Sequential(nn.Linear(W + in_channels_dir, W // 2), nn.ReLU(True))
        self.sigma = nn.Linear(W, 1)
        self.rgb = nn.Sequential(nn.Linear(W // 2, 3), nn.Sigmoid())
        self._calls = {}

    def layers(self, xyz_aux: AuxSpec, dir_aux: Optional[AuxSpec], sigma_only: bool) -> List[Layer]:
        """dst 0 = (P,4) [rgb | sigma] (or (P,1) sigma when sigma_only); backward source 4 = its gradient,
        5 = the output itself (sigmoid')."""
        out: List[Layer] = []
        for i in range(self.D):
            lin = getattr(self, f"xyz_encoding_{i + 1}")[0]
            if i == 0:
                out.append(Layer(f"xyz_encoding_{i + 1}", lin.weight, lin.bias, aux=xyz_aux, aux_c0=0, act="relu"))
            elif i in self.skips:   # cat([input_xyz, xyz_]) BEFORE layer i (models/nerf.py:107-110)
                out.append(Layer(f"xyz_encoding_{i + 1}", lin.weight, lin.bias, main=(self.in_channels_xyz, self.W),
                                 aux=xyz_aux, aux_c0=0, act="relu"))
            else:
                out.append(Layer(f"xyz_encoding_{i + 1}", lin.weight, lin.bias, main=(0, self.W), act="relu"))
        scol = 0 if sigma_only else 3
        out.append(Layer("sigma", self.sigma.weight, self.sigma.bias, main=(0, self.W), act="none", commit=False,
                         out=OutSpec(0, scol, "none"), grad_in=GradIn(4, scol)))
        if sigma_only:
            return out
        out.append(Layer("xyz_encoding_final", self.xyz_encoding_final.weight, self.xyz_encoding_final.bias,
                         main=(0, self.W), act="none"))
        de = self.dir_encoding[0]
        out.append(Layer("dir_encoding", de.weight, de.bias, main=(0, self.W), aux=dir_aux, aux_c0=self.W, act="relu"))
        rg = self.rgb[0]
        out.append(Layer("rgb", rg.weight, rg.bias, main=(0, self.W // 2), act="none", commit=False,
                         out=OutSpec(0, 0, "sigmoid"), grad_in=GradIn(4, 0, (5, 0))))
        return out

    def _embedded_call(self, sigma_only: bool) -> F.ProgramCall:
        call = self._calls.get(("emb", sigma_only))
        if call is None:
            xyz_aux = AuxSpec(copy_features(0, range(self.in_channels_xyz)))
            dir_aux = None if sigma_only else AuxSpec(
                copy_features(0, range(self.in_channels_xyz, self.in_channels_xyz + self.in_channels_dir)))
            call = F.ProgramCall(Program(self.layers(xyz_aux, dir_aux, sigma_only), name="NeRF"), [False],
                                 [1 if sigma_only else 4], [("g", 0), ("y", 0)])
            self._calls[("emb", sigma_only)] = call
        return call

    def fused_call(self, emb_xyz: Embedding, emb_dir: Embedding, sigma_only: bool) -> F.ProgramCall:
        """Program with both encoders generated in-kernel: sources 0 = xyz (P,3), 1 = ray directions (B,3)."""
        key = ("fused", sigma_only, emb_xyz.N_freqs, emb_dir.N_freqs, tuple(emb_xyz.freq_bands.tolist()))
        call = self._calls.get(key)
        if call is None:
            if emb_xyz.out_channels != self.in_channels_xyz or emb_dir.out_channels != self.in_channels_dir:
                raise ValueError("embedding widths do not match the NeRF input channels")
            xyz_aux = AuxSpec(emb_xyz.features(0))
            dir_aux = None if sigma_only else AuxSpec(emb_dir.features(1))
            call = F.ProgramCall(Program(self.layers(xyz_aux, dir_aux, sigma_only), name="NeRF_fused"),
                                 [False, True], [1 if sigma_only else 4], [("g", 0), ("y", 0)])
            self._calls[key] = call
        return call

    def forward(self, x, sigma_only=False):
        """x: (B, in_channels_xyz(+in_channels_dir)) embedded inputs -> (B,4) rgb+sigma, or (B,1) sigma."""
        if x.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("gradients w.r.t. the embedded inputs of NeRF are not implemented "
                                      "(the reference never needs them)")
        (y,) = F.run_program(self._embedded_call(bool(sigma_only)), [x], 1)
        return y
