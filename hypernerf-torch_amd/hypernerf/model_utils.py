"""MI355X drop-in for the reference's `hypernerf/model_utils.py` (function names, arguments and
return values as in the reference; file:line cited per function).  Device work goes through the
C-ABI HIP kernels; the ray-dict helpers are host-side dict plumbing."""
from __future__ import annotations

from typing import Optional

import torch

from .. import _lib as L
from .. import functional as F
from ..machine import AuxSpec, GradIn, Layer, OutSpec, Program, posenc_features, posenc_jax_features

_BOUNDS_CACHE = {}


def stratified_bounds(n: int, near: float, far: float, lindisp: bool, device):
    """(lower, upper, z_lin) of the n stratified bins, computed once on the host with the same ATen
    CPU ops as the reference (model_utils.py:25-33) so that z is bit-identical, then cached on `device`."""
    key = (n, float(near), float(far), bool(lindisp), str(device))
    if key not in _BOUNDS_CACHE:
        t = torch.linspace(0., 1., n)
        if not lindisp:
            z = near * (1. - t) + far * t
        else:
            z = 1. / (1. / near * (1. - t) + 1. / far * t)
        mids = .5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], dim=-1)
        lower = torch.cat([z[..., :1], mids], dim=-1)
        _BOUNDS_CACHE[key] = tuple(x.to(device) for x in (lower, upper, z))
    return _BOUNDS_CACHE[key]


def sample_along_rays(origins, directions, num_coarse_samples, near, far, use_stratified_sampling,
                      use_linear_disparity, t_rand: Optional[torch.Tensor] = None):
    """Stratified sampling along rays (reference: hypernerf/model_utils.py:6-41).
    Returns (z_vals (B,N), points (B,N,3)).  `t_rand` overrides the torch.rand draw (tests)."""
    L.require_gpu(origins, directions)
    lower, upper, z_lin = stratified_bounds(num_coarse_samples, near, far, use_linear_disparity, origins.device)
    b = origins.shape[0]
    if use_stratified_sampling:
        if t_rand is None:
            t_rand = torch.rand([b, num_coarse_samples], device=origins.device)
        return F.sample_along_rays(origins, directions, lower, upper, t_rand)
    return F.sample_along_rays(origins, directions, z_lin, None, None)


def volumetric_rendering(rgb, sigma, z_vals, dirs, use_white_background, sample_at_infinity=True, eps=1e-5):
    """Alpha compositing of ACTIVATED densities (reference: hypernerf/model_utils.py:43-107).
    The fused model path feeds raw densities + noise to the kernel; this entry point exists for API
    parity and inverts nothing: it runs the legacy-variant kernel arithmetic on sigma directly."""
    if eps != 1e-5:
        raise NotImplementedError("eps is fixed to 1e-5 in the HIP kernel (the reference never changes it)")
    outs = F.composite(rgb, sigma, None, z_vals, dirs, None, variant=2, white_bg=use_white_background,
                       sample_at_infinity=sample_at_infinity, want_median=True)
    return {"rgb": outs[0], "depth": outs[1], "med_depth": outs[4], "acc": outs[2], "weights": outs[3]}


def _draws(b, n, stratified, device, u):
    if u is not None:
        return u
    if stratified:
        return torch.rand(b, n, device=device)
    return torch.linspace(0, 1, n).to(device).expand(b, n).contiguous()


def piecewise_constant_pdf(bins, weights, num_coarse_samples, use_stratified_sampling, u=None):
    """Inverse-CDF samples (reference: hypernerf/model_utils.py:160-204).  bins (B,n+1), weights (B,n)."""
    L.require_gpu(bins, weights)
    u = _draws(weights.shape[0], num_coarse_samples, use_stratified_sampling, bins.device, u)
    _, _, _, zs = F.sample_pdf(weights, None, u, bins=bins, merge=False)
    return zs


def sample_pdf(bins, weights, origins, directions, z_vals, num_coarse_samples, use_stratified_sampling, u=None):
    """Hierarchical sampling: new samples merged into z_vals, sorted, + points
    (reference: hypernerf/model_utils.py:206-232)."""
    L.require_gpu(bins, weights, z_vals)
    u = _draws(weights.shape[0], num_coarse_samples, use_stratified_sampling, bins.device, u)
    z_all, pts, _, _ = F.sample_pdf(weights, z_vals, u, origins, directions, bins=bins)
    return z_all, pts


_BANDS = {}


def _bands(key, device):
    """Frequency bands of the two encoders, computed ONCE per (arguments, device) with the reference's ATen ops on the
    CPU (the values the CPU reference — and the golden fixtures — use, bit for bit) and kept on the device: no
    pageable host-to-device copy (a sync point, and an error inside HIP-graph capture) per call."""
    k = (key, str(device))
    t = _BANDS.get(k)
    if t is None:
        if key[0] == "orig":
            n = key[1]
            t = 2 ** torch.linspace(0, n - 1, n) if key[2] else torch.linspace(0, n - 1, n)
        else:
            t = 2. ** torch.linspace(key[1], key[2], steps=key[2] - key[1])
        t = _BANDS[k] = t.to(device)
    return t


def posenc_orig(x, N_freqs, log_scale=True):
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(N-1) x), cos(2^(N-1) x)] in blocks of C channels — the
    "SinusoidalEncoder" (reference: hypernerf/model_utils.py:234-246).  `log_scale=False` takes the reference's
    linspace(0, N-1, N) bands (frequency 0 included).  One HIP launch (hn_posenc), differentiable w.r.t. x."""
    L.require_gpu(x)
    n = int(N_freqs)
    return F.posenc(x, _bands(("orig", n, bool(log_scale)), x.device), identity=True, jax_cos=False)


def posenc(x, min_deg, max_deg, use_identity=False, alpha=None):
    """JAX-style encoder of the SE3 field (reference: hypernerf/model_utils.py:255-274), quirks kept: scales =
    2**linspace(min_deg, max_deg, steps=max_deg-min_deg) (non-integer exponents), cos as sin(x + 0.5*3.1415926),
    layout (*, F, 2, C) flattened; `alpha` windowing is disabled upstream (:264-266) and ignored here too."""
    L.require_gpu(x)
    return F.posenc(x, _bands(("jax", int(min_deg), int(max_deg)), x.device), identity=bool(use_identity), jax_cos=True)


def _depth_index(weights, z_vals, depth_threshold, want_index, want_depth, want_mask):
    L.require_gpu(weights)
    L.load()
    import ctypes as C
    shp = weights.shape
    s = shp[-1]
    w = weights.detach().reshape(-1, s).contiguous().float()
    b = w.shape[0]
    z = z_vals.detach().expand(shp).reshape(-1, s).contiguous().float() if z_vals is not None else None
    idx = torch.empty(b, dtype=torch.int64, device=w.device) if want_index else None
    dep = torch.empty(b, dtype=torch.float32, device=w.device) if want_depth else None
    msk = torch.empty(b, s, dtype=torch.float32, device=w.device) if want_mask else None
    L.launch("hn_depth_index", L.ptr(w), L.ptr(z), C.c_int(b), C.c_int(s), C.c_float(float(depth_threshold)),
             L.ptr(idx), L.ptr(dep), L.ptr(msk), L.stream_handle())
    return (idx.view(shp[:-1]) if idx is not None else None, dep.view(shp[:-1]) if dep is not None else None,
            msk.view(shp).to(weights.dtype) if msk is not None else None)


def compute_opaqueness_mask(weights, depth_threshold=0.5):
    """1.0 at the first sample whose accumulated weight reaches the threshold (reference: model_utils.py:319-340)."""
    return _depth_index(weights, None, depth_threshold, False, False, True)[2]


def compute_depth_index(weights, depth_threshold=0.5):
    """Sample index of the median depth accumulation (reference: model_utils.py:342-345)."""
    return _depth_index(weights, None, depth_threshold, True, False, False)[0]


def compute_depth_map(weights, z_vals, depth_threshold=0.5):
    """Depth by median accumulation = z at compute_depth_index, 0 where the threshold is never reached
    (reference: model_utils.py:347-362).  The reference's sum(mask * z_vals) is differentiable w.r.t. z_vals (never
    w.r.t. the weights: the mask is a comparison); when z_vals asks for a gradient the same product is taken in torch
    on the kernel's mask, otherwise the kernel returns the depth itself."""
    if z_vals is not None and z_vals.requires_grad and torch.is_grad_enabled():
        mask = _depth_index(weights, None, depth_threshold, False, False, True)[2]
        return torch.sum(mask * z_vals, dim=-1)
    return _depth_index(weights, z_vals, depth_threshold, False, True, False)[1]


def get_posenc_ch_orig(in_ch, N_freq, log_scale=True):
    """Channel count of posenc_orig (reference: hypernerf/model_utils.py:248-252), no device work."""
    return in_ch * (1 + 2 * N_freq)


def get_posenc_ch(in_ch, min_deg, max_deg, use_identity=False, alpha=None):
    """Channel count of posenc (reference: hypernerf/model_utils.py:276-280)."""
    return in_ch * (2 * (max_deg - min_deg) + (1 if use_identity else 0))


def noise_regularize(raw, noise_std, use_stratified_sampling, noise=None):
    """reference: hypernerf/model_utils.py:300-317.  (The fused path adds the noise inside the compositing
    kernel; this helper keeps the reference's dict-in/dict-out contract.)"""
    if (noise_std is not None) and noise_std > 0.0 and use_stratified_sampling:
        if noise is None:
            noise = torch.randn(raw['alpha'].shape, device=raw['alpha'].device, dtype=raw['alpha'].dtype) * noise_std
        raw["alpha"] = raw["alpha"] + noise
    return raw


class RayMetadata(dict):
    """`metadata` of prepare_ray_dict: the four keys of the reference ('warp', 'camera', 'appearance', 'time') all hold
    ONE int64 tensor, column 8 of the ray rows converted with `.type(torch.long)` (reference model_utils.py:389-398).
    The conversion is made on first access — NerfModel.forward's fused step head (hn_render_prologue) makes it inside
    its own launch instead (`raw` = the fp32 column view, `set_converted`) —, so a dict is all a caller ever sees."""
    KEYS = ('warp', 'camera', 'appearance', 'time')

    def __init__(self, column=None, **kw):
        if isinstance(column, torch.Tensor):
            super().__init__({k: None for k in self.KEYS})
            self.raw = column
            self._idx = None
        else:       # rebuilt from (key, value) pairs by code that maps over containers (DDP's input scatter, copy): a plain dict
            super().__init__(column if column is not None else (), **kw)
            self.raw = None
            self._idx = next((v for v in super().values() if v is not None), None)

    def converted(self) -> bool:
        return self._idx is not None or self.raw is None

    def set_converted(self, idx: torch.Tensor):
        self._idx = idx
        for k in self.KEYS:
            super().__setitem__(k, idx)

    def _get(self):
        if self._idx is None and self.raw is not None:
            self.set_converted(self.raw.type(torch.long))
        return self._idx

    def __getitem__(self, k):
        if k in self.KEYS and super().__getitem__(k) is None:
            self._get()
        return super().__getitem__(k)

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        self._get()
        return super().items()

    def values(self):
        self._get()
        return super().values()

    def copy(self):
        self._get()
        return dict(super().items())


def prepare_ray_dict(rays: torch.Tensor) -> dict:
    """(B,8|9) nerf_pl ray rows -> ray dict (reference: hypernerf/model_utils.py:365-404).
    near/far columns are read and dropped, viewdirs is None, column 8 feeds all four metadata keys."""
    use_meta = rays.shape[-1] == 9
    if len(rays.shape) > 2:
        rays = rays.view(-1, 8)
    b = rays.shape[0]
    if use_meta and rays.dtype == torch.float32 and rays.is_cuda:
        metadata = RayMetadata(rays[:, 8])          # converted on first use (or by the model's step head, in its launch)
    else:
        if use_meta:
            idx = rays[:, 8].type(torch.long)
        else:
            idx = torch.ones((b, 1), dtype=torch.long, device=rays.device)
        metadata = {k: idx for k in ('warp', 'camera', 'appearance', 'time')}     # read-only downstream: one tensor
    return {"origins": rays[:, :3], "directions": rays[:, 3:6], "viewdirs": None, "metadata": metadata}


def extract_rays_batch(rays: dict, start: int, end: int, drop_last=True) -> dict:
    """reference: hypernerf/model_utils.py:407-430."""
    out = {}
    for key, val in rays.items():
        if key == 'metadata':
            out[key] = {k: (v[start:end] if v is not None else None) for k, v in val.items()}
        else:
            out[key] = val[start:end] if val is not None else None
    return out


def append_batch(all_ret, batch) -> dict:
    """reference: hypernerf/model_utils.py:432-442."""
    for k, v in all_ret.items():
        if v is None:
            all_ret[k] = batch[k]
        else:
            for kk, vv in batch[k].items():
                if vv is not None:
                    all_ret[k][kk] = torch.cat([all_ret[k][kk], vv], dim=0)
    return all_ret


def concat_ray_batch(rays: list) -> dict:
    """reference: hypernerf/model_utils.py:444-461."""
    result = {k: None for k in rays[0].keys()}
    for c in rays:
        for k, v in c.items():
            result[k] = v if result[k] is None else torch.cat([result[k], v], dim=0)
    return result
