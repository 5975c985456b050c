"""CPU oracle for the HyperNeRF render hot path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.
The shipped path (package `hypernerf_torch_amd`) never imports anything under
`oracle/`; it calls the HIP kernels through the C-ABI and fails loudly without them.

What it is: a functional, plain-PyTorch (CPU, fp32) restatement of the algorithm of
songrise/HyperNeRF-torch for the path named in BASELINE.json — written from the
reference's behaviour, not copied from it.  Every function cites the reference
file:line it follows (paths relative to the reference checkout).  All randomness is
an explicit input tensor (`t_rand`, `noise_*`, `u`) so CPU and GPU runs share draws.

Parity pin: the reference ships no tests; `tests/golden/make_golden.py` imports the
reference itself in the build container (CPU, with three import stubs) and records
inputs/outputs as fixtures under `tests/golden/`.  `tests/test_oracle_golden.py`
checks this oracle against every one of them.  Two things are NOT pinned by the
reference: `torchsearchsorted` (absent third-party CUDA extension —
defined here as `torch.searchsorted(right=True)`, which is what the live path uses,
hypernerf/model_utils.py:190) and `SE3Field.warp` (broken upstream, never
instantiated) — parity unpinned for those two, beyond the single `exp_se3` matrix.

One deliberate, documented deviation (SURVEY.md §7 "bit-exact fine-sample indices"):
the pdf normaliser is an fp64-accumulated sum rounded once to fp32 instead of ATen's
ISA-dependent vectorised fp32 cascade sum; it differs from the reference by <= 1 ulp
and makes the oracle reproducible on any host and bit-equal to the HIP kernel.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# encoders
# --------------------------------------------------------------------------------------
def posenc_orig(x: Tensor, n_freqs: int) -> Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(N-1) x), cos(2^(N-1) x)], blocks of C.

    hypernerf/model_utils.py:234-246 (log_scale=True is the only reachable branch);
    same layout as the legacy models/nerf.py:4-38 `Embedding(logscale=True)`.
    """
    parts = [x]
    for k in range(n_freqs):
        f = float(2.0 ** k)
        parts.append(torch.sin(f * x))
        parts.append(torch.cos(f * x))
    return torch.cat(parts, dim=-1)


def posenc_ch(in_ch: int, n_freqs: int) -> int:
    """hypernerf/model_utils.py:248-252 (channel count only)."""
    return in_ch * (1 + 2 * n_freqs)


def posenc_jax(x: Tensor, min_deg: int, max_deg: int, use_identity: bool = False) -> Tensor:
    """JAX-style encoder with the reference's quirks, hypernerf/model_utils.py:255-274.

    scales = 2**linspace(min,max,steps=max-min) (non-integer exponents, :258) and cosine
    taken as sin(x + 0.5*3.1415926) (:262).  Output order: (freq, {sin,cos}, channel).
    """
    steps = max_deg - min_deg
    scales = 2.0 ** torch.linspace(float(min_deg), float(max_deg), steps=steps, dtype=x.dtype)
    xb = x[..., None, :] * scales[:, None]
    feat = torch.sin(torch.stack((xb, xb + 0.5 * 3.1415926), dim=-2))
    feat = feat.reshape(*x.shape[:-1], -1)
    return torch.cat([x, feat], dim=-1) if use_identity else feat


# --------------------------------------------------------------------------------------
# MLPs (hypernerf/modules.py)
# --------------------------------------------------------------------------------------
_BF16_OPERANDS = False


class bf16_operands:
    """Context manager for the tests of the bf16 product mode: inside it every Linear of the oracle rounds its
    matmul OPERANDS to bf16 (round-to-nearest-even) and accumulates in fp32 — forward (x, W) and backward
    (grad_out for both dX = g W and dW = g^T x, and for db = sum g).  Bias add, activations, encoders, compositing
    and the loss stay fp32.  That is the arithmetic contract of the MFMA path in bf16 mode, so a
    bf16 HIP result can be held against it far more tightly than against the fp32 oracle.  Not part of the
    reference's algorithm; the golden fixtures are never produced or checked under it."""

    def __enter__(self):
        global _BF16_OPERANDS
        self._prev = _BF16_OPERANDS
        _BF16_OPERANDS = True
        return self

    def __exit__(self, *exc):
        global _BF16_OPERANDS
        _BF16_OPERANDS = self._prev
        return False


def _r16(t: Tensor) -> Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


class _LinearBf16Operands(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        xr, wr = _r16(x), _r16(w)
        ctx.save_for_backward(xr, wr)
        return F.linear(xr, wr, b)

    @staticmethod
    def backward(ctx, g):
        xr, wr = ctx.saved_tensors
        gr = _r16(g)
        g2, x2 = gr.reshape(-1, gr.shape[-1]), xr.reshape(-1, xr.shape[-1])
        return gr @ wr, g2.t() @ x2, g2.sum(0)


def _linear(p: Params, prefix: str, x: Tensor) -> Tensor:
    if _BF16_OPERANDS:
        return _LinearBf16Operands.apply(x, p[prefix + ".weight"], p[prefix + ".bias"])
    return F.linear(x, p[prefix + ".weight"], p[prefix + ".bias"])


def mlp(p: Params, prefix: str, x: Tensor, depth: int, skips: Sequence[int] = (4,),
        out_act: str = "none") -> Tensor:
    """modules.MLP.forward, hypernerf/modules.py:116-127.

    `linears[i]` + ReLU for i < max(depth,1); after layer i in `skips` the running
    activation becomes cat([x, inputs]); then `logit_layer` + output activation.
    (depth=0 still owns one hidden layer, modules.py:99-101.)
    """
    inputs = x
    n_hidden = max(depth, 1)
    for i in range(n_hidden):
        x = torch.relu(_linear(p, f"{prefix}.linears.{i}", x))
        if i in skips:
            x = torch.cat([x, inputs], dim=-1)
    x = _linear(p, f"{prefix}.logit_layer", x)
    if out_act == "relu":
        x = torch.relu(x)
    elif out_act == "sigmoid":
        x = torch.sigmoid(x)
    elif out_act != "none":
        raise ValueError(out_act)
    return x


def glo_embed(table: Tensor, idx: Tensor) -> Tensor:
    """modules.GLOEmbed.forward, hypernerf/modules.py:155-167 (squeeze trailing 1)."""
    if idx.shape[-1] == 1:
        idx = idx.squeeze(-1)
    return table[idx]


def translation_field(p: Params, prefix: str, pts: Tensor, embed: Tensor) -> Tensor:
    """TranslationField.warp, hypernerf/warping.py:90-96; n_freq hard-coded 10 (:74)."""
    h = torch.cat([posenc_orig(pts, 10), embed], dim=-1)
    return pts + mlp(p, f"{prefix}.mlp", h, depth=6)


def hyper_sheet(p: Params, prefix: str, pts: Tensor, embed: Tensor) -> Tensor:
    """HyperSheetMLP.forward, hypernerf/modules.py:331-337; n_freq hard-coded 7 (:313)."""
    h = torch.cat([posenc_orig(pts, 7), embed], dim=-1)
    return mlp(p, f"{prefix}.mlp", h, depth=6)


def nerf_mlp(p: Params, prefix: str, x: Tensor, alpha_cond: Optional[Tensor],
             rgb_cond: Optional[Tensor], trunk_depth: int = 8, rgb_depth: int = 4,
             skips: Sequence[int] = (4,)):
    """NerfMLP.forward, hypernerf/modules.py:266-298.

    trunk MLP (logit 256->256 + ReLU) -> bottleneck Linear (no activation) ->
    alpha = Linear([bottleneck, alpha_cond]) raw; rgb = MLP([bottleneck, rgb_cond]) with
    sigmoid inside (hypernerf/models.py:164,288).  Conditions are (B,C), repeated over S.
    """
    s = x.shape[1]
    t = mlp(p, f"{prefix}.trunk_mlp", x, depth=trunk_depth, skips=skips, out_act="relu")
    b = _linear(p, f"{prefix}.bottleneck_mlp", t)
    a_in = b if alpha_cond is None else torch.cat(
        [b, alpha_cond[:, None, :].expand(-1, s, -1)], dim=-1)
    alpha = _linear(p, f"{prefix}.alpha_mlp", a_in)
    r_in = b if rgb_cond is None else torch.cat(
        [b, rgb_cond[:, None, :].expand(-1, s, -1)], dim=-1)
    rgb = mlp(p, f"{prefix}.rgb_mlp", r_in, depth=rgb_depth, skips=skips, out_act="sigmoid")
    return rgb, alpha


# --------------------------------------------------------------------------------------
# sampling + rendering (hypernerf/model_utils.py)
# --------------------------------------------------------------------------------------
def sample_along_rays(origins: Tensor, directions: Tensor, n: int, near: float, far: float,
                      t_rand: Optional[Tensor], lindisp: bool = False):
    """hypernerf/model_utils.py:6-41.  `t_rand` None => non-stratified branch (:36-38)."""
    t = torch.linspace(0.0, 1.0, n, dtype=origins.dtype)
    if not lindisp:
        z = near * (1.0 - t) + far * t
    else:
        z = 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    if t_rand is not None:
        mids = 0.5 * (z[1:] + z[:-1])
        upper = torch.cat([mids, z[-1:]])
        lower = torch.cat([z[:1], mids])
        z = lower + (upper - lower) * t_rand
    else:
        z = z[None, :].expand(origins.shape[0], n)
    pts = origins[:, None, :] + z[:, :, None] * directions[:, None, :]
    return z, pts


def median_depth_index(weights: Tensor, thresh: float = 0.5) -> Tensor:
    """First sample index with cumsum(w) >= thresh, 0 if none.

    hypernerf/model_utils.py:319-345 (opaqueness mask = xor of the shifted step; argmax).
    """
    opaque = torch.cumsum(weights, dim=-1) >= thresh
    prev = torch.cat([torch.zeros_like(opaque[..., :1]), opaque[..., :-1]], dim=-1)
    mask = torch.logical_xor(opaque, prev)
    return mask, torch.argmax(mask.to(weights.dtype), dim=-1)


def volumetric_rendering(rgb: Tensor, sigma: Tensor, z: Tensor, dirs: Tensor,
                         white_bg: bool = False, sample_at_infinity: bool = True,
                         eps: float = 1e-5):
    """hypernerf/model_utils.py:43-107 (+ compute_depth_map 347-362).

    Quirks kept: last distance 1e7 (or 1e-7) (:70); `+eps` inside the exclusive cumprod
    (:84); `acc` drops the last sample when sampling at infinity (:97-98) while the
    white-background composite uses the full sum (:93-95).
    """
    last = 1e7 if sample_at_infinity else 1e-7
    dists = torch.cat([z[..., 1:] - z[..., :-1],
                       torch.full_like(z[..., :1], last)], dim=-1)
    dists = dists * torch.norm(dirs[:, None, :], dim=-1)
    alpha = 1.0 - torch.exp(-sigma * dists)
    trans = torch.cat([torch.ones_like(alpha[..., :1]),
                       torch.cumprod(1.0 - alpha[..., :-1] + eps, dim=-1)], dim=-1)
    w = alpha * trans
    out_rgb = (w[..., None] * rgb).sum(dim=-2)
    depth = (w * z).sum(dim=-1)
    mask, _ = median_depth_index(w)
    med_depth = (mask.to(w.dtype) * z).sum(dim=-1)
    acc = w.sum(dim=-1)
    if white_bg:
        out_rgb = out_rgb + (1.0 - acc[..., None])
    if sample_at_infinity:
        acc = w[..., :-1].sum(dim=-1)
    return {"rgb": out_rgb, "depth": depth, "med_depth": med_depth, "acc": acc, "weights": w}


# When True the pdf normaliser is ATen's own fp32 `torch.sum` (bit-identical to the reference on
# the host that produced the goldens, ISA dependent elsewhere).  Tests flip it to show that the
# fp64 normaliser is the ONLY difference to the reference; everything else keeps the default.
REFERENCE_SUM = False


def pdf_cdf(weights: Tensor, eps: float = 1e-5) -> Tensor:
    """cdf = [0, cumsum((w+eps)/sum(w+eps))]; hypernerf/model_utils.py:177-180.

    Normaliser: fp64-accumulated, rounded once to fp32 (see module docstring).  Prefix
    sum: sequential fp64 accumulate of the fp32 pdf, each output rounded to fp32 — which
    is exactly what CPU `torch.cumsum` does for fp32 input (probed).
    """
    w = weights + eps
    if REFERENCE_SUM:
        norm = torch.sum(w, -1, keepdim=True)
    else:
        norm = w.double().sum(dim=-1, keepdim=True).to(w.dtype)
    pdf = w / norm
    cdf = torch.cumsum(pdf.double(), dim=-1).to(w.dtype)
    return torch.cat([torch.zeros_like(cdf[:, :1]), cdf], dim=-1)


def piecewise_constant_pdf(bins: Tensor, weights: Tensor, u: Tensor, eps: float = 1e-5):
    """Inverse-CDF sampling, hypernerf/model_utils.py:160-204 (== models/rendering.py:14-55).

    `u` is (B, N) in [0,1] — random draws, or linspace(0,1,N) for the deterministic branch.
    Returns (samples, inds) where inds = searchsorted(cdf, u, right=True) (int64).
    """
    n_bins = weights.shape[1]
    cdf = pdf_cdf(weights.detach(), eps)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, n_bins)
    cdf0, cdf1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf1 - cdf0
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    samples = b0 + (u - cdf0) / denom * (b1 - b0)
    return samples.detach(), inds


def sample_pdf(bins, weights, origins, directions, z, u):
    """hypernerf/model_utils.py:206-232: merge + sort, then points."""
    zs, inds = piecewise_constant_pdf(bins, weights, u)
    z_all, _ = torch.sort(torch.cat([z, zs], dim=-1), dim=-1)
    pts = origins[:, None, :] + z_all[..., None] * directions[:, None, :]
    return z_all, pts, inds


# --------------------------------------------------------------------------------------
# NerfModel (hypernerf/models.py)
# --------------------------------------------------------------------------------------
class ModelCfg:
    """The constructor arguments of NerfModel, hypernerf/models.py:111-127."""

    def __init__(self, near=0.0, far=1.0, n_samples_coarse=64, n_samples_fine=128,
                 noise_std=None, use_warp=True, use_nerf_embed=True, use_alpha_cond=True,
                 use_rgb_cond=False, hyper_slice_method=None, hyper_slice_out_dim=4,
                 GLO_dim=8, share_GLO=True, xyz_fourier_dim=10, hyper_fourier_dim=6,
                 view_fourier_dim=4, warp_kind="translation"):
        # warp_kind is not a reference constructor argument: the reference hard-codes TranslationField
        # (models.py:234); "se3" stands for BASELINE config 5 (`model.warp_field = SE3Field(...)`).
        self.warp_kind = warp_kind
        self.near, self.far = near, far
        self.nc, self.nf = n_samples_coarse, n_samples_fine
        self.noise_std = noise_std
        self.use_warp = use_warp
        self.use_nerf_embed = use_nerf_embed
        self.use_alpha_cond = use_alpha_cond
        self.use_rgb_cond = use_rgb_cond
        self.slice = hyper_slice_method or "none"
        self.hyper_dim = hyper_slice_out_dim
        self.glo_dim = GLO_dim
        self.share_glo = share_GLO
        self.xyz_f, self.hyper_f, self.view_f = xyz_fourier_dim, hyper_fourier_dim, view_fourier_dim


def filter_sigma(points: Tensor, sigma: Tensor, render_opts) -> Tensor:
    """models.py:35-63: dust threshold and bounding box on the activated density."""
    if render_opts is None:
        return sigma
    if "dust_threshold" in render_opts:
        sigma = (sigma >= render_opts.get("dust_threshold", 0.0)) * sigma
    if "bounding_box" in render_opts:
        x0, x1, y0, y1, z0, z1 = render_opts["bounding_box"]
        inside = ((points[..., 0] >= x0) & (points[..., 0] <= x1) & (points[..., 1] >= y0) & (points[..., 1] <= y1)
                  & (points[..., 2] >= z0) & (points[..., 2] <= z1))
        sigma = inside * sigma
    return sigma


def _render_level(p: Params, cfg: ModelCfg, level: str, pts, z, dirs, viewdirs, idx,
                  noise, use_warp: bool, sample_at_infinity: bool, render_opts=None):
    """NerfModel.render_samples, hypernerf/models.py:587-671."""
    b, s = pts.shape[:2]
    out = {"points": pts}
    warp_embed = None
    if use_warp:
        warp_embed = glo_embed(p["warp_embed.embed.weight"], idx)          # :609-610
    hyper_embed = None
    if cfg.slice != "none":                                                # :615-622
        # share_GLO=True is the only constructible setting (models.py:167-174,186):
        # hyper_use_warp_embed == use_warp.
        if cfg.use_warp:
            hyper_embed = warp_embed
        else:
            hyper_embed = glo_embed(p["hyper_embed.embed.weight"], idx)
    we = None if warp_embed is None else warp_embed[:, None, :].expand(b, s, -1)
    he = None if hyper_embed is None else hyper_embed[:, None, :].expand(b, s, -1)

    # map_points, models.py:545-581
    if not use_warp:
        warped = pts
    else:
        if cfg.warp_kind == "se3":
            spatial = se3_field(p, "warp_field", pts)                      # warping.py:212-240 (unpinned)
        else:
            spatial = translation_field(p, "warp_field", pts, we)          # :495-512
        if cfg.slice == "axis_aligned_plane":                              # :533-534
            hyper = he
        elif cfg.slice == "bendy_sheet":                                   # :535-539
            hyper = hyper_sheet(p, "hyper_sheet_mlp", pts, he)
        else:
            hyper = None
        warped = spatial if hyper is None else torch.cat([spatial, hyper], dim=-1)

    # query_template, models.py:447-493 + get_condition_inputs 404-445
    rgb_conds = [posenc_orig(viewdirs, cfg.view_f)]
    alpha_conds = []
    if cfg.use_nerf_embed:
        if cfg.use_warp:                                                   # :425-427
            ne = glo_embed(p["warp_embed.embed.weight"], idx)
        else:
            ne = glo_embed(p["nerf_embed.embed.weight"], idx)
        if cfg.use_alpha_cond:
            alpha_conds.append(ne)
        if cfg.use_rgb_cond:
            rgb_conds.append(ne)
    alpha_cond = torch.cat(alpha_conds, -1) if alpha_conds else None
    rgb_cond = torch.cat(rgb_conds, -1)
    feat = posenc_orig(warped[..., :3], cfg.xyz_f)
    if warped.shape[-1] > 3:
        feat = torch.cat([feat, posenc_orig(warped[..., 3:], cfg.hyper_f)], dim=-1)
    prefix = "nerf_mlps_fine" if level == "fine" else "nerf_mlps_coarse"
    rgb, alpha = nerf_mlp(p, prefix, feat, alpha_cond, rgb_cond)
    if noise is not None:                                                  # model_utils.py:300-317
        alpha = alpha + noise
    sigma = F.softplus(alpha.squeeze(-1))                                  # models.py:491
    sigma = filter_sigma(pts, sigma, render_opts)                          # models.py:650 (un-warped points)

    out["warped_points"] = warped
    out.update(volumetric_rendering(rgb, sigma, z, dirs, white_bg=False,
                                    sample_at_infinity=sample_at_infinity))
    _, di = median_depth_index(out["weights"])                             # :664
    out["med_points"] = torch.gather(warped, -2, di[..., None, None])      # :668 -> (B,1,1)
    return out


def nerf_model_forward(p: Params, cfg: ModelCfg, origins, directions, idx, rng: Dict[str, Tensor],
                       viewdirs=None, use_warp=True, render_opts=None):
    """NerfModel.forward, hypernerf/models.py:673-780.

    rng: 't_rand' (B,Nc) U[0,1); 'u' (B,Nf) U[0,1); optional 'noise_coarse' (B,Nc,1),
    'noise_fine' (B,Nc+Nf,1) ALREADY multiplied by noise_std (draw order SURVEY.md §3.1).
    """
    use_warp = cfg.use_warp and use_warp
    if viewdirs is None:
        viewdirs = directions                                               # :717-720
    z, pts = sample_along_rays(origins, directions, cfg.nc, cfg.near, cfg.far, rng["t_rand"])
    coarse = _render_level(p, cfg, "coarse", pts, z, directions, viewdirs, idx,
                           rng.get("noise_coarse"), use_warp, True)
    out = {"coarse": coarse}
    if cfg.nf > 0:
        mid = 0.5 * (z[..., 1:] + z[..., :-1])                              # :752
        z2, pts2, inds = sample_pdf(mid, coarse["weights"][..., 1:-1], origins, directions,
                                    z, rng["u"])
        out["fine"] = _render_level(p, cfg, "fine", pts2, z2, directions, viewdirs, idx,
                                    rng.get("noise_fine"), use_warp, True, render_opts)   # fine level only: :768
        out["fine"]["_inds"] = inds
    return out


def mse_loss(results, targets):
    """losses.py:9-14: mean((rgb_c-gt)^2) + mean((rgb_f-gt)^2)."""
    loss = ((results["coarse"]["rgb"] - targets) ** 2).mean()
    if "fine" in results:
        loss = loss + ((results["fine"]["rgb"] - targets) ** 2).mean()
    return loss


def psnr(pred, gt):
    """metrics.py:4-13."""
    return -10.0 * torch.log10(((pred - gt) ** 2).mean())


# --------------------------------------------------------------------------------------
# legacy nerf_pl path (models/nerf.py, models/rendering.py)
# --------------------------------------------------------------------------------------
def legacy_nerf(p: Params, x: Tensor, d=8, w=256, in_xyz=63, in_dir=27, skips=(4,),
                sigma_only=False, prefix=""):
    """models/nerf.py:83-124.  Skip cat order is [input_xyz, x] BEFORE layer i (:107-110)."""
    if sigma_only:
        xyz = x
    else:
        xyz, dirs = x[..., :in_xyz], x[..., in_xyz:in_xyz + in_dir]
    h = xyz
    for i in range(d):
        if i in skips:
            h = torch.cat([xyz, h], dim=-1)
        h = torch.relu(_linear(p, f"{prefix}xyz_encoding_{i + 1}.0", h))
    sigma = _linear(p, f"{prefix}sigma", h)
    if sigma_only:
        return sigma
    fin = _linear(p, f"{prefix}xyz_encoding_final", h)
    de = torch.relu(_linear(p, f"{prefix}dir_encoding.0", torch.cat([fin, dirs], dim=-1)))
    rgb = torch.sigmoid(_linear(p, f"{prefix}rgb.0", de))
    return torch.cat([rgb, sigma], dim=-1)


def _legacy_inference(p, prefix, xyz_freqs, pts, dirs, dir_emb, z, noise, white_back,
                      weights_only):
    """Inner `inference` of render_rays, models/rendering.py:91-172."""
    b, s = pts.shape[:2]
    emb = posenc_orig(pts.reshape(-1, 3), xyz_freqs)
    if weights_only:
        sig = legacy_nerf(p, emb, sigma_only=True, prefix=prefix, in_xyz=emb.shape[-1]).view(b, s)
        rgbs = None
    else:
        de = torch.repeat_interleave(dir_emb, s, dim=0)
        o = legacy_nerf(p, torch.cat([emb, de], dim=1), prefix=prefix,
                        in_xyz=emb.shape[-1], in_dir=de.shape[-1]).view(b, s, 4)
        rgbs, sig = o[..., :3], o[..., 3]
    deltas = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], dim=-1)
    deltas = deltas * torch.norm(dirs[:, None, :], dim=-1)
    alphas = 1.0 - torch.exp(-deltas * torch.relu(sig + noise))            # :155
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1.0 - alphas + 1e-10], dim=-1)
    w = alphas * torch.cumprod(shifted, dim=-1)[:, :-1]                    # :158-159
    wsum = w.sum(dim=1)
    if weights_only:
        return None, None, w, wsum
    rgb = (w[..., None] * rgbs).sum(dim=-2)
    depth = (w * z).sum(dim=-1)
    if white_back:
        rgb = rgb + 1.0 - wsum[:, None]
    return rgb, depth, w, wsum


def legacy_render_rays(params: Sequence[Params], freqs: Sequence[int], rays: Tensor, rng: Dict,
                       N_samples=64, use_disp=False, perturb=0, noise_std=1, N_importance=0,
                       white_back=False, test_time=False):
    """render_rays, models/rendering.py:58-244.

    params = [coarse_state_dict, fine_state_dict]; freqs = (N_freqs_xyz, N_freqs_dir).
    rng: 'perturb_rand' (B,N_samples) U[0,1) if perturb>0; 'noise_coarse' (B,N_samples) and
    'noise_fine' (B,N_samples+N_importance) N(0,1) (multiplied by noise_std here, :152);
    'u' (B,N_importance) U[0,1) if perturb>0 (else linspace, :36-38).
    """
    b = rays.shape[0]
    o, d = rays[:, 0:3], rays[:, 3:6]
    near, far = rays[:, 6:7], rays[:, 7:8]
    dir_emb = posenc_orig(d, freqs[1])
    t = torch.linspace(0, 1, N_samples, dtype=rays.dtype)
    if not use_disp:
        z = near * (1 - t) + far * t
    else:
        z = 1 / (1 / near * (1 - t) + 1 / far * t)
    z = z.expand(b, N_samples)
    if perturb > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        upper = torch.cat([mid, z[:, -1:]], -1)
        lower = torch.cat([z[:, :1], mid], -1)
        z = lower + (upper - lower) * (perturb * rng["perturb_rand"])
    pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
    nz = rng["noise_coarse"] * noise_std
    rgb_c, depth_c, w_c, ws_c = _legacy_inference(params[0], "", freqs[0], pts, d, dir_emb, z, nz,
                                                  white_back, weights_only=test_time)
    res = {"opacity_coarse": ws_c}
    if not test_time:
        res["rgb_coarse"], res["depth_coarse"] = rgb_c, depth_c
    if N_importance > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        if perturb == 0:
            u = torch.linspace(0, 1, N_importance, dtype=rays.dtype).expand(b, N_importance)
        else:
            u = rng["u"]
        zs, inds = piecewise_constant_pdf(mid, w_c[:, 1:-1], u)
        z, _ = torch.sort(torch.cat([z, zs], -1), -1)
        pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
        nz = rng["noise_fine"] * noise_std
        rgb_f, depth_f, w_f, ws_f = _legacy_inference(params[1], "", freqs[0], pts, d, dir_emb, z,
                                                      nz, white_back, weights_only=False)
        res["rgb_fine"], res["depth_fine"], res["opacity_fine"] = rgb_f, depth_f, ws_f
        res["_inds"] = inds
        res["_z_fine"] = z
    return res


# --------------------------------------------------------------------------------------
# SE(3) field (BASELINE config 5) — parity UNPINNED except exp_se3 golden G13
# --------------------------------------------------------------------------------------
def skew(w: Tensor) -> Tensor:
    """Batched cross-product matrix, Modern Robotics eq. 3.30 (hypernerf/rigid_body.py:21-38)."""
    z = torch.zeros_like(w[..., 0])
    return torch.stack([torch.stack([z, -w[..., 2], w[..., 1]], -1),
                        torch.stack([w[..., 2], z, -w[..., 0]], -1),
                        torch.stack([-w[..., 1], w[..., 0], z], -1)], -2)


def exp_se3(screw: Tensor, theta: Tensor):
    """(R, p) of exp([S] theta); Modern Robotics eq. 3.88 (hypernerf/rigid_body.py:55-83).

    Batched over leading dims (the reference handles a single point only).
    """
    w, v = screw[..., :3], screw[..., 3:]
    W = skew(w)
    W2 = W @ W
    th = theta[..., None, None]
    eye = torch.eye(3, dtype=screw.dtype).expand_as(W)
    R = eye + torch.sin(th) * W + (1.0 - torch.cos(th)) * W2
    G = th * eye + (1.0 - torch.cos(th)) * W + (th - torch.sin(th)) * W2
    pvec = (G @ v[..., None])[..., 0]
    return R, pvec


def se3_field(p: Params, prefix: str, pts: Tensor, min_deg=0, max_deg=8):
    """Intended behaviour of SE3Field.warp, hypernerf/warping.py:212-240 (metadata embed
    ignored as upstream does, :223-224); formulas per rigid_body.py docstrings.  UNPINNED."""
    h = posenc_jax(pts, min_deg, max_deg, use_identity=False)
    t = mlp(p, f"{prefix}.trunk", h, depth=6)
    w = mlp(p, f"{prefix}.w_net", t, depth=0)
    v = mlp(p, f"{prefix}.v_net", t, depth=0)
    theta = torch.norm(w, dim=-1)
    w = w / theta[..., None]
    v = v / theta[..., None]
    R, pvec = exp_se3(torch.cat([w, v], dim=-1), theta)
    return (R @ pts[..., None])[..., 0] + pvec


# --------------------------------------------------------------------------------------------
# ray generation (datasets/ray_utils.py) — SURVEY.md §8 f2
# --------------------------------------------------------------------------------------------
def ray_directions(H: int, W: int, focal: float) -> Tensor:
    """get_ray_directions, datasets/ray_utils.py:5-25.  kornia.create_meshgrid(H, W, normalized_coordinates=False)
    (third-party, absent here: pixel-index grid, x = column, y = row) is restated with torch.meshgrid."""
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    return torch.stack([(i - W / 2) / focal, -(j - H / 2) / focal, -torch.ones_like(i)], -1)


def rays_from_pose(directions: Tensor, c2w: Tensor):
    """get_rays, datasets/ray_utils.py:28-49."""
    d = directions @ c2w[:, :3].T
    d = d / torch.norm(d, dim=-1, keepdim=True)
    o = c2w[:, 3].expand(d.shape)
    return o.reshape(-1, 3), d.reshape(-1, 3)


def ndc_rays(H: int, W: int, focal: float, near: float, rays_o: Tensor, rays_d: Tensor):
    """get_ndc_rays, datasets/ray_utils.py:52-93."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    ox_oz = rays_o[..., 0] / rays_o[..., 2]
    oy_oz = rays_o[..., 1] / rays_o[..., 2]
    o0 = -1. / (W / (2. * focal)) * ox_oz
    o1 = -1. / (H / (2. * focal)) * oy_oz
    o2 = 1. + 2. * near / rays_o[..., 2]
    d0 = -1. / (W / (2. * focal)) * (rays_d[..., 0] / rays_d[..., 2] - ox_oz)
    d1 = -1. / (H / (2. * focal)) * (rays_d[..., 1] / rays_d[..., 2] - oy_oz)
    d2 = 1 - o2
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


def image_rays(H: int, W: int, focal: float, c2w: Tensor, near: float, far: float, ndc: bool, image_id=None):
    """The (H*W, 8|9) ray rows datasets/llff.py:244-264 builds for one image."""
    o, d = rays_from_pose(ray_directions(H, W, focal), c2w)
    if ndc:
        o, d = ndc_rays(H, W, focal, 1.0, o, d)
    cols = [o, d, near * torch.ones_like(o[:, :1]), far * torch.ones_like(o[:, :1])]
    if image_id is not None:
        cols.append(float(image_id) * torch.ones_like(o[:, :1]))
    return torch.cat(cols, 1)
