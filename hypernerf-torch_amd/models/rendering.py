"""MI355X drop-in for the reference's legacy `models/rendering.py`: nerf_pl `render_rays`.

Same signature and returned keys as models/rendering.py:58-244.  Per level: one fused launch of the
NeRF machine (both positional encoders generated in-kernel), one compositing launch; inverse-CDF
sampling + merge sort in one kernel between the levels.
"""
from __future__ import annotations

import torch

from .. import _lib as L
from .. import functional as F

__all__ = ['render_rays', 'sample_pdf']

_T_CACHE = {}


def _t_vals(n, device):
    key = (n, str(device))
    if key not in _T_CACHE:
        t = torch.linspace(0, 1, n)                 # host ATen, as the reference's CPU path
        _T_CACHE[key] = (t.to(device), (1 - t).to(device))
    return _T_CACHE[key]


def sample_pdf(bins, weights, N_importance, det=False, eps=1e-5, u=None):
    """Inverse-CDF samples (reference: models/rendering.py:14-55)."""
    if eps != 1e-5:
        raise NotImplementedError("eps is fixed to 1e-5 in the HIP kernel")
    b = weights.shape[0]
    if u is None:
        if det:
            u = torch.linspace(0, 1, N_importance).to(bins.device).expand(b, N_importance).contiguous()
        else:
            u = torch.rand(b, N_importance, device=bins.device)
    _, _, _, zs = F.sample_pdf(weights, None, u, bins=bins, merge=False)
    return zs


def render_rays(models, embeddings, rays, N_samples=64, use_disp=False, perturb=0, noise_std=1, N_importance=0,
                chunk=1024 * 32, white_back=False, test_time=False, rng=None):
    """Render rays with the coarse (and fine) NeRF (reference: models/rendering.py:58-244).

    rays: (N_rays, 3+3+2).  Returns rgb_/depth_/opacity_{coarse,fine}.  `chunk` is accepted for API
    compatibility; the fused kernels need no point chunking.  `rng` optionally supplies the draws
    ('perturb_rand' (B,N), 'noise_coarse' (B,N), 'u' (B,Nimp), 'noise_fine' (B,N+Nimp), N(0,1) unscaled)."""
    rng = rng or {}
    L.require_gpu(rays)
    model_coarse = models[0]
    emb_xyz, emb_dir = embeddings[0], embeddings[1]
    n_rays = rays.shape[0]
    if n_rays == 0:         # zero rays in, zero rays out (no launch)
        res = {'opacity_coarse': rays.new_zeros(0)}
        for typ in ([] if test_time else ['coarse']) + (['fine'] if N_importance > 0 else []):
            res[f'rgb_{typ}'], res[f'depth_{typ}'], res[f'opacity_{typ}'] = rays.new_zeros(0, 3), rays.new_zeros(0), rays.new_zeros(0)
        return res
    rays_d = rays[:, 3:6]
    t, omt = _t_vals(N_samples, rays.device)
    t_rand = None
    if perturb > 0:
        t_rand = rng.get('perturb_rand')
        if t_rand is None:
            t_rand = torch.rand(n_rays, N_samples, device=rays.device)
    z_vals, xyz = F.sample_legacy(rays, t, omt, use_disp, t_rand, float(perturb))

    def inference(model, pts, z, noise, weights_only):
        s = z.shape[1]
        call = model.fused_call(emb_xyz, emb_dir, weights_only)
        (o,) = F.run_program(call, [pts.reshape(-1, 3), rays_d], s)
        if noise is None:
            noise = torch.randn(n_rays, s, device=rays.device)      # drawn on every call (rendering.py:152)
        if weights_only:
            dummy_rgb = torch.zeros(n_rays, s, 3, device=rays.device)
            res = F.composite(dummy_rgb, o.view(n_rays, s), noise, z, rays_d, None, variant=1, white_bg=False,
                              sample_at_infinity=True, want_median=False, noise_scale=float(noise_std))
            return None, None, res[3], res[2]
        o = o.view(n_rays, s, 4)
        res = F.composite(o[..., :3], o[..., 3], noise, z, rays_d, None, variant=1, white_bg=white_back,
                          sample_at_infinity=True, want_median=False, noise_scale=float(noise_std))
        return res[0], res[1], res[3], res[2]

    rgb_c, depth_c, w_c, op_c = inference(model_coarse, xyz, z_vals, rng.get('noise_coarse'), test_time)
    if test_time:
        result = {'opacity_coarse': op_c}
    else:
        result = {'rgb_coarse': rgb_c, 'depth_coarse': depth_c, 'opacity_coarse': op_c}

    if N_importance > 0:
        if perturb == 0:
            u = torch.linspace(0, 1, N_importance).to(rays.device).expand(n_rays, N_importance).contiguous()
        else:
            u = rng.get('u')
            if u is None:
                u = torch.rand(n_rays, N_importance, device=rays.device)
        z_fine, xyz_fine, inds, _ = F.sample_pdf(w_c, z_vals, u, rays[:, 0:3], rays_d)
        rgb_f, depth_f, w_f, op_f = inference(models[1], xyz_fine, z_fine, rng.get('noise_fine'), False)
        result['rgb_fine'], result['depth_fine'], result['opacity_fine'] = rgb_f, depth_f, op_f
        render_rays.last_sampling = {'z_coarse': z_vals, 'z_fine': z_fine, 'inds': inds}
    return result
