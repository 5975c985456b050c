"""Diagnostic for the opt-in 8-bit stash (precision 'bf16s8'): one forward+backward of a small NerfModel in 'bf16' and in
'bf16s8' on the same inputs.  Outputs and input-side gradients must be identical (the forward and backward-data
machines do not change); the weight gradients differ by the rounding of the stash: prints the relative L2 distance
over the whole gradient buffer and per parameter."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import hashprng as H
import hypernerf_torch_amd as HN
from hypernerf_torch_amd import functional as F
from hypernerf_torch_amd.hypernerf import models, model_utils
from hypernerf_torch_amd.losses import MSELoss
from gpu_common import DEV, EMB, load_hash, rays_for

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
res = {}
for prec in ("bf16", "bf16s8"):
    HN.set_precision(prec)
    m = models.NerfModel(EMB, n_samples_coarse=nc, n_samples_fine=nf, noise_std=None, hyper_slice_method="bendy_sheet",
                         use_warp=True, use_nerf_embed=True, use_alpha_cond=True).to(DEV)
    load_hash(m, 5)
    arena = HN.ParamArena(m.parameters())
    o, d, idx = rays_for(5, B)
    rays = torch.cat([o, d, torch.zeros(B, 1), torch.ones(B, 1), idx.float()[:, None]], dim=1).to(DEV)
    gt = H.uniform(5, "gt", (B, 3), 0, 1).to(DEV)
    rng = {"t_rand": H.uniform(5, "t", (B, nc), 0, 1).to(DEV), "u": H.uniform(5, "u", (B, nf), 0, 1).to(DEV)}
    out = m(model_utils.prepare_ray_dict(rays), {}, rng=rng)
    loss = MSELoss()(out, gt)
    arena.zero_grad()
    F.backward(loss)
    torch.cuda.synchronize()
    names = [n for n, _ in m.named_parameters()]
    res[prec] = (out["fine"]["rgb"].detach().clone(), arena.grad.clone(),
                 {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}, float(loss))
a, b = res["bf16"], res["bf16s8"]
print("loss", a[3], b[3], "rgb max diff", float((a[0] - b[0]).abs().max()))
g0, g1 = a[1], b[1]
print("finite", bool(torch.isfinite(g1).all()), "rel L2 of the gradient buffer", float((g1 - g0).norm() / g0.norm()),
      "max abs", float((g1 - g0).abs().max()), "scale", float(g0.abs().max()))
e = g1 - g0
print("projection of the error on the gradient  <e, g> / <g, g> =", float((e * g0).sum() / (g0 * g0).sum()),
      "  signed / absolute error sum", float(e.sum() / e.abs().sum()))
worst = sorted(((float((b[2][n] - a[2][n]).norm() / (a[2][n].norm() + 1e-30)), n) for n in a[2]), reverse=True)
for r, n in worst[:12]:
    print(f"  {r:.4f}  {n}  |g| {float(a[2][n].norm()):.3e}")
