"""Per-kernel timings of one config-2 step (eager, HIP events): training vs inference (no stash/masks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import hypernerf_torch_amd as HN
from hypernerf_torch_amd import _lib as L
from hypernerf_torch_amd.hypernerf.models import NerfModel
from gpu_common import EMB, rays_for

HN.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
m = NerfModel(EMB, n_samples_coarse=64, n_samples_fine=64, noise_std=1.0, hyper_slice_method="bendy_sheet",
              use_nerf_embed=True, use_alpha_cond=True, view_fourier_dim=6).cuda()
o, d, idx = rays_for(1, B)
rays = {"origins": o.cuda(), "directions": d.cuda(), "viewdirs": None,
        "metadata": {k: idx.cuda() for k in ("warp", "camera", "appearance", "time")}}

def run(train, n=10):
    for _ in range(3):
        if train:
            out = m(rays, {}); (out["fine"]["rgb"].sum() + out["coarse"]["rgb"].sum()).backward()
        else:
            with torch.no_grad(): m(rays, {})
    torch.cuda.synchronize()
    L.KERNEL_TIMES = {}
    for _ in range(n):
        if train:
            out = m(rays, {}); (out["fine"]["rgb"].sum() + out["coarse"]["rgb"].sum()).backward()
        else:
            with torch.no_grad(): m(rays, {})
    t = L.collect_kernel_times(); L.KERNEL_TIMES = None
    return {k: sum(v) / n for k, v in t.items()}

tr, ev = run(True), run(False)
print(f"{'kernel':40s} {'train ms':>9s} {'eval ms':>9s}")
for k in sorted(tr, key=tr.get, reverse=True):
    print(f"{k:40s} {tr[k]:9.3f} {ev.get(k, float('nan')):9.3f}")
print(f"{'SUM':40s} {sum(tr.values()):9.3f} {sum(ev.values()):9.3f}")
