"""Diagnostic (HN_PROF build): shader-clock trace of wave 0 / workgroup 0 of the template forward machine.
Stamp codes: 100+op = LAYER op start, 1 = features done, 2 = tile MFMAs issued, 3 = tile epilogue done (results in
registers), 4 = tile stash issued."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import hypernerf_torch_amd as HN
from hypernerf_torch_amd import _lib as L
from hypernerf_torch_amd.hypernerf.models import NerfModel
from gpu_common import EMB, rays_for

HN.set_precision("bf16")
m = NerfModel(EMB, n_samples_coarse=64, n_samples_fine=64, noise_std=1.0, hyper_slice_method="bendy_sheet",
              use_nerf_embed=True, use_alpha_cond=True, view_fourier_dim=6).cuda()
o, d, idx = rays_for(1, 1024)
rays = {"origins": o.cuda(), "directions": d.cuda(), "viewdirs": None,
        "metadata": {k: idx.cuda() for k in ("warp", "camera", "appearance", "time")}}
for _ in range(2):
    out = m(rays, {}); (out["fine"]["rgb"].sum() + out["coarse"]["rgb"].sum()).backward()
torch.cuda.synchronize()
WHICH = sys.argv[1] if len(sys.argv) > 1 else "hn_mlp_forward"      # or hn_mlp_backward
L.PROF_BUFFER = torch.zeros(4096, dtype=torch.int64, device="cuda")
orig = L.launch
traces = {}
def launch(name, *a, tag=""):
    if name == WHICH:
        L.PROF_BUFFER.zero_(); torch.cuda.synchronize()
        orig(name, *a, tag=tag); torch.cuda.synchronize()
        traces[tag] = L.PROF_BUFFER.cpu().view(-1, 2).tolist()
    else:
        orig(name, *a, tag=tag)
L.launch = launch
out = m(rays, {}); (out["fine"]["rgb"].sum() + out["coarse"]["rgb"].sum()).backward()
torch.cuda.synchronize()
for tag, tr in traces.items():
    tr = [(c, t) for c, t in tr if c != 0]
    if not tr: continue
    print(f"== {tag}: {len(tr)} stamps, total {tr[-1][1] - tr[0][1]} cycles")
    prev = tr[0][1]; line = []
    for c, t in tr:
        if c >= 100:
            if line: print("   " + " ".join(line)); line = []
            print(f" op {c % 100:3d} (code {c // 100}) @ +{t - tr[0][1]}")
        line.append(f"{c}:{t - prev}")
        prev = t
    if line: print("   " + " ".join(line))
