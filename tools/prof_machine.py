"""Diagnostic: in-kernel shader-clock breakdown of the MLP machine (forward / backward) at config-2 sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import hypernerf_torch_amd as HN
from hypernerf_torch_amd import _lib as L
from hypernerf_torch_amd.hypernerf.models import NerfModel
from gpu_common import EMB, rays_for

HN.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
m = NerfModel(EMB, n_samples_coarse=64, n_samples_fine=64, noise_std=1.0, hyper_slice_method="bendy_sheet",
              use_nerf_embed=True, use_alpha_cond=True, view_fourier_dim=6).cuda()
o, d, idx = rays_for(1, 1024)
rays = {"origins": o.cuda(), "directions": d.cuda(), "viewdirs": None,
        "metadata": {k: idx.cuda() for k in ("warp", "camera", "appearance", "time")}}
for _ in range(2):
    out = m(rays, {}); (out["fine"]["rgb"].sum() + out["coarse"]["rgb"].sum()).backward()
torch.cuda.synchronize()
orig = L.launch
names = ["total", "layers", "wait_vmcnt", "barrier", "waves"]
def wrapped(name, *a, tag=""):
    if name in ("hn_mlp_forward", "hn_mlp_backward"):
        L.PROF_BUFFER = torch.zeros(16, dtype=torch.int64, device="cuda")
        # args struct was built before PROF_BUFFER existed for this call: rebuild not possible here, so we set it
        # globally one call ahead (all launches of this pass carry the pointer)
    orig(name, *a, tag=tag)
L.PROF_BUFFER = torch.zeros(16, dtype=torch.int64, device="cuda")
res = {}
import ctypes
def launch(name, *a, tag=""):
    if name in ("hn_mlp_forward", "hn_mlp_backward"):
        L.PROF_BUFFER.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(name, *a, tag=tag); e1.record(); torch.cuda.synchronize()
        v = L.PROF_BUFFER.cpu().tolist()
        res.setdefault(f"{name}[{tag}]", []).append((e0.elapsed_time(e1), v))
    else:
        orig(name, *a, tag=tag)
L.launch = launch
import hypernerf_torch_amd.machine as MM
out = m(rays, {}); (out["fine"]["rgb"].sum() + out["coarse"]["rgb"].sum()).backward()
torch.cuda.synchronize()
for k, lst in res.items():
    for ms, v in lst:
        w = max(1, v[4])
        print(f"{k:42s} {ms*1e3:8.1f} us  waves={w:6d}  per-wave cycles: total={v[0]/w:9.0f} layers={v[1]/w:9.0f} "
              f"wait_vmcnt={v[2]/w:8.0f} barrier={v[3]/w:8.0f} feat={v[5]/w:8.0f} gemm(incl wait/barrier)={v[6]/w:8.0f} epi={v[7]/w:8.0f} prologue={v[8]/w:8.0f}")
