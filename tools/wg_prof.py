"""Diagnostic (HN_PROF build): where hn_wgrad_kernel's stages spend their cycles — wait for the LDS-DMA, stage barrier,
issue of the next stage's DMA, LDS reads + MFMAs — for wave 0 of every 97th workgroup of the batched launch."""
import os, sys, ctypes
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[ROOT, ROOT+'/tests']
import torch
import hypernerf_torch_amd as HN
from hypernerf_torch_amd import _lib as L
from hypernerf_torch_amd.hypernerf.models import NerfModel
from gpu_common import EMB, rays_for
HN.set_precision(os.environ.get("HN_PRECISION", "bf16"))
m = NerfModel(EMB, n_samples_coarse=64, n_samples_fine=64, noise_std=1.0, hyper_slice_method="bendy_sheet", use_nerf_embed=True, use_alpha_cond=True, view_fourier_dim=6).cuda()
arena = HN.ParamArena(m.parameters())
o, d, idx = rays_for(1, 1024)
rays = {"origins": o.cuda(), "directions": d.cuda(), "viewdirs": None, "metadata": {k: idx.cuda() for k in ("warp", "camera", "appearance", "time")}}
def step():
    out = m(rays, {}); (out["fine"]["rgb"].sum() + out["coarse"]["rgb"].sum()).backward()
for _ in range(2): step()
torch.cuda.synchronize()
buf = torch.zeros(64*8, dtype=torch.int64, device="cuda")
lib = ctypes.CDLL(L.LIB_PATH)
lib.hn_set_wgrad_prof(ctypes.c_void_p(buf.data_ptr()))
step(); torch.cuda.synchronize()
r = buf.cpu().view(-1,8)
tot=[0,0,0,0]
for row in r.tolist():
    if row[4]==0: continue
    tw,tb,ti,tc,ns,shape,bps,bid=row
    s=tw+tb+ti+tc
    print(f"blk {bid:4d} rect {shape>>4}x{shape&15} bps {bps} stages {ns:4d} cycles/stage {s/ns:7.0f}  wait {tw/s:.2f} barrier {tb/s:.2f} issue {ti/s:.2f} compute {tc/s:.2f}")
