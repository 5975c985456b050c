"""Which jobs of the batched weight-gradient launch stream at what rate?  Runs hn_mlp_wgrad_batched_t on the job table of
the config-2 fine level, one tile-rectangle shape at a time (garbage stash: only the time matters), kernel-timeline
timed.   python tools/wgrad_shape_probe.py [n_points]"""
import collections, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import hypernerf_torch_amd as HN
from hypernerf_torch_amd import _lib as L, machine as M
from hypernerf_torch_amd.hypernerf.models import NerfModel
from gpu_common import EMB
n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
d = torch.device("cuda:0")
L.load()
m = NerfModel(EMB, n_samples_coarse=64, n_samples_fine=64, noise_std=1.0, view_fourier_dim=6, hyper_slice_method="bendy_sheet",
              use_nerf_embed=True, use_alpha_cond=True)
prog = m._level_call("fine").program
mode = L.HN_MODE_BF16
_, sb, _ = prog.layout(mode, n_points)
stash = torch.zeros(sb, dtype=torch.uint8, device=d)
_, gtot = prog.grad_offsets()
grads = torch.zeros(gtot, device=d)
total = prog.wgrad_stream_bytes(mode, n_points)
jobs = prog.wgrad_jobs(mode, n_points, job_bytes=M.WGRAD_JOB_BYTES, launch_bytes=total)
shapes = collections.Counter(zip(jobs["n_nt"].tolist(), jobs["n_kt"].tolist()))
print(f"{len(jobs)} jobs, {total / 1e9:.3f} GB")


def run(sel, reps=6, partial=True):
    part = np.ascontiguousarray(jobs[sel])
    tl = M.slab_tiles(part)
    part["p_tile"] = np.cumsum(tl) - tl
    jd = L.to_device_bytes(part, d)
    w = (part["n_nt"] + part["n_kt"]).astype(np.int64) * (part["blk1"] - part["blk0"])
    order = torch.from_numpy(np.argsort(-w, kind="stable").astype(np.int32)).to(d)
    slab = torch.empty(int(tl.sum()) * 1024, device=d)
    arr = (L.HnDwBatch * 1)()
    arr[0].jobs, arr[0].stash, arr[0].grads, arr[0].n_jobs = jd.data_ptr(), stash.data_ptr(), grads.data_ptr(), len(part)
    arr[0].partials = slab.data_ptr() if partial else 0
    t = torch.zeros(8, dtype=torch.int64, device=d)
    for i in range(reps + 2):
        if i == 2:
            torch.cuda.synchronize(); t.zero_()
        L.launch("hn_mlp_wgrad_batched_t", C.c_int(mode), arr, C.c_int(1), L.ptr(order), L.ptr(t), L.stream_handle())
    torch.cuda.synchronize()
    v = t.cpu().tolist()
    return w.sum() * 2048, v[4] * 1e-8 / v[5], len(part)


b, s, n = run(np.ones(len(jobs), dtype=bool))
print(f"all shapes           {n:4d} jobs  {b / 1e9:6.3f} GB  {s * 1e3:7.4f} ms  {b / s / 1e12:5.2f} TB/s")
for (nn, nk), cnt in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][1]):
    sel = (jobs["n_nt"] == nn) & (jobs["n_kt"] == nk)
    b, s, n = run(sel)
    print(f"rect {nn} x {nk}            {n:4d} jobs  {b / 1e9:6.3f} GB  {s * 1e3:7.4f} ms  {b / s / 1e12:5.2f} TB/s   ({n / 256:.2f} jobs per CU)")
big = (jobs["n_nt"] * jobs["n_kt"] >= 32)
for name, sel in (("rects >= 32 tiles", big), ("rects < 32 tiles", ~big)):
    b, s, n = run(sel)
    print(f"{name:20s} {n:4d} jobs  {b / 1e9:6.3f} GB  {s * 1e3:7.4f} ms  {b / s / 1e12:5.2f} TB/s")
