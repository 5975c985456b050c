"""Per-CU timeline of hn_wgrad_kernel (diagnostic build -DHN_WGRAD_JOBTIMES=1: tools/build_variant.sh
tools/variants/jobtimes.so HN_WGRAD_JOBTIMES=1; run with HN_LIB_PATH=tools/variants/jobtimes.so).  Every job of the batched
launch stamps the 100 MHz wall clock at its entry, when its first LDS stage has landed, behind its last product and behind
its flush, together with the CU it ran on.  From that: how much of the launch x CUs is (a) ramp — entry to first stage
ready —, (b) flush — accumulator slabs and bias sums —, (c) the gap between a job's end and the next job's entry on the
same CU (workgroup turnover), (d) the idle tail of a CU behind its last job; (a) + (b) + (c) is the most a persistent form
of the kernel (<= 256 workgroups walking the job list, the next job's first stage issued under the current flush) could
recover (verdict r05 item 3).    python tools/wg_jobtimes.py [config: 2 | 3] [launches]"""
import ctypes
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/tests']
import numpy as np
import torch
import hypernerf_torch_amd as HN
from hypernerf_torch_amd import _lib as L
from hypernerf_torch_amd.hypernerf.models import NerfModel
from gpu_common import EMB, rays_for

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
b, nc, nf = (1024, 64, 64) if cfg == 2 else (16384, 64, 128)
HN.set_precision("bf16")
m = NerfModel(EMB, n_samples_coarse=nc, n_samples_fine=nf, noise_std=1.0, hyper_slice_method="bendy_sheet",
              use_nerf_embed=True, use_alpha_cond=True, view_fourier_dim=6).cuda()
arena = HN.ParamArena(m.parameters())
o, d, idx = rays_for(1, b)
rays = {"origins": o.cuda(), "directions": d.cuda(), "viewdirs": None,
        "metadata": {k: idx.cuda() for k in ("warp", "camera", "appearance", "time")}}


def step():
    out = m(rays, {})
    (out["fine"]["rgb"].sum() + out["coarse"]["rgb"].sum()).backward()
    arena.zero_grad()


for _ in range(3):
    step()
torch.cuda.synchronize()
lib = L.load()
if not hasattr(lib, "hn_set_wgrad_prof"):
    sys.exit("this library has no hn_set_wgrad_prof: build it with -DHN_WGRAD_JOBTIMES=1 and point HN_LIB_PATH at it")
buf = torch.zeros((1 << 17) * 8, dtype=torch.int64, device="cuda")      # 8 int64 per job; config 3 cuts ~10 k jobs
lib.hn_set_wgrad_prof(ctypes.c_void_p(buf.data_ptr()))
acc = []
for _ in range(reps):
    buf.zero_()
    step()
    torch.cuda.synchronize()
    r = buf.cpu().view(-1, 8).numpy()
    r = r[r[:, 7] == 1]
    hw, xcc = r[:, 0] & 0xffffffff, r[:, 0] >> 32
    # gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; one workgroup per CU at a time (128 KiB of LDS)
    cu = ((hw >> 8) & 0xff) | (xcc << 8)
    t0, t1, t2, t3 = r[:, 1], r[:, 2], r[:, 3], r[:, 4]
    start, end = t0.min(), t3.max()
    span = float(end - start)
    cus = np.unique(cu)
    ramp = float((t1 - t0).sum())
    flush = float((t3 - t2).sum())
    busy = float((t3 - t0).sum())
    gap = tail = head = 0.0
    for c in cus:
        sel = np.argsort(t0[cu == c])
        a0, a3 = t0[cu == c][sel], t3[cu == c][sel]
        head += float(a0[0] - start)
        gap += float((a0[1:] - a3[:-1]).sum())
        tail += float(end - a3[-1])
    tot = span * len(cus)
    acc.append(dict(jobs=len(r), cus=len(cus), span_us=span / 100.0, ramp=ramp / tot, flush=flush / tot, turnover=gap / tot,
                    head=head / tot, tail=tail / tot, compute=(busy - ramp - flush) / tot,
                    ramp_us_per_job=ramp / len(r) / 100.0, flush_us_per_job=flush / len(r) / 100.0,
                    turnover_us_per_gap=gap / max(1, len(r) - len(cus)) / 100.0, GB=float(r[:, 5].sum()) / 1e9))
# per rectangle shape (dZ tiles x X tiles of the job's wave grid): how fast its jobs stream — the weights a time-based
# job order would use (last launch)
shapes = {}
for sh in np.unique(r[:, 6]):
    sel = r[:, 6] == sh
    dur = (t3 - t0)[sel].astype(np.float64) / 100.0
    mb = r[sel, 5].astype(np.float64) / 1e6
    shapes[f"{int(sh) >> 4}x{int(sh) & 15}"] = dict(jobs=int(sel.sum()), mean_mb=float(mb.mean()), mean_us=float(dur.mean()),
                                                  gb_per_s=float((mb.sum() / 1e3) / (dur.sum() / 1e6)),
                                                  start_us_mean=float(((t0[sel] - start) / 100.0).mean()))
med = {k: float(np.median([a[k] for a in acc])) for k in acc[0]}
med["per_shape_last_launch"] = shapes
med["recoverable_by_a_persistent_form"] = med["ramp"] + med["flush"] + med["turnover"]
med["config"], med["launches"], med["build"] = cfg, reps, L.build_id()
print(json.dumps(med))
print(f"# config {cfg}: {med['jobs']:.0f} jobs on {med['cus']:.0f} CUs, launch span {med['span_us']:.1f} us (first entry .. last exit), "
      f"{med['GB']:.2f} GB of stash tiles")
print(f"# share of span x CUs:  streaming + products {med['compute']:.3f} | ramp (entry -> first stage landed) {med['ramp']:.3f} "
      f"({med['ramp_us_per_job']:.1f} us per job) | flush {med['flush']:.3f} ({med['flush_us_per_job']:.1f} us per job) | "
      f"workgroup turnover on a CU {med['turnover']:.3f} ({med['turnover_us_per_gap']:.1f} us per gap) | "
      f"before a CU's first job {med['head']:.3f} | idle tail behind its last job {med['tail']:.3f}")
for k, v in sorted(shapes.items(), key=lambda kv: -kv[1]["jobs"] * kv[1]["mean_mb"]):
    print(f"#   shape {k:>5}: {v['jobs']:4d} jobs, {v['mean_mb']:6.2f} MB and {v['mean_us']:6.1f} us each = {v['gb_per_s']:5.1f} GB/s per CU, "
          f"mean start {v['start_us_mean']:6.1f} us")
print(f"# ceiling of a persistent form (ramp + flush + turnover hidden completely): {med['recoverable_by_a_persistent_form']:.3f} of the launch")
