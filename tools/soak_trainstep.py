"""Diagnostic soak: TrainStep (HIP-graph replays) for many steps on the synthetic scene of tools/psnr_parity.py —
device memory, host RSS and file descriptors must not grow, the loss must keep falling, the LR schedule must arrive.

usage: python tools/soak_trainstep.py [steps=20000]"""
import json
import os
import resource
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import hypernerf_torch_amd as HN                                  # noqa: E402
from hypernerf_torch_amd.hypernerf.models import NerfModel        # noqa: E402
from hypernerf_torch_amd.training import TrainStep                # noqa: E402
from gpu_common import EMB                                        # noqa: E402
import psnr_parity                                                # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = "cuda:0"
HN.set_precision(os.environ.get("HN_PRECISION", "bf16"))      # HN_PRECISION=bf16s8: soak the opt-in 8-bit-stash mode
torch.manual_seed(0)
m = NerfModel(EMB, near=0.2, far=2.0, n_samples_coarse=64, n_samples_fine=64, noise_std=1.0,
              hyper_slice_method="bendy_sheet", use_warp=True, use_nerf_embed=True, use_alpha_cond=True,
              view_fourier_dim=6).to(dev)
g = torch.Generator().manual_seed(7)
rays, col = (x.to(dev) for x in psnr_parity.scene(65536, g))
ts = TrainStep(m, lr=1e-3, decay_step=[1, 2], decay_gamma=0.3)
gd = torch.Generator(device=dev).manual_seed(3)
marks = []
t0 = time.perf_counter()
for it in range(steps):
    sel = torch.randint(0, rays.shape[0], (1024,), device=dev, generator=gd)
    log = ts.step(rays[sel], col[sel])
    if it in (steps // 3, 2 * steps // 3):
        ts.epoch_end()
    if it % (steps // 10) == 0 or it == steps - 1:
        torch.cuda.synchronize()
        marks.append({"it": it, "loss": float(log["train/loss"]), "lr": log["lr"],
                      "cuda_alloc_mib": torch.cuda.memory_allocated() >> 20, "cuda_reserved_mib": torch.cuda.memory_reserved() >> 20,
                      "host_maxrss_mib": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10,
                      "fds": len(os.listdir("/proc/self/fd")), "steps_done": float(ts.optimizer.step_count)})
dt = time.perf_counter() - t0
ok = (marks[-1]["cuda_alloc_mib"] <= marks[1]["cuda_alloc_mib"] + 1 and marks[-1]["fds"] == marks[1]["fds"]
      and marks[-1]["host_maxrss_mib"] <= marks[1]["host_maxrss_mib"] + 64 and marks[-1]["loss"] < 0.2 * marks[0]["loss"]
      and marks[-1]["steps_done"] == steps)
print(json.dumps({"steps": steps, "seconds": round(dt, 1), "ms_per_step_incl_host": round(1e3 * dt / steps, 3), "ok": ok, "marks": marks}))
