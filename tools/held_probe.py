"""Diagnostic: duration of the two weight-gradient launches of the data-parallel overlap (dist.GradSync) on one GPU."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench                                                      # noqa: E402
import hypernerf_torch_amd as HN                                  # noqa: E402
from hypernerf_torch_amd import _lib as L                         # noqa: E402
from hypernerf_torch_amd import functional as F                   # noqa: E402
from hypernerf_torch_amd.dist import GradSync                     # noqa: E402

a = types.SimpleNamespace(rays=1024, nc=64, nf=64, kind="hypernerf", precision="bf16")
dev = torch.device("cuda", 0)
HN.set_precision("bf16")
fwd_bwd, params, programs, workload, model = bench.build_workload(a, dev, 0)
arena = HN.ParamArena(params)
sync = GradSync(arena, model, tail_prefix=os.environ.get('HN_TAIL', 'nerf_mlps_'))
print('split at', sync.split, 'of', arena.numel)
for split in (False, True):
    for it in range(4):
        L.KERNEL_TIMES = {} if it == 3 else None
        if split:
            with sync.splitting():
                fwd_bwd()
            F.flush_held_wgrads()
        else:
            fwd_bwd()
        torch.cuda.synchronize()
    t = L.collect_kernel_times()
    L.KERNEL_TIMES = None
    print("split" if split else "one  ", {k: [round(x, 4) for x in v] for k, v in t.items() if "wgrad" in k})
    arena.zero_grad()
