"""Diagnostic: where the host time of an EAGER training step goes (cProfile over 30 steps of bench.py's workload)."""
import cProfile
import os
import pstats
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench                                                      # noqa: E402
import hypernerf_torch_amd as HN                                  # noqa: E402

a = types.SimpleNamespace(rays=1024, nc=64, nf=64, kind="hypernerf", precision="bf16")
dev = torch.device("cuda", 0)
HN.set_precision("bf16")
fwd_bwd, params, programs, workload, model = bench.build_workload(a, dev, 0)
arena = HN.ParamArena(params)
opt = HN.ArenaAdam(arena, lr=5e-4)


def step():
    fwd_bwd()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    step()
torch.cuda.synchronize()
print("eager ms/step", round((time.perf_counter() - t0) / 30 * 1e3, 3))
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
