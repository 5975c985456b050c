"""Round 6, verdict item 5 — the config-3 decision with a number measured on ONE box: the layer-stationary pipeline of round
4 (tools/ls_bench, mode 2: eight 256 -> 256 stages x 32 CUs, W^T and dW resident, dZ handed CU to CU; dX = W^T dZ and
dW += dZ X^T from the same dZ bytes) against what the PRODUCT's two kernels — hn_mlp_bwd_kernel and hn_wgrad_kernel —
spend on the same eight plain 256 -> 256 layers at the same number of points.  The product's share is isolated by
difference: an MLP chain of 8 plain layers between a 32 -> 256 first layer and a wide 256 -> 256 output layer, minus the
same chain without the plain layers (HIP events around the C-ABI launches, bf16, training stash).
    python tools/ls_vs_product.py [points = 4194304]"""
import json
import os
import re
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/tests']
import torch
import hypernerf_torch_amd as HN
from hypernerf_torch_amd import _lib as L
from hypernerf_torch_amd.hypernerf import modules

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
dev = "cuda"
HN.set_precision("bf16")
exe = os.path.join(ROOT, "tools", "ls_bench")
if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(exe + ".hip"):
    subprocess.run([L.hipcc_path(), "--offload-arch=gfx950", "-O3", "-o", exe, exe + ".hip"], check=True, timeout=600)


def product(depth):
    torch.manual_seed(0)
    m = modules.MLP(in_ch=32, out_ch=256, depth=depth, width=256, skips=[]).to(dev)
    arena = HN.ParamArena(m.parameters())
    x = torch.rand(P, 32, device=dev) * 2 - 1
    g = torch.randn(P, 256, device=dev) * 1e-3
    times = []
    for it in range(5):
        L.KERNEL_TIMES = {} if it >= 2 else None
        y = m(x)
        y.backward(g)
        arena.zero_grad()
        if it >= 2:
            t = L.collect_kernel_times()
            times.append({k.split("[")[0]: sum(v) for k, v in t.items()})
        L.KERNEL_TIMES = None
        del y
    torch.cuda.synchronize()
    med = {k: sorted(t[k] for t in times)[len(times) // 2] for k in times[0]}
    del m, arena, x, g
    torch.cuda.empty_cache()
    return med


full, base = product(9), product(1)
fwd = full["hn_mlp_forward"] - base["hn_mlp_forward"]
bwd = full["hn_mlp_backward"] - base["hn_mlp_backward"]
wg_key = next(k for k in full if "wgrad" in k and "reduce" not in k)
wg = full[wg_key] - base[wg_key]
flop = 2.0 * 256 * 256 * P * 8           # one product over the eight plain layers
out = subprocess.run([exe, str(P), "8", "6", "2", "32", "16", "0", "0"], capture_output=True, text=True, timeout=600).stdout
m = re.search(r"median ([0-9.]+) ms .*? = ([0-9.]+) of 2.5 PF", out)
ls_ms, ls_frac = float(m.group(1)), float(m.group(2))
res = {"points": P, "layers": "8 plain 256 -> 256 ReLU layers (the trunk's shape), bf16",
       "product_ms": {"forward": fwd, "backward_data": bwd, "weight_gradient": wg, "backward_pair": bwd + wg},
       "product_frac_of_2.5PF": {"forward": flop / (fwd * 1e-3) / 2.5e15, "backward_data": flop / (bwd * 1e-3) / 2.5e15,
                                 "weight_gradient": flop / (wg * 1e-3) / 2.5e15, "backward_pair": 2 * flop / ((bwd + wg) * 1e-3) / 2.5e15},
       "layer_stationary_pair_ms": ls_ms, "layer_stationary_pair_frac_of_2.5PF": ls_frac,
       "pair_speedup_of_the_layer_stationary_form": (bwd + wg) / ls_ms,
       "ls_bench_line": [ln for ln in out.splitlines() if ln.startswith("mode")][-1] if "mode" in out else out[-300:],
       "build": L.build_id()}
print(json.dumps(res))
