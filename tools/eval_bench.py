"""Diagnostic: throughput of the evaluation path (SURVEY.md §8 f4 — eval.py's image loop): one 378 x 504 LLFF-sized
image rendered by `inference.render_image` with the BASELINE config-2 model (64+64 samples, bf16), chunked.
Prints one JSON line: images/s, rays/s, ray-samples/s, and the share of the dense bf16 MFMA peak of the forward
FLOPs (SURVEY.md §8d's per-point FLOP count / 3, the forward third).

usage: python tools/eval_bench.py [chunk] [reps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                      # noqa: E402
import hypernerf_torch_amd as HN                                  # noqa: E402
from hypernerf_torch_amd import inference                         # noqa: E402
from hypernerf_torch_amd.hypernerf import models                  # noqa: E402


def main():
    chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    h, w, nc, nf = 378, 504, 64, 64
    dev = torch.device("cuda", 0)
    HN.set_precision("bf16")
    emb = {k: list(range(100)) for k in ("warp", "appearance", "time")}
    emb["camera"] = [0]
    torch.manual_seed(0)
    m = models.NerfModel(emb, n_samples_coarse=nc, n_samples_fine=nf, hyper_slice_method="bendy_sheet",
                         use_nerf_embed=True, use_alpha_cond=True).to(dev)
    m.eval()
    m.use_stratified_sampling = False
    n = h * w
    g = torch.Generator().manual_seed(1)
    o = torch.rand(n, 3, generator=g) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    rays = torch.cat([o, d, torch.zeros(n, 1), torch.ones(n, 1), torch.full((n, 1), 7.0)], dim=1).to(dev)
    for _ in range(2):
        out = inference.render_image(m, rays, chunk=chunk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = inference.render_image(m, rays, chunk=chunk)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert torch.isfinite(out["rgb"]).all()
    res = {"what": "render_image, one 378x504 image, config-2 model, bf16, deterministic eval branch",
           "chunk": chunk, "s_per_image": dt, "images_per_s": 1.0 / dt, "rays_per_s": n / dt,
           "ray_samples_per_s": n * (nc + nf) / dt}
    # forward GEMM FLOPs of the image: 2 x MACs of every Linear x evaluated points (bench.py's definition)
    fl = sum(2.0 * bench.macs_per_point(prog) * (pts / min(chunk, n)) * n for _, prog, pts in m.compiled_programs(min(chunk, n)))
    res["forward_flops"] = fl
    res["forward_tflops"] = fl / dt / 1e12
    res["mfma_frac_of_2.5PF"] = fl / dt / 2.5e15
    from hypernerf_torch_amd import _lib
    res["build"] = _lib.build_id()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
