#!/usr/bin/env python3
"""Benchmark of the HyperNeRF render hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
(N>1: launched by torch.distributed.run, one rank per GPU, RCCL).  A "step" = one training step of the
reference's hot loop on one synthetic ray batch already resident in HBM: NerfModel forward (coarse + fine),
MSE loss, backward, gradient all-reduce (N>1), Adam step.  Workload at N=1 = BASELINE configs[1]:
use_warp + bendy_sheet, 1024 rays x (64+64) samples, bf16 MFMA operands / fp32 accumulate.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

PEAK = {"bf16": 2.5e15, "fp32": 157.3e12}   # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8.0e12                             # HBM3E bytes/s, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--nc", type=int, default=64)
    ap.add_argument("--nf", type=int, default=64)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a HIP graph")
    return ap.parse_args()


def macs_per_point(prog):
    return sum(ly.weight.shape[0] * ly.weight.shape[1] for ly in prog.layers)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU over RCCL ("nccl" on ROCm).  HN_DIST_BACKEND=gloo + more ranks than GPUs is a debugging
    # aid only: it runs the N>1 code path (graph capture, gradient all-reduce, eager Adam) on a single-GPU box.
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("HN_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    dev = torch.device("cuda", local)

    import hypernerf_torch_amd as HN
    from hypernerf_torch_amd import _lib as L
    from hypernerf_torch_amd.dist import GradBucket, all_gather_pixels
    from hypernerf_torch_amd.hypernerf import model_utils
    from hypernerf_torch_amd.hypernerf.models import NerfModel
    from hypernerf_torch_amd.losses import MSELoss, psnr
    from gpu_common import EMB

    HN.set_precision(a.precision)
    torch.manual_seed(0)     # identical weights on every rank
    model = NerfModel(EMB, near=0.0, far=1.0, n_samples_coarse=a.nc, n_samples_fine=a.nf, noise_std=1.0,
                      hyper_slice_method="bendy_sheet", use_warp=True, use_nerf_embed=True, use_alpha_cond=True,
                      view_fourier_dim=6).to(dev)
    use_graph = not a.no_graph
    # parameters and gradients live in one flat arena each: the kernels accumulate dW straight into it, Adam
    # steps one tensor, and data parallelism all-reduces the gradient buffer in place.
    # The optimizer step is part of the captured graph on one GPU; with N>1 the gradient all-reduce sits between
    # the captured forward+backward and a second captured graph holding the fused Adam step
    arena = HN.ParamArena(model.parameters())
    # HIP fused Adam over the arena: one launch updates all parameters and clears the gradient buffer for the next step
    opt = HN.ArenaAdam(arena, lr=5e-4, eps=1e-8, zero_grad=True)
    bucket = arena
    loss_fn = MSELoss()

    # synthetic rays (B,9): origins U(-1,1)^3, unit-ish directions, near/far 0/1, image id; rgb targets
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    b = a.rays
    o = torch.rand(b, 3, generator=g) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(b, 3, generator=g), dim=-1)
    ids = torch.randint(0, 100, (b, 1), generator=g).float()
    rays = torch.cat([o, d, torch.zeros(b, 1), torch.ones(b, 1), ids], dim=1).to(dev)
    target = torch.rand(b, 3, generator=g).to(dev)
    extra = {'nerf_alpha': None, 'warp_alpha': None, 'hyper_alpha': None, 'hyper_sheet_alpha': None}

    def fwd_bwd():
        rd = model_utils.prepare_ray_dict(rays)
        out = model(rd, extra)
        loss = loss_fn(out, target)
        loss.backward()             # accumulates into arena.grad, which the previous opt.step() left zeroed
        return out, loss

    def eager_step():
        out, loss = fwd_bwd()
        if world > 1:
            bucket.all_reduce_mean()
        opt.step()
        return out, loss

    def whole_step():
        out, loss = fwd_bwd()
        opt.step()
        return out, loss

    if use_graph:
        from hypernerf_torch_amd.graphs import GraphedStep
        if world == 1:
            step = GraphedStep(whole_step, warmup=3)
        else:
            # two graphs around the one collective: forward+backward | all-reduce of the gradient buffer | Adam
            gfb = GraphedStep(fwd_bwd, warmup=3)
            gopt = GraphedStep(opt.step, warmup=1)

            def step():
                res = gfb()
                bucket.all_reduce_mean()
                gopt()
                return res
    else:
        step = eager_step

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out, loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        all_gather_pixels(out['fine']['rgb'].detach())     # eval-style pixel assembly works on this topology
    samples = world * b * (a.nc + a.nf) * a.steps
    value = samples / dt

    res = {
        "metric": "ray-samples/sec (fwd+bwd+Adam), whole job; per-GPU = value/n_gpus",
        "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.precision, "data": "synthetic",
        "config": {"workload": f"NerfModel use_warp bendy_sheet nerf_embed+alpha_cond, {b} rays x ({a.nc}+{a.nf}) "
                               f"samples per GPU, fwd+bwd+Adam", "rays_per_gpu": b, "n_samples": a.nc,
                   "n_importance": a.nf, "parallelism": f"dp{world}", "hip_graph": use_graph},
        "per_gpu": value / world, "final_loss": float(loss.detach()),
    }

    if rank == 0 and not a.no_roofline:
        # second pass: the same steps with HIP events around every C-ABI launch (on the launch stream)
        L.KERNEL_TIMES = {}
        for _ in range(a.steps):
            fwd_bwd()           # rank-local: no collective here, the other ranks are not in this pass
            opt.step()
        times = L.collect_kernel_times()
        L.KERNEL_TIMES = None
        tot = {k: sum(v) for k, v in times.items()}
        # group launches by KERNEL (= rocprofv3's per-symbol rows): forward / backward-data / weight-gradient machine
        progs = {}
        for lvl, pts in (("coarse", b * a.nc), ("fine", b * (a.nc + a.nf))):
            call = [c for k, c in model._template_calls.items() if k[0] == lvl][0]
            progs[f"template_{lvl}"] = (macs_per_point(call.program), pts)
        warp_prog = model.warp_field._calls[next(iter(model.warp_field._calls))].program
        sheet_prog = model.hyper_sheet_mlp._calls[next(iter(model.hyper_sheet_mlp._calls))].program
        all_pts = b * (2 * a.nc + a.nf)
        progs["TranslationField"] = (macs_per_point(warp_prog), all_pts)
        progs["HyperSheetMLP"] = (macs_per_point(sheet_prog), all_pts)
        flops_per_step = sum(2.0 * m * p for m, p in progs.values())        # per machine kernel and step
        # the weight-gradient kernel streams every stash tile once: its algorithmic bytes are what its job list reads
        mode = 1 if a.precision == "bf16" else 0
        tile_bytes = 2048 if a.precision == "bf16" else 4096
        prog_of = {"template_coarse": [c for k, c in model._template_calls.items() if k[0] == "coarse"][0].program,
                   "template_fine": [c for k, c in model._template_calls.items() if k[0] == "fine"][0].program}
        wgrad_bytes = 0.0
        for prog, pts_list in ((prog_of["template_coarse"], [b * a.nc]), (prog_of["template_fine"], [b * (a.nc + a.nf)]),
                               (warp_prog, [b * a.nc, b * (a.nc + a.nf)]), (sheet_prog, [b * a.nc, b * (a.nc + a.nf)])):
            for pts in pts_list:
                j = prog.wgrad_jobs(mode, pts)
                wgrad_bytes += float(((j["n_nt"] + j["n_kt"]).astype("int64") * (j["blk1"] - j["blk0"])).sum()) * tile_bytes
        kern = {}
        for sym in ("hn_mlp_forward", "hn_mlp_backward", "hn_mlp_wgrad"):
            ks = [k for k in tot if k.startswith(sym + "[") or k.startswith(sym + "_batched[")]
            kern[sym] = (sum(tot[k] for k in ks) / a.steps, sum(len(times[k]) for k in ks) / a.steps)
        names = {"hn_mlp_forward": "hn_mlp_fwd_kernel", "hn_mlp_backward": "hn_mlp_bwd_kernel",
                 "hn_mlp_wgrad": "hn_wgrad_kernel"}
        per_kernel = {}
        for sym, (ms_step, launches) in kern.items():
            mf = flops_per_step / (ms_step * 1e-3)
            e = {"ms_per_step": ms_step, "launches_per_step": launches,
                 "mfma": {"achieved": mf / 1e12, "peak": PEAK[a.precision] / 1e12, "unit": "TFLOP/s",
                          "frac": mf / PEAK[a.precision], "algorithmic_flops_per_step": flops_per_step}}
            if sym == "hn_mlp_wgrad":
                bw = wgrad_bytes / (ms_step * 1e-3)
                e["hbm"] = {"achieved": bw / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": bw / HBM_PEAK,
                            "algorithmic_bytes_per_step": wgrad_bytes}
            per_kernel[names[sym]] = e
        dom = max(kern, key=lambda k: kern[k][0])
        ms_step, launches = kern[dom]
        pk = per_kernel[names[dom]]
        bound = "hbm" if "hbm" in pk else "mfma"       # the weight-gradient kernel is a pure stream; the others GEMM
        rl = dict(pk[bound])
        per_launch = (rl.pop("algorithmic_bytes_per_step", None) or rl.pop("algorithmic_flops_per_step")) / launches
        res["roofline"] = {"bound": bound, "kernel": names[dom] + ("<true>" if a.precision == "bf16" else "<false>"),
                           "achieved": rl["achieved"], "peak": rl["peak"], "unit": rl["unit"], "frac": rl["frac"],
                           "traffic": None, "avg_launch_ms": ms_step / launches, "launches_per_step": launches,
                           ("algorithmic_bytes_per_launch" if bound == "hbm" else "algorithmic_flops_per_launch"): per_launch,
                           "note": "dominant kernel by time.  achieved = algorithmic bytes (stash tiles the job list "
                                   "reads, each once) or GEMM FLOPs of all its launches in a step / their summed "
                                   "HIP-event duration on the launch stream; measured HBM traffic per launch: "
                                   "profiles/r01_pmc_per_kernel.csv",
                           "per_kernel": per_kernel,
                           "kernel_ms_per_step": {k: tot[k] / a.steps for k in sorted(tot, key=tot.get, reverse=True)},
                           "sum_kernel_ms_per_step": sum(tot.values()) / a.steps}
        res["step_tflops"] = 3.0 * flops_per_step / (dt / a.steps) / 1e12

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(a)

    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(a):
    """The CPU oracle (oracle/hypernerf_oracle.py, validated against the reference's own outputs) timed on
    this node's host cores on a bounded sample of the same workload, fp32."""
    import hashprng as H
    from gpu_common import EMB, rays_for
    from hypernerf_torch_amd.hypernerf.models import NerfModel
    from oracle import hypernerf_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))        # cores this process may actually use (cgroup-aware)
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    torch.set_num_threads(cores)
    b = 64
    kw = dict(hyper_slice_method="bendy_sheet", use_nerf_embed=True, use_alpha_cond=True)
    m = NerfModel(EMB, n_samples_coarse=a.nc, n_samples_fine=a.nf, noise_std=1.0, view_fourier_dim=6, **kw)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    cfg = O.ModelCfg(n_samples_coarse=a.nc, n_samples_fine=a.nf, noise_std=1.0, view_fourier_dim=6, **kw)
    o, d, idx = rays_for(1, b)
    gt = H.uniform(1, "gt", (b, 3), 0, 1)

    def one():
        rng = {"t_rand": torch.rand(b, a.nc), "noise_coarse": torch.randn(b, a.nc, 1),
               "u": torch.rand(b, a.nf), "noise_fine": torch.randn(b, a.nc + a.nf, 1)}
        for v in p.values():
            v.grad = None
        out = O.nerf_model_forward(p, cfg, o, d, idx, rng)
        O.mse_loss(out, gt).backward()

    one()
    t0 = time.perf_counter()
    n = 0
    while n < 3 or (time.perf_counter() - t0 < 10.0 and n < 50):
        one()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": b * (a.nc + a.nf) / dt, "unit": "ray-samples/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{b} rays x ({a.nc}+{a.nf}) samples, fp32, fwd+bwd (no optimizer), "
                                      f"{n} timed iterations after 1 warm-up"}


if __name__ == "__main__":
    main()
