#!/usr/bin/env python3
"""Benchmark of the HyperNeRF render hot path on MI355X (BASELINE.json metric: ray-samples/s, forward + backward).

    python bench.py --gpus N --steps K --warmup W [--config {1,2,3,5}]
N>1: one rank per GPU over RCCL.  Under `torch.distributed.run` (WORLD_SIZE set) this process IS a rank; started
plainly, `--gpus N` makes it the LAUNCHER (the reference's Lightning Trainer spawns `devices=num_gpus` ranks itself,
train.py:224-229): it starts N child rank processes BEFORE any GPU call of its own, relays rank 0's JSON line and
exits with the children's worst return code.  Every run fails (rc != 0) unless the collective library itself counted
as many ranks as `--gpus` asked for.  `--launch-plan` prints the N commands + environments instead of running them.
A "step" = one training step of the reference's
hot loop on one synthetic ray batch already resident in HBM: model forward (coarse + fine), MSE loss, backward,
gradient all-reduce (N>1), Adam step.  Workloads (BASELINE.json `configs`):
    2 (default, the configuration the metric is quoted on)  NerfModel use_warp + bendy_sheet, 1024 rays x (64+64), bf16
    3  same model, 16384 rays x (64+128), bf16 (the "saturate the GPU" size)
    5  SE3Field warp + axis_aligned_plane (hyper_slice_out_dim = GLO_dim = 8) at config-2 shapes, bf16
    1  legacy nerf_pl render_rays, coarse only, 256 rays x 64 samples, fp32 (the reference's CPU-runnable case)
The K timed steps are repeated `--repeats` times back to back (each repeat bracketed by barrier + synchronize);
`value` comes from the MEDIAN repeat, `ms_per_step_repeats` lists all of them.  Prints ONE JSON line (rank 0).
With N>1 over RCCL the whole step — forward, backward, gradient all-reduce, Adam — is ONE HIP graph (the collective is
captured with it; `config.dp_step` says which form ran).  The default single-GPU config-2 run appends an `also` block to
its line: configs 3, 5, 1 and the eval image loop as short child runs (`--no-also` skips it), so that the one command
the driver runs measures every single-GPU configuration of BASELINE.json.
"""
import argparse
import ctypes
import json
import os
import statistics
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
# the shader clock a mid-pool MI355X holds under the config-2 step (hwmon): lines are re-priced at it so
# that a box's clock luck does not read as a code change
NOMINAL_SCLK_MHZ = 2250.0

CONFIGS = {
    1: dict(rays=256, nc=64, nf=0, precision="fp32", kind="legacy"),
    2: dict(rays=1024, nc=64, nf=64, precision="bf16", kind="hypernerf"),
    3: dict(rays=16384, nc=64, nf=128, precision="bf16", kind="hypernerf"),
    5: dict(rays=1024, nc=64, nf=64, precision="bf16", kind="se3"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--rays", type=int, default=None)
    ap.add_argument("--nc", type=int, default=None)
    ap.add_argument("--nf", type=int, default=None)
    ap.add_argument("--precision", default=None, choices=["bf16", "fp32", "bf16s8"],
                    help="bf16s8: OPT-IN variant of bf16 with the training stash in 8 bits (HN_MODE_BF16_S8) — never the default, "
                         "reported as dtype 'bf16 (8-bit stash: e4m3 X, e5m2 dZ)'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-calibration", action="store_true",
                    help="skip the box calibration (MFMA / HBM-stream probes before the timed region, hwmon power and "
                         "clock during it)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a HIP graph")
    ap.add_argument("--overlap", action="store_true",
                    help="data-parallel runs: all-reduce the gradient buffer in two buckets, the first in flight while the "
                         "weight gradients of the second are computed (dist.GradSync).  Off by default: splitting the "
                         "batched weight-gradient launch costs ~0.2 ms on one MI355X, more than the all-reduce it hides")
    ap.add_argument("--fork-wgrad", action="store_true",
                    help="fork each program's weight-gradient jobs onto a side stream behind its backward-data kernel "
                         "(parallel graph branches) instead of one batched launch at the end of backward.  Measured "
                         "slower on one MI355X (functional.py, profiles/r03_wgrad_fork_trace.txt); off by default")
    ap.add_argument("--launch-plan", action="store_true",
                    help="print the rank launch plan of --gpus N (commands + per-rank environment) as JSON and exit")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the `also` block (short lines for BASELINE configs 3, 5, 1 and the eval image loop that the "
                         "default config-2 run appends to its JSON line)")
    ap.add_argument("--no-capture-collective", action="store_true",
                    help="data-parallel runs: keep the gradient all-reduce OUT of the step graph (graph | eager all-reduce "
                         "| graph) instead of capturing it with the step")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the data-parallel code path (process group + all-reduce between two graphs) even with one rank")
    a = ap.parse_args()
    cfg = CONFIGS[a.config]
    for k in ("rays", "nc", "nf", "precision"):
        if getattr(a, k) is None:
            setattr(a, k, cfg[k])
    a.kind = cfg["kind"]
    return a


def macs_per_point(prog):
    return sum(ly.n_out * ly.in_features for ly in prog.layers)      # row-stacked layers count all their matrices


def build_workload(a, dev, rank):
    """(model-like callable returning (out, loss), parameters, {program name: (program, points per step)})."""
    import hypernerf_torch_amd as HN
    from hypernerf_torch_amd import functional as HF
    from hypernerf_torch_amd.hypernerf import model_utils
    from hypernerf_torch_amd.losses import MSELoss
    from gpu_common import EMB
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    b = a.rays
    o = torch.rand(b, 3, generator=g) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(b, 3, generator=g), dim=-1)
    ids = torch.randint(0, 100, (b, 1), generator=g).float()
    target = torch.rand(b, 3, generator=g).to(dev)
    loss_fn = MSELoss()
    torch.manual_seed(0)     # identical weights on every rank
    if a.kind == "legacy":
        from hypernerf_torch_amd.models import nerf as legacy_nerf
        from hypernerf_torch_amd.models.rendering import render_rays
        near, far = torch.full((b, 1), 2.0), torch.full((b, 1), 6.0)
        rays = torch.cat([o, d, near, far], dim=1).to(dev)
        coarse = legacy_nerf.NeRF().to(dev)
        emb = [legacy_nerf.Embedding(3, 10), legacy_nerf.Embedding(3, 4)]
        params = list(coarse.parameters())

        def fwd_bwd():
            res = render_rays([coarse], emb, rays, N_samples=a.nc, N_importance=0, perturb=1.0, noise_std=1.0)
            loss = ((res["rgb_coarse"] - target) ** 2).mean()     # losses.py:10 on the coarse level only
            HF.backward(loss)
            return {"fine": {"rgb": res["rgb_coarse"]}}, loss

        def programs():
            call = coarse._calls[next(iter(coarse._calls))] if hasattr(coarse, "_calls") else None
            return {} if call is None else {"NeRF": (call.program, b * a.nc)}
        workload = f"legacy render_rays, NeRF coarse only, {b} rays x {a.nc} samples, fwd+bwd+Adam"
        data = dict(o=o, d=d, near=near, far=far, target=target.cpu(), emb=emb)
        return fwd_bwd, params, programs, workload, coarse, data
    kw = dict(hyper_slice_method="bendy_sheet", use_warp=True, use_nerf_embed=True, use_alpha_cond=True)
    if a.kind == "se3":
        kw = dict(hyper_slice_method="axis_aligned_plane", hyper_slice_out_dim=8, use_warp=True, use_nerf_embed=True,
                  use_alpha_cond=True)
    from hypernerf_torch_amd.hypernerf.models import NerfModel
    model = NerfModel(EMB, near=0.0, far=1.0, n_samples_coarse=a.nc, n_samples_fine=a.nf, noise_std=1.0,
                      view_fourier_dim=6, **kw)
    if a.kind == "se3":
        from hypernerf_torch_amd.hypernerf import warping
        model.warp_field = warping.SE3Field(in_ch=3)
    model = model.to(dev)
    rays = torch.cat([o, d, torch.zeros(b, 1), torch.ones(b, 1), ids], dim=1).to(dev)
    extra = {'nerf_alpha': None, 'warp_alpha': None, 'hyper_alpha': None, 'hyper_sheet_alpha': None}

    def fwd_bwd():
        out = model(model_utils.prepare_ray_dict(rays), extra)
        loss = loss_fn(out, target)
        HF.backward(loss)           # accumulates into arena.grad, which the previous opt.step() left zeroed
        return out, loss

    def programs():
        return {name: (prog, pts) for name, prog, pts in model.compiled_programs(b)}

    def reference_programs():
        # the reference's own count: every fine sample through every network (models.py:752-768), no reuse
        return {name: (prog, pts) for name, prog, pts in model.compiled_programs(b, reference=True)}
    what = "SE3Field warp + axis_aligned_plane" if a.kind == "se3" else "use_warp bendy_sheet"
    workload = (f"NerfModel {what} nerf_embed+alpha_cond, {b} rays x ({a.nc}+{a.nf}) samples per GPU, fwd+bwd+Adam")
    data = dict(o=o, d=d, ids=ids, target=target.cpu(), model_kw=kw, extra=extra)
    programs.reference = reference_programs
    return fwd_bwd, list(model.parameters()), programs, workload, model, data


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_plan(n, argv, port=None, env=None):
    """The N rank processes of `bench.py --gpus N`: [(command, environment overrides)], rank 0 first.  Pure host
    logic (tests/test_host_api.py::test_bench_launch_plan): same script, same arguments, the torch.distributed
    environment contract per rank, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    env = dict(os.environ if env is None else env)
    port = int(env.get("MASTER_PORT") or port or _free_port())
    args = [x for x in argv if x != "--launch-plan"]
    plan = []
    for r in range(n):
        e = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
             "MASTER_ADDR": env.get("MASTER_ADDR", "127.0.0.1"), "MASTER_PORT": str(port)}
        # HSA_ENABLE_IPC_MODE_LEGACY is INHERITED as the caller has it (this image exports 0: dmabuf IPC).  The launcher
        # only writes it on request — HN_SET_IPC_ENV=<value> — or on its one retry after a rank died before its first
        # step (launch_ranks), where it flips the inherited setting and says so.
        if env.get("HN_SET_IPC_ENV") not in (None, ""):
            v = env["HN_SET_IPC_ENV"]
            e["HSA_ENABLE_IPC_MODE_LEGACY"] = "0" if v == "1" else v      # HN_SET_IPC_ENV=1: the pool's known-good value
        plan.append(([sys.executable, os.path.abspath(__file__)] + args, e))
    return plan


def _run_ranks(plan, ready_dir, drop_env=()):
    """Start the ranks of `plan`, relay nothing yet: returns (return codes, rank 0's stdout, ranks that got through their
    warm-up steps — i.e. process group, collectives and graph capture all worked)."""
    import subprocess
    import threading
    procs = []
    for r, (cmd, e) in enumerate(plan):
        env = {**os.environ, **e, "HN_READY_DIR": ready_dir}
        for k in drop_env:
            env.pop(k, None)
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None, text=True))
    # rank 0's stdout is drained by a thread; the main loop watches for a rank that died (the survivors would sit in
    # the rendezvous / a collective until its timeout) and ends exactly the processes it started
    buf = []
    rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            time.sleep(2.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    rcs = [p.wait() for p in procs]
    rd.join(timeout=10.0)
    ready = sum(os.path.exists(os.path.join(ready_dir, f"rank{r}")) for r in range(len(plan)))
    return rcs, (buf[0] if buf else ""), ready


def launch_ranks(a, argv):
    """Parent of a plain `bench.py --gpus N` (N > 1): no GPU call happens in this process.  Children are ordinary
    child processes (never an exec of a process that touched the GPU); rank 0's stdout is relayed, its JSON line
    checked against --gpus, and the exit code is the worst of the children's.  If a rank dies BEFORE every rank got
    through its warm-up steps (rendezvous, RCCL initialisation, first collectives), the launcher retries ONCE in fresh
    child processes with HSA_ENABLE_IPC_MODE_LEGACY flipped (set to 0 if the first attempt ran without it or with
    another value, removed if it ran with 0) and reports which setting worked."""
    import shutil
    import tempfile
    plan = launch_plan(a.gpus, argv)
    if a.launch_plan:
        print(json.dumps({"n_ranks": len(plan), "ranks": [{"cmd": c, "env": e} for c, e in plan]}))
        return 0
    first_ipc = plan[0][1].get("HSA_ENABLE_IPC_MODE_LEGACY", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
    attempts = []
    ready_dir = tempfile.mkdtemp(prefix="hn_bench_ranks_")
    try:
        rcs, out0, ready = _run_ranks(plan, ready_dir)
        attempts.append({"HSA_ENABLE_IPC_MODE_LEGACY": first_ipc if first_ipc is not None else "<unset>", "rcs": rcs,
                         "ranks_through_warmup": ready})
        if any(rcs) and ready < len(plan) and os.environ.get("HN_NO_IPC_RETRY", "0") != "1":
            shutil.rmtree(ready_dir, ignore_errors=True)
            os.makedirs(ready_dir, exist_ok=True)
            flipped = None if first_ipc == "0" else "0"
            plan2 = launch_plan(a.gpus, argv)         # a fresh rendezvous port
            for _, e in plan2:
                e.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
                if flipped is not None:
                    e["HSA_ENABLE_IPC_MODE_LEGACY"] = flipped
            print(f"bench.py launcher: rank return codes {rcs} with {ready} of {len(plan)} ranks through their warm-up; "
                  f"retrying once with HSA_ENABLE_IPC_MODE_LEGACY={'<unset>' if flipped is None else flipped} "
                  f"(was {'<unset>' if first_ipc is None else first_ipc})", file=sys.stderr)
            rcs, out0, ready = _run_ranks(plan2, ready_dir, drop_env=("HSA_ENABLE_IPC_MODE_LEGACY",) if flipped is None else ())
            attempts.append({"HSA_ENABLE_IPC_MODE_LEGACY": "<unset>" if flipped is None else flipped, "rcs": rcs,
                             "ranks_through_warmup": ready})
            plan = plan2
            if not any(rcs):
                print(f"bench.py launcher: the retry worked — HSA_ENABLE_IPC_MODE_LEGACY="
                      f"{'<unset>' if flipped is None else flipped} is the setting for this node", file=sys.stderr)
    finally:
        shutil.rmtree(ready_dir, ignore_errors=True)
    if any(rcs):
        print(f"bench.py launcher: rank return codes {rcs}; attempts: {json.dumps(attempts)} (HN_SET_IPC_ENV=<value> pins the "
              "variable for the ranks, HN_NO_IPC_RETRY=1 disables the retry), MASTER_ADDR=" + plan[0][1]["MASTER_ADDR"],
              file=sys.stderr)
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln)
    rc = max((abs(r) for r in rcs), default=0)
    if rc == 0 and line is None:
        print("bench.py launcher: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    if line is not None:
        res = json.loads(line)
        seen = res.get("config", {}).get("ranks_seen_by_collective")
        if res.get("n_gpus") != a.gpus or seen != a.gpus:
            print(f"bench.py launcher: asked for {a.gpus} ranks, result says n_gpus={res.get('n_gpus')}, "
                  f"collective saw {seen}", file=sys.stderr)
            rc = rc or 3
        res["config"]["launched_by"] = "bench.py (self-spawned ranks)"
        res["config"]["launch_attempts"] = attempts
        print(json.dumps(res), flush=True)
    return rc


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.launch_plan):
        sys.exit(launch_ranks(a, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        # the driver's contract: `--gpus N` under a launcher that started N ranks; anything else would report a
        # curve point for a job that is not the one asked for
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(4)
    # one process per GPU over RCCL ("nccl" on ROCm).  HN_DIST_BACKEND=gloo + more ranks than GPUs is a debugging
    # aid only: it runs the N>1 code path (graph capture, gradient all-reduce, Adam) on a single-GPU box.
    n_dev = max(1, torch.cuda.device_count())
    backend = os.environ.get("HN_DIST_BACKEND", "nccl")
    if local >= n_dev:
        if backend != "gloo":
            # two RCCL ranks on one device: RCCL refuses or hangs — never fold silently
            print(f"bench.py: LOCAL_RANK {local} but only {n_dev} visible GPU(s); more ranks than GPUs is a debugging "
                  "mode of the gloo backend only (HN_DIST_BACKEND=gloo)", file=sys.stderr)
            sys.exit(5)
        local = local % n_dev
    torch.cuda.set_device(local)
    # --force-dp: take the N>1 code path (process group, two graphs around the gradient all-reduce) with ONE rank: a
    # 1-GPU box can then exercise RCCL initialisation next to HIP graphs and price the split of the step graph
    dp = world > 1 or a.force_dp
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local)

    import hypernerf_torch_amd as HN
    from hypernerf_torch_amd import _lib as L
    from hypernerf_torch_amd.dist import all_gather_pixels

    HN.set_precision(a.precision)
    # the three machine kernels time themselves (HnMlpArgs.timeline): per-kernel durations of the TIMED graph replays,
    # not of a second, eager pass.  Enabled before the step is captured (the slots' addresses are baked into the graph).
    L.TIMELINE = {} if not a.no_roofline else None
    from hypernerf_torch_amd import functional as HF0
    if a.fork_wgrad:
        HF0.set_wgrad_overlap(True)
    wgrad_schedule = "forked: side stream / parallel graph branch" if HF0.WGRAD_OVERLAP else "serial: one batched launch"
    fwd_bwd, params, programs, workload, model, data = build_workload(a, dev, rank)
    reference_programs = getattr(programs, "reference", programs)
    use_graph = not a.no_graph
    # parameters and gradients live in one flat arena each: the kernels accumulate dW straight into it, Adam steps
    # one tensor, and data parallelism SUM-all-reduces the gradient buffer in place (the 1/N sits in the Adam kernel).
    arena = HN.ParamArena(params)
    if dp:
        dist.broadcast(arena.data, src=0)
    # the weights BEFORE any training step: what cpu_baseline.check renders with (the check measures the kernels'
    # arithmetic against the oracle's, not how far ~130 bf16 training steps have carried two trajectories apart)
    init_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()} if rank == 0 else None
    from hypernerf_torch_amd import optim as HO
    # one GPU: the launch that completes the gradient also applies Adam (hn_mlp_wgrad_reduce_adam); with an all-reduce
    # between backward and the optimizer the two launches stay
    opt = HN.ArenaAdam(arena, lr=5e-4, eps=1e-8, zero_grad=True, grad_scale=1.0 / world,
                       fuse_reduce=(not dp) and HO.FUSE_REDUCE)

    def whole_step():
        out, loss = fwd_bwd()
        opt.step()
        return out, loss

    sync = None
    if dp:
        # gradient all-reduce: ONE in-place SUM of the flat buffer by default; with --overlap in two buckets, template
        # networks first (in flight while the warp / sheet weight gradients are still being computed), the rest
        # behind them (hypernerf_torch_amd.dist.GradSync)
        from hypernerf_torch_amd import functional as HF
        from hypernerf_torch_amd.dist import GradSync
        sync = GradSync(arena, model, overlap=a.overlap)

        def fwd_bwd_dp():
            with sync.splitting():
                return fwd_bwd()

    dp_graph = None
    if use_graph:
        from hypernerf_torch_amd.graphs import GraphedStep
        from hypernerf_torch_amd import dist as HD
        step = None
        if not dp:
            step = GraphedStep(whole_step, warmup=3)
        elif sync.split is not None:
            dp_graph = "three pieces (two overlapped buckets)"
        elif a.no_capture_collective:
            dp_graph = "three pieces (--no-capture-collective)"
        elif not HD.collective_capturable():
            dp_graph = f"three pieces (backend {dist.get_backend()} stages through the host)"
        else:
            # forward + backward | in-place SUM all-reduce of the gradient arena (RCCL's kernel, enqueued on the
            # capturing stream) | Adam: ONE graph, one replay per step, no host between backward and the optimizer
            def whole_step_dp():
                out, loss = fwd_bwd_dp()
                HF.flush_held_wgrads()
                arena.all_reduce_sum(force=True)
                opt.step()
                return out, loss
            state = (arena.data, arena.grad, opt.exp_avg, opt.exp_avg_sq, opt.step_count)
            snap = [t.clone() for t in state]
            try:
                step = GraphedStep(whole_step_dp, warmup=3)
                dp_graph = "one graph: forward + backward + all-reduce + Adam"
            except Exception as e:      # noqa: BLE001 (whatever the runtime says about capturing the collective)
                dp_graph = f"three pieces (capturing the all-reduce raised {type(e).__name__}: {str(e)[:120]})"
                with torch.no_grad():
                    for dst, src in zip(state, snap):
                        dst.copy_(src)
                arena.bump()
        if step is None:
            # graphs around the collectives: forward+backward (+ first weight-gradient bucket) | all-reduce(bucket 0)
            # with the held weight-gradient bucket replayed next to it | all-reduce(bucket 1) | Adam
            HF.flush_held_wgrads()
            gfb = GraphedStep(fwd_bwd_dp, warmup=3, mutates_params=False,
                              warmup_fn=lambda: (fwd_bwd_dp(), HF.flush_held_wgrads()))
            gheld = (GraphedStep(HF.flush_held_wgrads, warmup=0, pool=gfb.graph.pool(), mutates_params=False)
                     if HF.held_wgrads() else None)
            arena.zero_grad()
            gopt = GraphedStep(opt.step, warmup=1)

            def step():
                res = gfb()
                sync.reduce(gheld, force=True)
                gopt()
                return res
    else:
        def step():
            out, loss = fwd_bwd_dp() if dp else fwd_bwd()
            if dp:
                sync.reduce(HF.flush_held_wgrads, force=True)
            opt.step()
            return out, loss

    def barrier():
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    if os.environ.get("HN_READY_DIR"):       # tells the launcher that rendezvous, collectives and capture all worked
        try:
            open(os.path.join(os.environ["HN_READY_DIR"], f"rank{rank}"), "w").close()
        except OSError:
            pass
    calibration = None
    if not a.no_calibration:
        from hypernerf_torch_amd import calibration as CAL
        if rank == 0:
            calibration = CAL.probes(dev)
        for _ in range(2):                  # back into the step's own thermal / clock state — on EVERY rank: with N > 1
            step()                          # a step holds collectives, the ranks must run the same number of them
    if L.TIMELINE is not None:
        L.timeline_reset()
    sampler = None
    if rank == 0 and not a.no_calibration:
        sampler = CAL.PowerSampler(dev).__enter__()
    reps = []
    for _ in range(max(1, a.repeats)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out, loss = step()
        barrier()
        dt = time.perf_counter() - t0
        if dp:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        reps.append(dt)
    timeline = L.timeline_read() if L.TIMELINE is not None else None       # the timed replays, nothing else
    if sampler is not None:
        sampler.__exit__(None, None, None)
        calibration["timed_region"] = sampler.summary()
        # hwmon's power figure is a running average over roughly a second: the timed region (steps x repeats, often
        # < 0.2 s) is too short for it to settle, so the same step is replayed for another second, untimed, and sampled
    if not a.no_calibration:
        # ... on every rank, the same number of steps (fixed from the max-over-ranks step time: collectives inside)
        n_soak = max(20, int(CALIB_SOAK_S / max(1e-6, statistics.median(reps) / a.steps)))
        soak = CAL.PowerSampler(dev).__enter__() if rank == 0 else None
        for _ in range(n_soak):
            step()
        barrier()
        if soak is not None:
            soak.__exit__(None, None, None)
            calibration["soak"] = dict(soak.summary(), seconds=CALIB_SOAK_S, steps=n_soak,
                                       what="the same step replayed back to back right after the timed region (untimed)")
    ranks_seen = world
    if dp:
        all_gather_pixels(out['fine']['rgb'].detach())     # eval-style pixel assembly works on this topology
        cnt = torch.ones(1, device=dev)
        dist.all_reduce(cnt)
        ranks_seen = int(cnt.item())                       # what the collective library itself reports
    dt = statistics.median(reps)
    b = a.rays
    samples_step = world * b * (a.nc + a.nf)
    value = samples_step * a.steps / dt

    res = {
        "metric": "ray-samples/sec (fwd+bwd+Adam), whole job; per-GPU = value/n_gpus",
        "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 (8-bit stash: e4m3 X, e5m2 dZ; opt-in)" if a.precision == "bf16s8" else a.precision, "data": "synthetic",
        "config": {"workload": workload, "baseline_config": a.config, "rays_per_gpu": b, "n_samples": a.nc,
                   "n_importance": a.nf, "parallelism": f"dp{world}", "hip_graph": use_graph, "wgrad_schedule": wgrad_schedule, "optimizer_launch": "fused with the gradient reduce (hn_mlp_wgrad_reduce_adam)" if opt.fuse_reduce else "hn_adam_step", "dp_code_path": dp, "dp_step": dp_graph, "grad_sync": None if sync is None else ("one all-reduce" if sync.split is None else f"two buckets split at float {sync.split} of {arena.numel}, overlapped"),
                   "ranks_seen_by_collective": ranks_seen},
        "per_gpu": value / world, "final_loss": float(loss.detach()),
        "repeats": len(reps), "ms_per_step_repeats": [1e3 * r / a.steps for r in reps],
        "ms_per_step_spread": [1e3 * min(reps) / a.steps, 1e3 * max(reps) / a.steps],
    }

    res["build"] = L.build_id()
    if calibration is not None:
        res["calibration"] = calibration
    if rank == 0 and not a.no_roofline:
        # per-kernel times of the machine kernels come from the timed region itself (kernel timeline); the eager pass
        # (HIP events around every C-ABI launch) adds the small kernels' list at N = 1 only — with N > 1 every rank
        # goes straight to the final barrier, no rank waits for another's rank-local work
        n_timed = a.steps * max(1, a.repeats)
        res.update(roofline(a, L, fwd_bwd, opt, programs(), reference_programs(), dt / a.steps, b, timeline, n_timed,
                            eager=(world == 1 and not dp), calibration=calibration))
        if world > 1 and "roofline" in res:
            res["roofline"]["scope"] = f"rank 0 (1 of {world} GPUs): its own kernels inside the timed region"
            if "hbm" in res:
                res["hbm"]["scope"] = res["roofline"]["scope"]
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(a, model, data, dev, init_state, arena)
    if (rank == 0 and world == 1 and not dp and not a.no_also and a.config == 2 and a.precision == "bf16"
            and a.rays == CONFIGS[2]["rays"] and use_graph):
        # the headline run is over; its ~10 GB stay allocated beside the children's (config 3: a 60 GB stash of 288).
        # The headline must survive whatever happens to the extras: if this process is told to stop while the children
        # run (a driver timeout sends SIGTERM), it prints the line it has and exits.
        import signal
        torch.cuda.synchronize()

        def _bail(signum, frame):
            res["also"] = {"skipped": f"interrupted by signal {signum} while the extra configurations ran"}
            sys.stdout.flush()
            print(json.dumps(res), flush=True)
            os._exit(0)
        old_handlers = {sg: signal.signal(sg, _bail) for sg in (signal.SIGTERM, signal.SIGINT)}
        try:
            res["also"] = also_block()
        finally:
            for sg, h in old_handlers.items():
                signal.signal(sg, h)
    if dp:
        dist.barrier()                      # every rank is done with its own work: tear the group down together
        dist.destroy_process_group()
    if ranks_seen != world or world != a.gpus:
        print(f"bench.py: the collective counted {ranks_seen} ranks, WORLD_SIZE={world}, --gpus {a.gpus}", file=sys.stderr)
        sys.exit(3)
    # the JSON line is the LAST thing on stdout: RCCL writes a "Librccl path" banner through C stdio, which would
    # otherwise be flushed at exit, after the line
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(res), flush=True)


CHECK_BOUND = {"fp32": 1e-4, "bf16": 1e-2, "bf16s8": 1e-2}      # cpu_baseline.check against the fp32 oracle, see its note
CHECK_BOUND_BF16_CONTRACT = 1e-3      # bf16 modes against the oracle under the same arithmetic contract (O.bf16_operands)
CALIB_SOAK_S = 1.2
ALSO_BUDGET_S = 60.0        # the whole block; a child that would start later is recorded as skipped


def also_block():
    """The driver runs ONE command (`python bench.py`, BASELINE config 2).  Four of the five BASELINE configurations and
    the evaluation loop would otherwise only ever be builder-run lines under profiles/: the default run therefore
    appends short measurements of configs 3, 5, 1 (<= 10 steps x 3 repeats, HIP-graph replay, no CPU baseline) and of
    `inference.render_image` (tools/eval_bench.py) — each its own child process, started after the headline has been
    measured (its ~10 GB of device memory stay allocated next to the children's; ordinary children, never an exec),
    each with value, ms/step, the dominant kernel's fractions and the kernel build it ran.  Bounded: ALSO_BUDGET_S for
    the block, 40 s per child; the parent prints the headline anyway if it is told to stop meanwhile."""
    import subprocess
    t0 = time.perf_counter()
    out = {"note": "short runs appended to the config-2 headline; same metric, same definitions (bench.py --config N); "
                   "not part of `value`"}
    jobs = [("config3", ["--config", "3", "--steps", "5", "--warmup", "2", "--repeats", "3"]),
            ("config5", ["--config", "5", "--steps", "10", "--warmup", "3", "--repeats", "3"]),
            ("config1", ["--config", "1", "--steps", "10", "--warmup", "3", "--repeats", "3"]),
            ("render_image", None)]
    for name, argv in jobs:
        left = ALSO_BUDGET_S - (time.perf_counter() - t0)
        if left < 8.0:
            out[name] = {"skipped": "time budget of the also block spent"}
            continue
        if argv is None:
            cmd = [sys.executable, os.path.join(ROOT, "tools", "eval_bench.py"), "32768", "3"]
        else:
            cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--no-cpu-baseline", "--no-also", "--no-calibration"]
        try:
            t1 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=min(40.0, left))
            line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
            if r.returncode != 0 or line is None:
                out[name] = {"error": f"rc {r.returncode}", "stderr_tail": r.stderr[-300:]}
                continue
            j = json.loads(line)
            if argv is None:
                out[name] = {"value": j["images_per_s"], "unit": "images/s (378 x 504, config-2 model, bf16)",
                             "ms_per_image": 1e3 * j["s_per_image"], "ray_samples_per_s": j["ray_samples_per_s"],
                             "kernel": "hn_mlp_fwd_kernel (inference build)", "frac": j["mfma_frac_of_2.5PF"],
                             "frac_of": "dense bf16 MFMA peak, forward GEMM FLOPs of the image / time of the WHOLE image loop",
                             "build": j.get("build"), "wall_s": time.perf_counter() - t1}
            else:
                rl = j.get("roofline", {})
                out[name] = {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "dtype": j["dtype"],
                             "steps": j["steps"], "repeats": j["repeats"], "ms_per_step_spread": j["ms_per_step_spread"],
                             "workload": j["config"]["workload"], "kernel": rl.get("kernel"), "bound": rl.get("bound"),
                             "frac": rl.get("frac"), "mfma_frac": rl.get("mfma_frac"),
                             "step_mfma_frac": j.get("step_mfma_frac"), "build": j.get("build"),
                             "wall_s": time.perf_counter() - t1}
        except subprocess.TimeoutExpired:
            out[name] = {"skipped": "child exceeded its time limit"}
        except Exception as e:      # a broken extra must never cost the headline line
            out[name] = {"error": repr(e)[:200]}
    out["wall_s"] = time.perf_counter() - t0
    return out


def roofline(a, L, fwd_bwd, opt, progs, ref_progs, step_s, b, timeline, n_timed, eager=True, calibration=None):
    """Per-kernel view of the step.

    Times: the three machine kernels time THEMSELVES inside the timed graph replays (kernel timeline, include/
    hn_kernels.h HnMlpArgs.timeline: last workgroup's end - first workgroup's start on the 100 MHz wall clock, summed
    over the `n_timed` timed steps) — so `machine_kernel_ms_per_step` belongs to the timed region and sums to less than
    `ms_per_step`; the rest of the step (13 small launches + dispatch gaps) is `other_ms_per_step`.  With `eager` a
    second, eager pass with HIP events around every C-ABI launch lists the small kernels too (`eager_pass`).

    FLOPs (2 x MACs of every Linear x evaluated points, the same for forward, backward-data and
    weight-gradient products): `executed` = what these kernels really computed (REUSE_COARSE evaluates warp field and
    hyper sheet once per coarse sample), `algorithmic` = the reference's own count for the same render.  Every MFMA
    fraction is EXECUTED FLOPs / time / dense peak — skipped work is not credited.

    `roofline` describes the dominant kernel by time, priced as SURVEY.md §8(d) defines it: `bound: "mfma"`, achieved =
    its executed GEMM FLOPs / its duration, peak = the dense MFMA peak of the operand dtype, `frac` = achieved / peak.
    hn_wgrad_kernel also streams the activation stash once — traffic of this design's own making: that view
    (`stash_stream`, `hbm_frac`, `frac_of_box_stream_probe`) is a diagnostic beside the roofline, never `frac`."""
    from hypernerf_torch_amd import functional as HF
    from hypernerf_torch_amd import machine as HM
    prec_key = "bf16" if a.precision == "bf16s8" else a.precision      # the 8-bit MFMA of the opt-in mode runs at the bf16 rate
    flops_pass = sum(2.0 * macs_per_point(p) * pts for p, pts in progs.values())      # one of fwd / bwd / wgrad, executed
    flops_pass_ref = sum(2.0 * macs_per_point(p) * pts for p, pts in ref_progs.values())
    mode = HF.mode_of(a.precision)
    tile_bytes = HM.mode_consts(mode)[1]
    stash_read = 0.0
    for prog, pts_list in _points_by_program(progs):
        for pts in pts_list:
            stash_read += prog.wgrad_stream_bytes(mode, pts)
    sym = {"forward": "hn_mlp_fwd_kernel", "backward": "hn_mlp_bwd_kernel", "wgrad": "hn_wgrad_kernel"}
    prefix = {"forward": "hn_mlp_forward", "backward": "hn_mlp_backward", "wgrad": "hn_mlp_wgrad"}
    kern, detail = {}, {}
    if timeline:
        for g, pre in prefix.items():
            hits = {k: v for k, v in timeline.items() if k.startswith(pre) and v["runs"] > 0}
            if hits:
                kern[g] = (sum(v["ms"] for v in hits.values()) / n_timed, sum(v["runs"] for v in hits.values()) / n_timed)
                detail.update({k: v["ms"] / n_timed for k, v in hits.items()})
    eager_times = None
    if eager or not kern:
        # per-kernel timing needs the kernels one after the other: concurrent launches (the forked weight-gradient
        # schedule of the timed region) share the chip and inflate each other's durations
        HF.set_wgrad_overlap(False)
        keep_tl, L.TIMELINE = L.TIMELINE, None
        L.KERNEL_TIMES = {}
        for _ in range(a.steps):
            fwd_bwd()           # rank-local: no collective here
            opt.step()
        times = L.collect_kernel_times()
        L.KERNEL_TIMES = None
        L.TIMELINE = keep_tl
        eager_times = {k: sum(v) / a.steps for k, v in times.items()}
        if not kern:            # no timeline (HN kernels built without it / --no-graph paths): fall back to the events
            for g, pre in prefix.items():
                ks = [k for k in times if k.split("[")[0].startswith(pre)]
                if ks:
                    kern[g] = (sum(eager_times[k] for k in ks), sum(len(times[k]) for k in ks) / a.steps)
    per_kernel = {}
    for g, (ms_step, launches) in kern.items():
        mf = flops_pass / (ms_step * 1e-3)
        per_kernel[sym[g]] = {"ms_per_step": ms_step, "launches_per_step": launches,
                              "mfma": {"achieved": mf / 1e12, "peak": PEAK[prec_key] / 1e12, "unit": "TFLOP/s",
                                       "frac": mf / PEAK[prec_key], "executed_flops_per_step": flops_pass}}
    out = {"executed_flops_per_step": 3.0 * flops_pass, "algorithmic_flops_per_step": 3.0 * flops_pass_ref,
           "flops_note": "3 x (2 x MACs of every Linear x points): forward, backward-data and weight-gradient products. "
                         "executed = the points these kernels evaluated; algorithmic = the reference's count for the same "
                         "render (every fine sample through warp field and hyper sheet again, models.py:752-768)"}
    if kern:
        dom = max(kern, key=lambda k: kern[k][0])
        ms_step, launches = kern[dom]
        pk = per_kernel[sym[dom]]["mfma"]
        traffic, traffic_note = _pmc_traffic(a)
        # rays 36 B + target 12 B + outputs 20 B per ray; weights read once forward and once backward
        # in the operand dtype; fp32 gradients written once
        uniq = {id(q): q for p, _ in progs.values() for q in p.params}     # a parameter shared by two programs counts once
        n_params = sum(q.numel() for q in uniq.values())
        alg_bytes = b * 68.0 + 2.0 * n_params * (2 if prec_key == "bf16" else 4) + 4.0 * n_params
        machine_ms = sum(v[0] for v in kern.values())
        rl = {"kernel": sym[dom] + ("<true>" if prec_key == "bf16" else "<false>"),
              "avg_launch_ms": ms_step / launches, "launches_per_step": launches,
              "times_from": "kernel timeline inside the timed graph replays" if timeline else "eager pass, HIP events",
              "timeline_active_in_timed_region": bool(timeline),
              "timeline_note": "the three machine kernels stamp the 100 MHz wall clock at entry and — behind a workgroup "
                               "barrier — at exit: two device-scope atomics per workgroup, INSIDE the timed region that "
                               "produces `value` (bench.py --no-roofline times the step without them; the independent check "
                               "is profiles/rNN_graph_trace_configC.txt, a rocprofv3 kernel trace of the same replay)",
              "mfma": {"achieved": pk["achieved"], "peak": pk["peak"], "unit": "TFLOP/s", "frac": pk["frac"],
                       "executed_flops_per_launch": flops_pass / launches},
              "traffic": None,
              "per_kernel": per_kernel, "machine_kernel_ms_per_step": dict(sorted(detail.items(), key=lambda kv: -kv[1])),
              "sum_machine_kernel_ms_per_step": machine_ms,
              "other_ms_per_step": 1e3 * step_s - machine_ms,
              "sum_kernel_ms_per_step": 1e3 * step_s,
              "sum_note": "machine kernels (timeline) + other_ms_per_step (the small launches and every dispatch gap of the "
                          "replayed graph) = ms_per_step by construction"}
        # the path is MFMA-bound — `frac` of the dominant kernel is ALWAYS its executed GEMM FLOPs / its
        # duration / the dense peak of the operand dtype.  The HBM view of hn_wgrad_kernel (it streams the activation stash
        # once: traffic this design inflicts on itself, not algorithmic bytes) sits beside it as a diagnostic.
        rl.update({"bound": "mfma", "achieved": pk["achieved"], "peak": pk["peak"], "unit": "TFLOP/s",
                   "frac": pk["frac"], "mfma_frac": pk["frac"],
                   "note": "dominant kernel by time; achieved = executed GEMM FLOPs of its launches in a step (SURVEY.md "
                           "§8d) / their duration inside the timed region; traffic = measured HBM bytes per launch "
                           "(profiles/, PMC passes)"})
        if calibration:
            rl["frac_of_box_mfma_probe"] = pk["achieved"] / calibration["mfma_probe_tflops"]
        if dom == "wgrad":
            gbps = stash_read / launches / (ms_step / launches * 1e-3) / 1e9
            rl["stash_stream"] = {"what": "diagnostic, NOT the roofline: the launch reads every stash tile (layer inputs X and "
                                          "layer gradients dZ of the step, written by the forward / backward machines) exactly "
                                          "once by LDS-DMA; these bytes are the design's own, ~500x the algorithmic bytes (`hbm`)",
                                  "achieved": gbps, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "hbm_frac": gbps / (HBM_PEAK / 1e9),
                                  "bytes_per_launch": stash_read / launches}
            rl["hbm_frac"] = gbps / (HBM_PEAK / 1e9)
            if calibration:
                rl["frac_of_box_stream_probe"] = gbps / 1e3 / calibration["hbm_probe_tbps"]
                rl["stash_stream"]["frac_of_box_stream_probe"] = rl["frac_of_box_stream_probe"]
        if eager_times is not None:
            rl["eager_pass"] = {"kernel_ms_per_step": {k: eager_times[k] for k in sorted(eager_times, key=eager_times.get, reverse=True)},
                                "sum_kernel_ms_per_step": sum(eager_times.values()),
                                "note": "second pass, launches issued eagerly with HIP events around each: lists the small "
                                        "kernels; runs ~2 % slower than the replayed graph (event records between launches)"}
        hbm = {"algorithmic_bytes_per_step": alg_bytes, "stash_bytes_read_per_step": stash_read,
               "stash_bytes_moved_per_step": 2.0 * stash_read, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
               "wgrad_stream_achieved": (stash_read / (kern["wgrad"][0] * 1e-3) / 1e9) if "wgrad" in kern else None,
               "traffic_per_step": None, "wasted_ratio": None}
        if traffic is not None:
            hbm["traffic_per_step"] = traffic["bytes_per_step"]
            hbm["wasted_ratio"] = traffic["bytes_per_step"] / alg_bytes
            hbm["traffic_source"] = traffic["source"]
            rl["traffic"] = traffic["per_kernel_launch"].get(sym[dom])
        else:
            rl["traffic_note"] = traffic_note
        out["roofline"] = rl
        out["hbm"] = hbm
    out["step_tflops"] = 3.0 * flops_pass / step_s / 1e12
    out["step_mfma_frac"] = 3.0 * flops_pass / step_s / PEAK[prec_key]
    out["step_mfma_frac_if_reference_flops_were_credited"] = 3.0 * flops_pass_ref / step_s / PEAK[prec_key]
    if calibration and kern:
        # Comparing lines from different boxes.  The two probes turned out NOT to separate the boxes of this pool (1885-1912
        # TFLOP/s and 6.75-7.05 TB/s on boxes whose step times differ by 7 %): both run in regimes of their own (the MFMA probe
        # at the power cap, ~1.9 GHz; the stream probe waits on HBM).  What does track the spread is the shader clock the
        # STEP ITSELF holds (hwmon during the soak): time x clock is constant to +-1.5 % over the boxes seen
        # (2322 MHz / 1.608 ms ... 2212 MHz / 1.693 ms).  Hence: the step re-priced at a nominal clock.
        soak = calibration.get("soak") or {}
        clk = soak.get("sclk_mhz") or (calibration.get("timed_region") or {}).get("sclk_mhz")
        if clk:
            norm = 1e3 * step_s * clk / NOMINAL_SCLK_MHZ
            calibration["normalised"] = {"ms_per_step_at_nominal_clock": norm, "nominal_sclk_mhz": NOMINAL_SCLK_MHZ,
                                         "step_sclk_mhz": clk,
                                         "value_at_nominal_clock": a.rays * (a.nc + a.nf) / (norm * 1e-3),
                                         "how": "ms_per_step x (shader clock the step held, hwmon mean over the soak) / nominal "
                                                "clock: the boxes of the pool differ in the clock they sustain under this "
                                                "step's load, and every kernel of the step scales with it to first order"}
    return out


def _points_by_program(progs):
    by = {}
    for name, (prog, pts) in progs.items():
        by.setdefault(id(prog), (prog, []))[1].append(pts)
    return list(by.values())


def _pmc_traffic(a):
    """HBM bytes per step measured with rocprofv3 PMC passes on this configuration (tools/collect_profiles.sh ->
    tools/make_profiles.py -> profiles/rNN_traffic_configC.json), newest round first.  A summary is only quoted when
    it was collected on THESE kernels (its `build.kernel_src_sha256` equals the running build's, or — comment-only edits of the
    sources — its `build.lib_sha256` equals the hash of the library this process loaded): a round that changes
    the stash must not report last round's bytes.  Returns (summary | None, note | None)."""
    import glob
    from hypernerf_torch_amd import _lib as L
    ident = L.build_id()
    mine = ident["kernel_src_sha256"]
    mine_lib = ident.get("lib_sha256")          # the built library itself: a comment-only source edit leaves it unchanged
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_traffic_config{a.config}.json")), reverse=True)
    stale = None
    for h in hits:
        try:
            t = json.load(open(h))
        except (OSError, ValueError):
            continue
        if not (t.get("rays") == a.rays and t.get("nc") == a.nc and t.get("nf") == a.nf
                and t.get("dtype", "bf16") == a.precision):      # the fp32 mode moves 3x the bytes of the bf16 mode
            continue
        theirs = (t.get("build") or {}).get("kernel_src_sha256")
        theirs_lib = (t.get("build") or {}).get("lib_sha256")
        if theirs == mine or (mine_lib is not None and theirs_lib == mine_lib):
            t["source"] = os.path.relpath(h, ROOT)
            return t, None
        stale = stale or (f"{os.path.relpath(h, ROOT)} was collected on kernel sources {theirs}, this build is {mine}: "
                          "not quoted")
    return None, stale or "no PMC summary for this configuration under profiles/"


def cpu_baseline(a, model, data, dev, init_state, arena):
    """The CPU oracle (oracle/hypernerf_oracle.py, validated against the reference's own outputs) timed on this
    node's host cores, fp32, forward + backward, on the SAME rays and targets as the GPU run, the weights the GPU run
    STARTED from (`init_state`, snapshotted before the first training step) and one seeded set of random draws: the FULL
    ray batch of the configuration when that fits the time budget (configs 1, 2, 5), else its first 2048 rays (config 3;
    per-ray-sample rate).  One 64-ray warm-up, then >= 3 timed iterations; `value` is the MEDIAN.  `check`: those
    weights are then put back into the GPU model, which renders the same rays with the same draws once — against the
    fp32 oracle (bound 1e-4 fp32 mode / 1e-2 bf16 mode) and, in the bf16 modes, against the oracle under the SAME
    arithmetic contract (`O.bf16_operands()`: every Linear rounds its matmul operands to bf16, accumulates fp32) at 1e-3."""
    from oracle import hypernerf_oracle as O
    from hypernerf_torch_amd import machine as HM
    from hypernerf_torch_amd.hypernerf import model_utils
    try:
        cores = len(os.sched_getaffinity(0))        # cores in this process's affinity mask ...
    except AttributeError:
        cores = os.cpu_count() or 1
    try:                                            # ... capped by the cgroup CPU quota (the pool's GPU boxes show 256
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]      # CPUs under a 16-CPU quota)
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cores = max(1, min(cores, 64))
    torch.set_num_threads(cores)
    b = min(a.rays, 2048)
    g = torch.Generator(device="cpu").manual_seed(4321)
    p = {k: v.clone().requires_grad_(True) for k, v in init_state.items()}
    gt = data["target"][:b]
    o, d = data["o"][:b], data["d"][:b]
    if a.kind == "legacy":
        rays = torch.cat([o, d, data["near"][:b], data["far"][:b]], dim=1)
        rng = {"perturb_rand": torch.rand(b, a.nc, generator=g), "noise_coarse": torch.randn(b, a.nc, generator=g)}

        def one(n, grad=True):
            out = O.legacy_render_rays([p], (10, 4), rays[:n], {k: v[:n] for k, v in rng.items()}, N_samples=a.nc,
                                       perturb=1.0, noise_std=1.0)
            if grad:
                ((out["rgb_coarse"] - gt[:n]) ** 2).mean().backward()
            return out["rgb_coarse"].detach()

        def gpu_once():
            from hypernerf_torch_amd.models.rendering import render_rays
            res = render_rays([model], data["emb"], rays.to(dev), N_samples=a.nc, N_importance=0, perturb=1.0,
                              noise_std=1.0, rng={k: v.to(dev) for k, v in rng.items()})
            return res["rgb_coarse"]
    else:
        idx = data["ids"][:b].reshape(-1).long()
        kw = dict(data["model_kw"])
        kw.pop("use_warp", None)
        cfg = O.ModelCfg(n_samples_coarse=a.nc, n_samples_fine=a.nf, noise_std=1.0, view_fourier_dim=6,
                         warp_kind="se3" if a.kind == "se3" else "translation", **kw)
        rng = {"t_rand": torch.rand(b, a.nc, generator=g), "noise_coarse": torch.randn(b, a.nc, 1, generator=g),
               "u": torch.rand(b, a.nf, generator=g), "noise_fine": torch.randn(b, a.nc + a.nf, 1, generator=g)}

        def one(n, grad=True):
            out = O.nerf_model_forward(p, cfg, o[:n], d[:n], idx[:n], {k: v[:n] for k, v in rng.items()})
            if grad:
                O.mse_loss(out, gt[:n]).backward()
            return out["fine"]["rgb"].detach()

        def gpu_once():
            rays = torch.cat([o, d, torch.zeros(b, 1), torch.ones(b, 1), data["ids"][:b]], dim=1).to(dev)
            out = model(model_utils.prepare_ray_dict(rays), data["extra"], rng={k: v.to(dev) for k, v in rng.items()})
            return out["fine"]["rgb"]

    def clear():
        for v in p.values():
            v.grad = None

    one(min(64, b))                 # warm-up (thread pool, allocator) on a small batch
    times, ref = [], None
    t_all = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_all < 12.0 and len(times) < 9):
        clear()
        t0 = time.perf_counter()
        ref = one(b)
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    with torch.no_grad():
        # the GPU model back on the weights the run started from (the arena's views: copied in place; the packed weight
        # streams are told to repack)
        model.load_state_dict({k: v.to(dev) for k, v in init_state.items()})
        arena.bump()
        HM.note_parameters_changed()
        got = gpu_once().float().cpu()
        delta = (got - ref).abs()
        diff, diff_mean = float(delta.max()), float(delta.mean())
        contract = None
        if a.precision != "fp32":
            with O.bf16_operands():
                ref16 = one(b, grad=False)
            d16 = (got - ref16).abs()
            contract = {"rgb_max_abs_diff": float(d16.max()), "rgb_mean_abs_diff": float(d16.mean()),
                        "bound": CHECK_BOUND_BF16_CONTRACT, "within_bound": bool(float(d16.max()) <= CHECK_BOUND_BF16_CONTRACT),
                        "what": "the same render against the oracle under the SAME arithmetic contract (O.bf16_operands(): every "
                                "Linear rounds its matmul operands to bf16 and accumulates in fp32; everything else fp32) — "
                                "what is left is summation order, the fast sine and samples whose pdf bin flips"}
    n_s = a.nc + a.nf
    ok = bool(diff <= CHECK_BOUND[a.precision]) and (contract is None or contract["within_bound"])
    return {"value": b * n_s / dt, "unit": "ray-samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{b} rays x {n_s} samples of the GPU run's own batch (same rays, targets; the weights the run started "
                      f"from{'' if b == a.rays else '; first 2048 rays'}), fp32, fwd+bwd (no optimizer), median of "
                      f"{len(times)} timed iterations after a 64-ray warm-up",
            "s_per_iteration": times,
            "check": {"gpu_vs_cpu_rgb_max_abs_diff": diff, "gpu_vs_cpu_rgb_mean_abs_diff": diff_mean,
                      "gpu_precision": a.precision, "bound": CHECK_BOUND[a.precision],
                      "within_bound": ok, "bf16_contract": contract,
                      "weights": "snapshot taken before the first training step, loaded back into the GPU model for this render",
                      "note": "the GPU model (in the run's precision mode) rendered the baseline's rays with the baseline's "
                              "weights and draws.  `bound`: 1e-4 in fp32 mode (the parity tests' element-wise bound), 1e-2 of "
                              "the [0,1] colour range in the bf16 modes against the fp32 oracle (tests/test_gpu_model.py holds "
                              "the bf16 forward to the same 1e-2) and, tighter, `bf16_contract` at 1e-3"}}


if __name__ == "__main__":
    main()
