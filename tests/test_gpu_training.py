"""GPU tests (MI355X) of the training harness around the hot path (e): the fused Adam follows
a learning-rate schedule through HIP-graph replays, the first graphed step applies exactly one update, inference after
graph replays sees the updated weights, an out-of-range GLO index fails loudly, the HIP training loop tracks the CPU
oracle trained with torch.optim.Adam on the same batches and draws, and 2 data-parallel ranks (gloo, both on this one
GPU) reproduce the single-rank gradient and stay bit-identical to each other."""
import math
import datetime
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch

import hashprng as H
import hypernerf_torch_amd as HN
from gpu_common import oracle_threads, DEV, EMB, assert_close, load_hash, rays_for
from hypernerf_torch_amd import functional as F
from hypernerf_torch_amd.hypernerf import models
from hypernerf_torch_amd.dist import GradSync
from hypernerf_torch_amd.training import TrainStep
from oracle import hypernerf_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(hyper_slice_method="bendy_sheet", use_nerf_embed=True, use_alpha_cond=True, view_fourier_dim=6)


def ray_rows(seed, b):
    o, d, idx = rays_for(seed, b)
    rays = torch.cat([o, d, torch.zeros(b, 1), torch.ones(b, 1), idx.float()[:, None]], dim=1)
    return o, d, idx, rays


def small_model(seed, nc=16, nf=16, noise_std=None, precision="fp32"):
    HN.set_precision(precision)
    m = models.NerfModel(EMB, n_samples_coarse=nc, n_samples_fine=nf, noise_std=noise_std, **KW)
    sd = load_hash(m, seed)
    return m.to(DEV), sd


@pytest.fixture(autouse=True)
def restore_precision():
    old = HN.get_precision()
    yield
    HN.set_precision(old)


@pytest.mark.parametrize("use_graph", [False, True])
def test_lr_schedule_reaches_the_captured_adam_launch(use_graph):
    """lr lived in the kernel arguments and was frozen into the graph.  Now it is read from device memory:
    with lr = 0 a step must leave every parameter untouched, with lr restored it must move them again; MultiStepLR
    (utils/__init__.py:43-46) drives it through TrainStep.epoch_end()."""
    m, _ = small_model(31)
    _, _, _, rays = ray_rows(31, 64)
    rgbs = H.uniform(31, "rgbs", (64, 3), 0.1, 0.9).to(DEV)
    ts = TrainStep(m, lr=1e-3, use_graph=use_graph, decay_step=[1, 2], decay_gamma=0.5)
    ts.step(rays.to(DEV), rgbs)
    ts.step(rays.to(DEV), rgbs)
    p0 = ts.arena.data.clone()
    ts.optimizer.param_groups[0]["lr"] = 0.0
    log = ts.step(rays.to(DEV), rgbs)
    assert log["lr"] == 0.0
    assert torch.equal(ts.arena.data, p0), "lr = 0 must freeze the parameters (also through a graph replay)"
    ts.optimizer.param_groups[0]["lr"] = 1e-3
    ts.step(rays.to(DEV), rgbs)
    moved = (ts.arena.data - p0).abs().max().item()
    assert 0.0 < moved <= 1e-3 * 1.05 * 10, moved          # Adam's per-step move is bounded by ~lr (bias-corrected)
    ts.epoch_end()
    assert abs(ts.step(rays.to(DEV), rgbs)["lr"] - 5e-4) < 1e-12
    ts.epoch_end()
    assert abs(ts.step(rays.to(DEV), rgbs)["lr"] - 2.5e-4) < 1e-12
    assert float(ts.optimizer.hyper[0]) == pytest.approx(2.5e-4)


def test_first_graphed_step_applies_exactly_one_update():
    """the capture warm-ups used to be real steps (3 updates on the first batch).  Adam's first update
    moves every parameter by at most lr; after the first step() the device step counter reads 1 and the result
    equals an eager first step."""
    _, _, _, rays = ray_rows(32, 64)
    rgbs = H.uniform(32, "rgbs", (64, 3), 0.1, 0.9).to(DEV)
    res = {}
    for use_graph in (False, True):
        m, sd = small_model(32)
        m.use_stratified_sampling = False           # no random draws: eager and graph see the same samples
        ts = TrainStep(m, lr=1e-3, use_graph=use_graph)
        before = ts.arena.data.clone()
        ts.step(rays.to(DEV), rgbs)
        assert float(ts.optimizer.step_count) == 1.0
        delta = ts.arena.data - before
        assert float(delta.abs().max()) <= 1e-3 * 1.001
        res[use_graph] = delta
    # same update up to the summation order of the weight-gradient atomics (a sign flip needs |g| ~ 1e-7 |g|_typ)
    diff = (res[True] - res[False]).abs()
    assert float((diff > 1e-5).float().mean()) < 1e-3, float((diff > 1e-5).float().mean())


def test_inference_after_graph_replays_uses_the_updated_weights():
    """a replay steps the parameters without running Python; the packed weight streams of a following
    no_grad forward must be rebuilt — every time, not only after the first replay."""
    m, _ = small_model(33)
    m.use_stratified_sampling = False
    o, d, idx, rays = ray_rows(33, 64)
    rgbs = H.uniform(33, "rgbs", (64, 3), 0.1, 0.9).to(DEV)
    rd = {"origins": o.to(DEV), "directions": d.to(DEV), "viewdirs": None,
          "metadata": {k: idx.to(DEV) for k in ("warp", "camera", "appearance", "time")}}
    ts = TrainStep(m, lr=5e-3, use_graph=True)
    prev = None
    for rnd in range(3):
        ts.step(rays.to(DEV), rgbs)
        with torch.no_grad():
            got = m(rd, {})["fine"]["rgb"].clone()
        fresh = models.NerfModel(EMB, n_samples_coarse=16, n_samples_fine=16, noise_std=None, **KW).to(DEV)
        fresh.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
        fresh.use_stratified_sampling = False
        with torch.no_grad():
            ref = fresh(rd, {})["fine"]["rgb"]
        assert_close(got, ref, 1e-6, f"eval after replay {rnd}")
        if prev is not None:
            assert float((got - prev).abs().max()) > 1e-5, "training did not move the render"
        prev = got


def test_out_of_range_glo_index_poisons_the_output():
    """nn.Embedding raises a device assert on an out-of-range index (modules.py:155-167); the HIP gather must not
    silently train another row: the affected rows come out NaN, the gradient of the table stays finite."""
    tab = torch.nn.Parameter(H.uniform(9, "tab", (10, 8), -1, 1).to(DEV))
    idx = torch.tensor([0, 3, 10, -1, 9], device=DEV)
    out = F.embed_lookup(tab, idx)
    assert torch.isnan(out[2]).all() and torch.isnan(out[3]).all()
    assert torch.equal(out[[0, 1, 4]], tab.detach()[[0, 3, 9]])
    out[[0, 1, 4]].sum().backward()
    assert torch.isfinite(tab.grad).all() and float(tab.grad[[0, 3, 9]].min()) == 1.0


def test_training_tracks_the_cpu_oracle():
    """SURVEY.md §8 f1's pin: the HIP path (fp32 mode, TrainStep: chunk loop + fused Adam + graph replay) and the CPU
    oracle (autograd + torch.optim.Adam, the reference's optimizer) train 16 steps from the same weights on the same
    batches with the same random draws; loss curves agree to 2e-3 relative, PSNR to 0.02 dB — the measurable form of
    the north star's 'PSNR within 0.1 dB of the reference'.  (Two fp32 implementations of a training run drift apart
    exponentially — measured x2-3 per step at lr 5e-4 on this scene, from 4e-7 after the first update — so the run is
    kept short and the rate moderate; the per-step agreement is what pins the step function.)"""
    nc = nf = 16
    b, steps, seed = 96, 16, 41
    m, sd = small_model(seed, nc, nf, noise_std=0.5)
    cfg = O.ModelCfg(n_samples_coarse=nc, n_samples_fine=nf, noise_std=0.5, **KW)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    topt = torch.optim.Adam(list(p.values()), lr=1e-4, eps=1e-8)          # utils/__init__.py:29-31
    ts = TrainStep(m, lr=1e-4, use_graph=True, chunk=40)                   # 96 rays in chunks of 40, 40, 16
    torch.set_num_threads(oracle_threads(32))
    cpu_loss, hip_loss, cpu_psnr, hip_psnr = [], [], [], []
    for it in range(steps):
        o, d, idx, rays = ray_rows(seed + it % 3, b)                       # three batches, cycled
        gt = 0.5 + 0.5 * torch.sin(3.0 * o + 2.0 * d)                      # a smooth "scene"
        rng = {"t_rand": H.uniform(seed + it, "t", (b, nc), 0, 1), "u": H.uniform(seed + it, "u", (b, nf), 0, 1),
               "noise_coarse": H.normal(seed + it, "n1", (b, nc, 1)) * 0.5,
               "noise_fine": H.normal(seed + it, "n2", (b, nc + nf, 1)) * 0.5}
        topt.zero_grad()
        ref = O.nerf_model_forward(p, cfg, o, d, idx, rng)
        loss = O.mse_loss(ref, gt)
        loss.backward()
        topt.step()
        cpu_loss.append(float(loss.detach()))
        cpu_psnr.append(float(-10 * torch.log10(((ref["fine"]["rgb"].detach() - gt) ** 2).mean())))
        log = ts.step(rays.to(DEV), gt.to(DEV), rng={k: v.to(DEV) for k, v in rng.items()})
        hip_loss.append(float(log["train/loss"]))
        hip_psnr.append(float(log["train/psnr"]))
    cl, hl = np.array(cpu_loss), np.array(hip_loss)
    rel = np.abs(hl - cl) / cl
    print("loss rel err per step:", np.array2string(rel, precision=2))
    assert rel.max() <= 2e-3, (rel.max(), cl, hl)
    assert np.abs(np.array(hip_psnr) - np.array(cpu_psnr)).max() <= 0.02
    assert cl[-3:].mean() < cl[:3].mean(), "the oracle run itself must learn"
    print("cpu loss:", np.array2string(cl, precision=4))
    from gpu_common import _record
    _record("TrainStep vs CPU oracle + torch Adam: loss curve, 16 steps", "max rel", rel.max(), 2e-3)


def test_bf16_training_psnr_vs_cpu_oracle():
    """The north star's 'PSNR within 0.1 dB of reference' for the THROUGHPUT mode, tied to the reference's arithmetic
    (train.py:147-163, metrics.py:4-13) instead of to this repository's own fp32 mode: the bf16 `TrainStep` (HIP graph,
    fused Adam) and the CPU oracle + torch.optim.Adam train the same small scene — 400 steps x 128 rays x (16+16)
    samples, lr 1e-3 -> 1e-4 — from the same weights on the same batches and draws, 8 seeds; both final parameter sets
    are scored on held-out rays by the SAME evaluator (fp32 oracle, deterministic branch), so the two numbers differ by
    the training trajectory only.  Two fp32 implementations of such a run already end 0.4-0.5 dB apart seed by seed
    (chaotic trajectories), so with 8 seeds the standard error of the mean gap is ~0.15 dB: the test
    asserts what 8 seeds can carry — the mean gap is statistically compatible with the 0.1 dB bound (|mean| <= 0.1 +
    2 s.e.), no seed is off by more than 4 dB, both sides learned — and prints the numbers; the 64-seed run of the same
    harness (tests/psnr_vs_oracle.py, profiles/r04_psnr_vs_oracle_64seeds.json) is the tight statement."""
    import torch.multiprocessing as mp
    import oracle_train as OT
    steps, b, nc, nf, n_seeds, lr, lr_end, freq, noise = 400, 128, 16, 16, 8, 1e-3, 1e-4, 0.5, 0.5
    procs = max(1, min(n_seeds, OT.usable_cores() // 2))
    ctx = mp.get_context("spawn")
    pool = ctx.Pool(procs)
    try:
        fut = pool.map_async(OT.cpu_run, [(s_, steps, b, nc, nf, lr, noise, 2, lr_end, freq) for s_ in range(n_seeds)])
        gpu = [OT.gpu_run(s_, steps, b, nc, nf, lr, noise, "bf16", lr_end=lr_end, freq=freq) for s_ in range(n_seeds)]
        cpu = sorted(fut.get(timeout=900))
    finally:
        pool.terminate()
    ref = np.array([c[1] for c in cpu])
    got = np.array([g[0] for g in gpu])
    diffs = got - ref
    mean, se = float(diffs.mean()), float(diffs.std(ddof=1) / math.sqrt(n_seeds))
    print(f"bf16 TrainStep vs CPU oracle + torch Adam, held-out PSNR over {n_seeds} seeds: oracle {ref.mean():.2f} dB, "
          f"bf16 {got.mean():.2f} dB, gap {mean:+.3f} dB (s.e. {se:.3f}); per seed {np.array2string(diffs, precision=2)}")
    from gpu_common import _record
    _record("bf16 TrainStep vs CPU oracle: mean held-out PSNR gap (dB), 8 seeds", "abs", abs(mean), 0.1 + 2 * se)
    for c, g in zip(cpu, gpu):
        assert abs(c[2][0] - g[1][0]) <= 5e-2 * c[2][0], "the first step's loss: same weights, batch and draws"
        assert c[2][-1] < 0.5 * c[2][0] and g[1][-1] < 0.5 * g[1][0], "both runs must learn"
    assert ref.mean() > 20.0 and got.mean() > 20.0
    # per-seed gaps scatter with s.d. 0.78 dB (64-seed run): 8 seeds give s.e. 0.28 +- 0.07 and a largest gap of 1-2.5 dB;
    # the bounds leave 5 sigma to the run-to-run scatter (bf16 trajectories are not bit-reproducible: float atomics)
    assert se <= 0.6, f"standard error {se:.3f} dB: the harness lost its resolution"
    assert abs(mean) <= 0.1 + 2.0 * se, f"mean held-out PSNR gap {mean:+.3f} dB (s.e. {se:.3f}) vs the CPU oracle"
    assert float(np.abs(diffs).max()) <= 4.0, diffs


def test_eager_training_steps_do_not_leak_device_memory():
    """Eager (no HIP graph) training: device memory in use must not grow from step to step.  (The program autograd
    node once kept its own outputs alive through a reference cycle: ~10 MB per step at config 2, 288 GB after 27,000
    steps — found by a 5,000-iteration PSNR run.)"""
    HN.set_precision("bf16")
    m = models.NerfModel(EMB, n_samples_coarse=32, n_samples_fine=32, noise_std=1.0, **KW).to(DEV)
    arena = HN.ParamArena(m.parameters())
    opt = HN.ArenaAdam(arena, lr=1e-4)
    o, d, idx = rays_for(61, 256)
    rays = {"origins": o.to(DEV), "directions": d.to(DEV), "viewdirs": None,
            "metadata": {k: idx.to(DEV) for k in ("warp", "camera", "appearance", "time")}}
    gt = H.uniform(61, "gt", (256, 3), 0, 1).to(DEV)
    from hypernerf_torch_amd.losses import MSELoss
    lf = MSELoss()
    used = []
    for it in range(12):
        loss = lf(m(rays, {}), gt)
        loss.backward()
        opt.step()
        del loss
        torch.cuda.synchronize()
        used.append(torch.cuda.memory_allocated())
    assert max(used[4:]) == min(used[4:]), f"device memory in use changes from step to step: {used}"
    # a forward pass in training mode that is never back-propagated must be released as well
    for it in range(4):
        out = m(rays, {})
        del out
        torch.cuda.synchronize()
        used.append(torch.cuda.memory_allocated())
    assert used[-1] == used[-2] == used[4], used


def test_results_do_not_depend_on_stale_memory_or_timing():
    """tools/poison_probe.py as a test: the model is run on a clean allocator and again after the allocator has been
    poisoned (every later torch.empty() returns NaN- / 3.4e38- / 1000-filled memory); every output and every gradient
    must come out the same — forward tensors bit for bit, gradients to 1e-4 of their scale (float atomics) — anything
    else was computed from memory nobody wrote, or by a race
    (this is what found the counted-vmcnt race of the weight-gradient kernel's last, partial LDS stage: small jobs
    read LDS that had not landed yet, a handful of gradient tensors came out wrong every few runs)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import poison_probe
    # round 3: the LDS of every CU is poisoned too (tools/lds_poison.hip): staging planes / ring stages read before
    # they are written show up the same way as uninitialised global memory
    if not poison_probe.poison_lds(0x7fc00000):
        print("tools/liblds_poison.so missing and not buildable here: global-memory poisoning only")
    bad = poison_probe.probe(cases=("bendy_cond", "axis", "se3_axis"), arena_modes=(False, True),
                             sizes=((96, 32, 32), (100, 16, 24), (13, 7, 5)), verbose=False)
    assert not bad, bad[:8]
    # the opt-in 8-bit stash has its own 1-KiB tile layout and a 3-stage weight-gradient ring: same probe
    bad = poison_probe.probe(precisions=("bf16s8",), cases=("bendy_cond", "se3_axis"), arena_modes=(True,),
                             sizes=((96, 32, 32), (13, 7, 5)), verbose=False)
    assert not bad, bad[:8]


def test_train_step_recaptures_when_the_precision_mode_changes():
    """A captured step replays the kernels of the mode it was captured in; switching hypernerf_torch_amd.set_precision
    between steps must lead to a new capture, not to silent replays of the old mode."""
    HN.set_precision("bf16")
    try:
        m = models.NerfModel(EMB, n_samples_coarse=16, n_samples_fine=16, noise_std=None, **KW).to(DEV)
        m.use_stratified_sampling = False
        _, _, _, rays = ray_rows(71, 64)
        rgbs = H.uniform(71, "rgbs", (64, 3), 0.1, 0.9)
        ts = TrainStep(m, lr=0.0, use_graph=True)                 # lr 0: every step sees the same weights
        l_bf16 = float(ts.step(rays.to(DEV), rgbs.to(DEV))["train/loss"])
        g1 = ts._graph
        HN.set_precision("fp32")
        l_fp32 = float(ts.step(rays.to(DEV), rgbs.to(DEV))["train/loss"])
        assert ts._graph is not g1
        eager = TrainStep(m, lr=0.0, use_graph=False)
        l_ref = float(eager.step(rays.to(DEV), rgbs.to(DEV))["train/loss"])
        assert abs(l_fp32 - l_ref) <= 1e-6 * abs(l_ref), (l_fp32, l_ref)
        assert l_bf16 != l_fp32 and abs(l_bf16 - l_fp32) <= 2e-2 * abs(l_ref)
    finally:
        HN.set_precision("bf16")


# ------------------------------------------------------------------------------------------------------------------
# two data-parallel ranks on ONE GPU (gloo): the N>1 code path of TrainStep end to end
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, use_graph, overlap, q, backend="gloo"):
    for pth in (ROOT, os.path.join(ROOT, "tests")):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    global DEV
    if backend == "nccl":           # one rank per GPU over RCCL
        torch.cuda.set_device(rank)
        DEV = f"cuda:{rank}"
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    try:
        import hypernerf_torch_amd as HN2
        from hypernerf_torch_amd.dist import shard_rays
        from hypernerf_torch_amd.training import TrainStep as TS
        HN2.set_precision("fp32")
        m = models.NerfModel(EMB, n_samples_coarse=16, n_samples_fine=16, noise_std=None, **KW)
        load_hash(m, 50 + rank)                  # DIFFERENT weights per rank: TrainStep must broadcast rank 0's
        m = m.to(DEV)
        m.use_stratified_sampling = False
        _, _, _, rays = ray_rows(51, 64)
        rgbs = H.uniform(51, "rgbs", (64, 3), 0.1, 0.9)
        from hypernerf_torch_amd import functional as F2
        ts = TS(m, lr=1e-3, use_graph=use_graph, overlap_grad_sync=overlap)
        # two buckets: the template networks are the tail of the arena, the warp field / sheet / GLO table the head
        assert (ts.sync.split is not None) == overlap and (not overlap or 0 < ts.sync.split < ts.arena.numel)
        mine_r, mine_c = shard_rays(rays).to(DEV), shard_rays(rgbs).to(DEV)
        # gradient of the first step (before Adam consumes it): run the step body by hand
        ts._rays, ts._rgbs = mine_r.clone(), mine_c.clone()
        ts._forward_backward()
        assert (F2.held_wgrads() > 0) == overlap       # the warp / sheet weight gradients are still to be launched
        ts.sync.reduce(F2.flush_held_wgrads)
        assert F2.held_wgrads() == 0
        grad = (ts.arena.grad / world).cpu().clone()
        ts.arena.zero_grad()
        for _ in range(3):
            ts.step(mine_r, mine_c)
        q.put((rank, grad.numpy(), ts.arena.data.cpu().numpy().copy(), float(ts.optimizer.step_count), ts.dp_graph))
    finally:
        dist.destroy_process_group()


def _rccl_one_rank_worker(port, q):
    """One RCCL rank: TrainStep(force_dp=True) — broadcast, gradient all-reduce, Adam — as ONE captured graph, against
    the plain single-rank graph and against the three-piece path (capture_collective=False), same weights and batch."""
    for pth in (ROOT, os.path.join(ROOT, "tests")):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    import time
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
    try:
        import hypernerf_torch_amd as HN2
        from hypernerf_torch_amd.training import TrainStep as TS
        from hypernerf_torch_amd.graphs import GraphedStep as GS
        HN2.set_precision("bf16")
        b = 1024
        _, _, _, rays = ray_rows(61, b)
        rays, rgbs = rays.to(DEV), H.uniform(61, "rgbs", (b, 3), 0.1, 0.9).to(DEV)
        rng = {"t_rand": H.uniform(61, "t", (b, 64), 0, 1).to(DEV), "u": H.uniform(61, "u", (b, 64), 0, 1).to(DEV)}

        def make(**kw):
            m = models.NerfModel(EMB, n_samples_coarse=64, n_samples_fine=64, noise_std=None, **KW)
            load_hash(m, 60)
            return TS(m.to(DEV), lr=1e-3, use_graph=True, **kw)
        res = {}
        for name, kw in (("single", {}), ("one_graph", dict(force_dp=True)),
                         ("three_pieces", dict(force_dp=True, capture_collective=False))):
            ts = make(**kw)
            # gradient of the first step, by hand (what the all-reduce must leave untouched with one rank)
            ts._rays, ts._rgbs, ts._rng = rays.clone(), rgbs.clone(), {k: v.clone() for k, v in rng.items()}
            ts._forward_backward()
            if ts.dp:
                ts.sync.reduce(None, force=True)
            ts.optimizer.finish_gradients()     # one GPU: the reduce launch is held for the optimizer's fused launch (round 6)
            grad = ts.arena.grad.cpu().clone()
            ts.arena.zero_grad()
            start = ts.arena.data.cpu().numpy().copy()
            for _ in range(3):
                log = ts.step(rays, rgbs, rng=rng)
            torch.cuda.synchronize()
            data3 = ts.arena.data.cpu().numpy().copy()      # weights after 3 updates: all-reduce and Adam in order
            times = []
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(40):
                    ts.step(rays, rgbs, rng=rng)
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t0) / 40)
            res[name] = dict(grad=grad.numpy(), data3=data3, start=start, ms=1e3 * sorted(times)[2],
                             steps=float(ts.optimizer.step_count), dp_graph=ts.dp_graph,
                             graph_kind=type(ts._graph).__name__, loss=float(log["train/loss"]))
        q.put(res)
    finally:
        dist.destroy_process_group()


def _capture_beside_watchdog_worker(port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from hypernerf_torch_amd.graphs import GraphedStep, _capture_mode
        buf = torch.ones(1 << 20, device="cuda")
        acc = torch.zeros(1 << 20, device="cuda")

        def body():
            dist.all_reduce(buf)
            acc.add_(buf)
        n = 0
        for _ in range(8):
            # eager collectives right up to the capture: their works are still on the watchdog's list when it opens
            for _ in range(16):
                dist.all_reduce(buf)
            g = GraphedStep(body, warmup=1, mutates_params=False)
            for _ in range(3):
                g()
            n += 1
        torch.cuda.synchronize()
        q.put({"captures": n, "mode": _capture_mode(), "acc": float(acc[0])})
    finally:
        dist.destroy_process_group()


def test_graph_capture_survives_the_process_group_watchdog():
    """ProcessGroupNCCL's watchdog thread polls the events of eagerly enqueued collectives (hipEventQuery, every ~100 ms).
    Under the default "global" capture mode that query, made while the main thread has a capture open, fails with
    hipErrorStreamCaptureUnsupported, the watchdog rethrows and the whole rank aborts — one of three `bench.py --force-dp`
    runs died that way on the MI355X (round 4), i.e. an 8-rank job would hardly ever have started.  GraphedStep now
    captures in "thread_local" mode whenever a process group is alive and first lets the watchdog retire what was
    enqueued eagerly.  Eight captures, each directly behind 16 eager all-reduces, in one process."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p_ = ctx.Process(target=_capture_beside_watchdog_worker, args=(_free_port(), q))
    p_.start()
    try:
        res = q.get(timeout=300)
        p_.join(timeout=60)
        assert p_.exitcode == 0
    finally:
        if p_.is_alive():
            p_.kill()
    assert res["captures"] == 8 and res["mode"] == "thread_local"
    assert res["acc"] == 8 * (1 + 3)        # per capture: one eager warm-up run + three replays (the capture pass does not execute)


def test_one_rank_rccl_step_is_one_graph_with_the_all_reduce_inside():
    """N>1 readiness on one GPU: with the nccl (= RCCL) backend the data-parallel step — forward, backward, in-place SUM
    all-reduce of the gradient arena, Adam — is captured as ONE HIP graph (the collective's kernel is recorded like any
    launch): one replay per step, no host between backward and the optimizer.  Checked with a one-rank group (the only
    RCCL topology this box offers): same first gradient as the plain single-rank step (float atomics apart), the same
    WEIGHTS after three replayed updates as the single-rank graph and as the three-piece path (graph | eager all-reduce
    | eager Adam) — an Adam replayed before the reduce, or a reduce that missed part of the buffer, would show here —,
    and the replay times printed (bench.py --force-dp on one box: 1.916 against 1.877 ms = +2.1 %, the one-rank
    all-reduce being a 6 MB in-place copy kernel; only a gross regression, > 25 %, fails the test: the pool's boxes and
    their neighbours are too noisy for a tight bound)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p_ = ctx.Process(target=_rccl_one_rank_worker, args=(_free_port(), q))
    p_.start()
    try:
        res = q.get(timeout=420)
        p_.join(timeout=60)
        assert p_.exitcode == 0
    finally:
        if p_.is_alive():
            p_.kill()
    assert res["one_graph"]["graph_kind"] == "GraphedStep", res["one_graph"]["dp_graph"]
    assert res["one_graph"]["dp_graph"].startswith("one graph")
    assert res["three_pieces"]["graph_kind"] == "tuple" and "capture_collective=False" in res["three_pieces"]["dp_graph"]
    assert res["single"]["steps"] == res["one_graph"]["steps"] == res["three_pieces"]["steps"] == 203.0
    # float atomics order the weight-gradient sums differently from launch to launch: 1e-5 of the gradient's scale
    g0 = res["single"]["grad"]
    for k in ("one_graph", "three_pieces"):
        assert np.abs(res[k]["grad"] - g0).max() <= 1e-5 * np.abs(g0).max(), k
    # Adam's first steps are sign-like (lr * g / (|g| + eps)): elements whose gradient is smaller than the atomics'
    # rounding noise move by +-lr either way from run to run — the relative L2 of the three-step UPDATE is the measure
    # (the two-rank test below measures 1.4e-2 between two computations of the same updates)
    d0, upd = res["single"]["data3"], res["single"]["data3"] - res["single"]["start"]
    assert np.array_equal(res["single"]["start"], res["one_graph"]["start"])
    for k in ("one_graph", "three_pieces"):
        err = float(np.linalg.norm(res[k]["data3"] - d0) / np.linalg.norm(upd))
        print(f"{k}: weights after 3 replayed updates vs the single-rank graph, rel L2 of the update {err:.3e}")
        assert err <= 8e-2, (k, err)
    ratio = res["one_graph"]["ms"] / res["single"]["ms"]
    print(f"one-rank RCCL step as one graph: {res['one_graph']['ms']:.4f} ms, single-rank graph {res['single']['ms']:.4f} ms "
          f"(ratio {ratio:.4f}), three pieces {res['three_pieces']['ms']:.4f} ms")
    assert ratio <= 1.25, ratio


@pytest.mark.parametrize("use_graph,overlap", [(False, True), (True, True), (True, False)])
def test_train_step_two_ranks_match_one_rank(use_graph, overlap):
    """N>1 path of TrainStep end to end: broadcast of rank 0's weights, the gradient all-reduce in two overlapped
    buckets (dist.GradSync: the second bucket's weight gradients are launched while the first is being reduced) or
    as one, 1/world inside Adam — against one rank on the whole batch: first gradient and the weights after 3 steps."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, use_graph, overlap, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    got = {}
    try:
        for _ in range(2):
            rank, grad, data, steps, _form = q.get(timeout=240)
            got[rank] = (grad, data, steps)
        for p_ in procs:
            p_.join(timeout=60)
            assert p_.exitcode == 0
    finally:
        for p_ in procs:            # a rank that died leaves its peer inside a collective: do not wait for it
            if p_.is_alive():
                p_.kill()
    assert got[0][2] == got[1][2] == 3.0
    assert np.array_equal(got[0][0], got[1][0]), "the all-reduced gradient must be identical on both ranks"
    assert np.array_equal(got[0][1], got[1][1]), "replicas must stay bit-identical (broadcast at start + same updates)"
    # one rank on the whole batch, rank 0's weights
    HN.set_precision("fp32")
    m = models.NerfModel(EMB, n_samples_coarse=16, n_samples_fine=16, noise_std=None, **KW)
    load_hash(m, 50)
    m = m.to(DEV)
    m.use_stratified_sampling = False
    _, _, _, rays = ray_rows(51, 64)
    rgbs = H.uniform(51, "rgbs", (64, 3), 0.1, 0.9)
    ts = TrainStep(m, lr=1e-3, use_graph=False)
    assert ts.sync is None
    start = ts.arena.data.cpu().clone()
    ts._rays, ts._rgbs = rays.to(DEV), rgbs.to(DEV)
    ts._forward_backward()
    ts.optimizer.finish_gradients()     # one GPU: the reduce launch is held for the optimizer's fused launch (round 6)
    ref = ts.arena.grad.cpu()
    err = float((torch.from_numpy(got[0][0]) - ref).norm() / ref.norm())
    assert err <= 1e-5, f"mean of the two shard gradients vs the full-batch gradient: rel L2 {err:.2e}"
    # every bucket got reduced: the error of the head (warp / sheet / GLO) and of the tail (template) separately
    k = GradSync._template_offset(ts.arena, m)
    for name, sl in (("head", slice(0, k)), ("tail", slice(k, None))):
        e = float((torch.from_numpy(got[0][0])[sl] - ref[sl]).norm() / ref[sl].norm())
        assert e <= 1e-5, f"{name} bucket: rel L2 {e:.2e}"
    ts.arena.zero_grad()
    for _ in range(3):
        ts.step(rays.to(DEV), rgbs.to(DEV))
    moved = ts.arena.data.cpu() - start
    err = float((torch.from_numpy(got[0][1]) - ts.arena.data.cpu()).norm() / moved.norm())
    # Adam's first steps are sign-like (lr * g / (|g| + eps)): the ~1e-5 of elements whose gradient is smaller than the
    # rounding difference between the two computations move by +-lr either way; measured 1.4e-2 of the update
    assert err <= 5e-2, f"weights after 3 data-parallel steps vs 3 one-rank steps: rel L2 of the update {err:.2e}"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible GPUs (the builder's boxes have one)")
def test_two_rank_rccl_step():
    """The N > 1 path over RCCL itself, whenever the box shows two GPUs (the builder's pool never did: this test is
    written on one-GPU boxes and self-enables on the driver's node).  (1) `bench.py --gpus 2` launching its own ranks:
    the collective counts 2 ranks and the step is ONE graph with the all-reduce inside.  (2) TrainStep on two RCCL ranks,
    one per GPU, against one rank on the whole batch: replicas start from rank 0's weights, the all-reduced first
    gradient equals the full-batch gradient, replicas stay bit-identical, weights after 3 steps match."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                          "--repeats", "2", "--no-also", "--no-cpu-baseline"], capture_output=True, text=True, env=env,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["ranks_seen_by_collective"] == 2
    assert line["config"]["dp_step"].startswith("one graph"), line["config"]["dp_step"]
    assert line["value"] > 0 and line["config"]["launched_by"].startswith("bench.py")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, True, False, q, "nccl")) for r in range(2)]
    for p_ in procs:
        p_.start()
    got = {}
    try:
        for _ in range(2):
            rank, grad, data, steps, form = q.get(timeout=300)
            got[rank] = (grad, data, steps, form)
        for p_ in procs:
            p_.join(timeout=60)
            assert p_.exitcode == 0
    finally:
        for p_ in procs:
            if p_.is_alive():
                p_.kill()
    assert got[0][2] == got[1][2] == 3.0
    assert got[0][3].startswith("one graph") and got[1][3].startswith("one graph"), (got[0][3], got[1][3])
    assert np.array_equal(got[0][0], got[1][0]), "the all-reduced gradient must be identical on both ranks"
    assert np.array_equal(got[0][1], got[1][1]), "replicas must stay bit-identical"
    HN.set_precision("fp32")
    m = models.NerfModel(EMB, n_samples_coarse=16, n_samples_fine=16, noise_std=None, **KW)
    load_hash(m, 50)
    m = m.to(DEV)
    m.use_stratified_sampling = False
    _, _, _, rays = ray_rows(51, 64)
    rgbs = H.uniform(51, "rgbs", (64, 3), 0.1, 0.9)
    ts = TrainStep(m, lr=1e-3, use_graph=False)
    start = ts.arena.data.cpu().clone()
    ts._rays, ts._rgbs = rays.to(DEV), rgbs.to(DEV)
    ts._forward_backward()
    ts.optimizer.finish_gradients()     # one GPU: the reduce launch is held for the optimizer's fused launch (round 6)
    ref = ts.arena.grad.cpu()
    err = float((torch.from_numpy(got[0][0]) - ref).norm() / ref.norm())
    assert err <= 1e-5, f"mean of the two shard gradients vs the full-batch gradient: rel L2 {err:.2e}"
    ts.arena.zero_grad()
    for _ in range(3):
        ts.step(rays.to(DEV), rgbs.to(DEV))
    moved = ts.arena.data.cpu() - start
    err = float((torch.from_numpy(got[0][1]) - ts.arena.data.cpu()).norm() / moved.norm())
    assert err <= 5e-2, f"weights after 3 two-rank RCCL steps vs 3 one-rank steps: rel L2 of the update {err:.2e}"


def test_model_under_autocast_and_grad_scaler():
    """The reference trains with Lightning `precision=16` (opt.py:44): native AMP = torch.autocast(float16) around the
    forward pass + a GradScaler around backward / optimizer.  The HIP path keeps its own precision mode (autocast only
    re-types ATen ops): same loss as without autocast, finite unscaled gradients, the scaler's step goes through."""
    HN.set_precision("bf16")
    m = models.NerfModel(EMB, n_samples_coarse=16, n_samples_fine=16, noise_std=None, **KW).to(DEV)
    m.use_stratified_sampling = False
    _, _, _, rays = ray_rows(73, 64)
    rgbs = H.uniform(73, "rgbs", (64, 3), 0.1, 0.9).to(DEV)
    from hypernerf_torch_amd.hypernerf import model_utils as MU2
    from hypernerf_torch_amd.losses import MSELoss
    extra = dict(nerf_alpha=None, warp_alpha=None, hyper_alpha=None, hyper_sheet_alpha=None)
    rd = MU2.prepare_ray_dict(rays.to(DEV))
    plain = float(MSELoss()(m(rd, extra), rgbs).detach())
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    before = [p.detach().clone() for p in m.parameters()]
    with torch.autocast(device_type="cuda", dtype=torch.float16):
        out = m(rd, extra)
        loss = MSELoss()(out, rgbs)
    assert out["fine"]["rgb"].dtype == torch.float32
    assert abs(float(loss.detach()) - plain) <= 1e-6 * plain
    scaler.scale(loss).backward()
    scaler.unscale_(opt)
    grads = [p.grad for p in m.parameters() if p.grad is not None]
    assert grads and all(torch.isfinite(g).all() for g in grads)
    scaler.step(opt)
    scaler.update()
    assert scaler.get_scale() == 1024.0, "no inf/nan was found, the scale must not have been backed off"
    moved = sum(float((p.detach() - b).abs().sum()) for p, b in zip(m.parameters(), before))
    assert moved > 0.0


def _ddp_worker(rank, world, port, q):
    for pth in (ROOT, os.path.join(ROOT, "tests")):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    try:
        import hypernerf_torch_amd as HN2
        from hypernerf_torch_amd.dist import shard_rays
        from hypernerf_torch_amd.losses import MSELoss
        HN2.set_precision("fp32")
        m = models.NerfModel(EMB, n_samples_coarse=16, n_samples_fine=16, noise_std=None, **KW)
        load_hash(m, 50)
        m = m.to(DEV)
        m.use_stratified_sampling = False
        ddp = torch.nn.parallel.DistributedDataParallel(m, find_unused_parameters=True)   # nerf_embed is unused
        _, _, _, rays = ray_rows(51, 64)
        rgbs = H.uniform(51, "rgbs", (64, 3), 0.1, 0.9)
        mine_r, mine_c = shard_rays(rays).to(DEV), shard_rays(rgbs).to(DEV)
        from hypernerf_torch_amd.hypernerf import model_utils as MU2
        out = ddp(MU2.prepare_ray_dict(mine_r), dict(nerf_alpha=None, warp_alpha=None, hyper_alpha=None, hyper_sheet_alpha=None))
        MSELoss()(out, mine_c).backward()
        g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in m.parameters()])
        q.put((rank, g.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_model_under_torch_ddp_two_ranks():
    """The drop-in claim for the reference's own trainer (Lightning wraps the model in DistributedDataParallel,
    train.py:225-229): without a ParamArena the kernels return their gradients through autograd, so DDP's reducer
    averages them like any module's — 2 ranks (gloo, one GPU) against the full-batch gradient of one process."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    got = {}
    try:
        for _ in range(2):
            rank, g = q.get(timeout=240)
            got[rank] = g
        for p_ in procs:
            p_.join(timeout=60)
            assert p_.exitcode == 0
    finally:
        for p_ in procs:
            if p_.is_alive():
                p_.kill()
    assert np.array_equal(got[0], got[1]), "DDP leaves the same averaged gradient on both ranks"
    HN.set_precision("fp32")
    try:
        m = models.NerfModel(EMB, n_samples_coarse=16, n_samples_fine=16, noise_std=None, **KW)
        load_hash(m, 50)
        m = m.to(DEV)
        m.use_stratified_sampling = False
        _, _, _, rays = ray_rows(51, 64)
        rgbs = H.uniform(51, "rgbs", (64, 3), 0.1, 0.9).to(DEV)
        from hypernerf_torch_amd.hypernerf import model_utils as MU2
        from hypernerf_torch_amd.losses import MSELoss
        out = m(MU2.prepare_ray_dict(rays.to(DEV)), dict(nerf_alpha=None, warp_alpha=None, hyper_alpha=None, hyper_sheet_alpha=None))
        MSELoss()(out, rgbs).backward()
        ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in m.parameters()]).cpu()
    finally:
        HN.set_precision("bf16")
    err = float((torch.from_numpy(got[0]) - ref).norm() / ref.norm())
    assert err <= 1e-5, f"DDP-averaged shard gradients vs the full-batch gradient: rel L2 {err:.2e}"


def test_eval_image_loop_vs_oracle_deterministic(tmp_path):
    """the per-image loop of eval.py:145-178 — rays generated on the device, the fine level rendered
    in chunks with deterministic sampling (use_stratified_sampling=False: linspace depths and linspace u,
    model_utils.py:36-38, 226-227), the 8-bit frame and the PSNR — against the CPU oracle rendering the same image."""
    from hypernerf_torch_amd.inference import evaluate_images
    h, w, focal = 6, 8, 7.5
    m, sd = small_model(61, 16, 16, noise_std=None, precision="fp32")
    m = m.eval()
    m.use_stratified_sampling = False
    c2w = torch.tensor([[1.0, 0.0, 0.0, 0.1], [0.0, 1.0, 0.0, -0.2], [0.0, 0.0, 1.0, 1.5]])
    cfg = O.ModelCfg(n_samples_coarse=16, n_samples_fine=16, noise_std=None, **KW)
    samples, refs = [], []
    for img_id in (3, 7):
        rays = F.generate_rays(h, w, focal, c2w.to(DEV), near=0.0, far=1.0, ndc=False, image_id=img_id)
        ref_rays = O.image_rays(h, w, focal, c2w, 0.0, 1.0, False, image_id=img_id)
        assert_close(rays, ref_rays, 2e-6, "generated rays")
        o, d = ref_rays[:, 0:3], ref_rays[:, 3:6]
        idx = torch.full((h * w,), img_id, dtype=torch.int64)
        u = torch.linspace(0, 1, 16).expand(h * w, -1).contiguous()
        ref = O.nerf_model_forward({k: v.clone() for k, v in sd.items()}, cfg, o, d, idx, {"t_rand": None, "u": u})
        gt = (ref["fine"]["rgb"] + 0.02 * H.uniform(61 + img_id, "gt", (h * w, 3), -1, 1)).clamp(0, 1)
        refs.append((ref["fine"]["rgb"], ref["fine"]["depth"], gt))
        samples.append({"rays": rays, "rgbs": gt.to(DEV), "hw": (h, w)})
    res = evaluate_images(m, samples, chunk=20, save_dir=str(tmp_path))
    assert len(res["images"]) == 2 and res["images"][0].shape == (h, w, 3) and res["images"][0].dtype == torch.uint8
    for i, (rgb, depth, gt) in enumerate(refs):
        img8_ref = (rgb.view(h, w, 3) * 255).to(torch.uint8)
        diff = (res["images"][i].int() - img8_ref.int()).abs()
        assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 0.02      # truncation at a .0 boundary
        assert_close(res["depths"][i], depth.view(h, w), 1e-4, f"image {i} depth")
        psnr_ref = float(-10 * torch.log10(((gt - rgb) ** 2).mean()))
        assert abs(res["psnrs"][i] - psnr_ref) <= 1e-3, (res["psnrs"][i], psnr_ref)
        from hypernerf_torch_amd.inference import read_png
        back = read_png(os.path.join(str(tmp_path), f"{i:03d}.png"))           # eval.py:166 writes {i:03d}.png
        assert back.shape == (h, w, 3) and np.array_equal(back, res["images"][i].numpy())
    assert abs(res["mean_psnr"] - sum(res["psnrs"]) / 2) < 1e-12


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no outer launcher: the parent spawns the two ranks before any GPU call of its
    own (reference: Lightning spawns `devices=num_gpus` ranks, train.py:224-229), relays rank 0's line and fails unless
    the collective library counted 2 ranks.  On this one-GPU box the ranks share the device and talk gloo
    (HN_DIST_BACKEND) — the N>1 code path of bench.py itself: two graphs around the gradient all-reduce, 1/N in Adam."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HN_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                          "--repeats", "2", "--rays", "256", "--no-roofline", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=env, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["ranks_seen_by_collective"] == 2
    assert line["config"]["dp_code_path"] and line["config"]["parallelism"] == "dp2"
    assert line["config"]["launched_by"].startswith("bench.py")
    assert line["value"] > 0 and np.isfinite(line["final_loss"])


def test_checkpoint_into_the_hip_model_renders_like_the_oracle(tmp_path):
    """SURVEY.md §8 f3 on the GPU: a Lightning-layout checkpoint ({'state_dict': {'nerf.<name>': ...}}, train.py:48,
    200-204) holding hash weights -> utils.load_ckpt into an ARENA-backed NerfModel already on the device -> the HIP
    render equals the CPU oracle on those weights (1e-4) -> two TrainStep replays -> save_ckpt -> reload into a fresh
    model -> the same render, bit for bit, and the file holds the trained weights, not the loaded ones."""
    from hypernerf_torch_amd.utils import load_ckpt, save_ckpt
    nc = nf = 16
    b, seed = 48, 61
    HN.set_precision("fp32")
    donor = models.NerfModel(EMB, n_samples_coarse=nc, n_samples_fine=nf, noise_std=None, **KW)
    sd = load_hash(donor, seed)
    path = os.path.join(tmp_path, "epoch=1.ckpt")
    torch.save({"epoch": 1, "global_step": 10, "state_dict": {"nerf." + k: v.clone() for k, v in sd.items()}}, path)

    m = models.NerfModel(EMB, n_samples_coarse=nc, n_samples_fine=nf, noise_std=None, **KW).to(DEV)
    ts = TrainStep(m, lr=1e-3, use_graph=True)                      # parameters now live in ts.arena
    assert ts.arena.attached(m.warp_field.mlp.linears[0].weight) is not None
    load_ckpt(m, path, "nerf")
    assert ts.arena.attached(m.warp_field.mlp.linears[0].weight) is not None, "loading must go through the arena views"
    o, d, idx, rays = ray_rows(seed, b)
    rng = {"t_rand": H.uniform(seed, "t", (b, nc), 0, 1), "u": H.uniform(seed, "u", (b, nf), 0, 1)}
    cfg = O.ModelCfg(n_samples_coarse=nc, n_samples_fine=nf, noise_std=None, **KW)
    ref = O.nerf_model_forward({k: v.clone() for k, v in sd.items()}, cfg, o, d, idx, rng)
    from hypernerf_torch_amd.hypernerf import model_utils

    def render(model):
        with torch.no_grad():
            return model(model_utils.prepare_ray_dict(rays.to(DEV)), {}, rng={k: v.to(DEV) for k, v in rng.items()})
    out = render(m)
    for lvl in ("coarse", "fine"):
        for key in ("rgb", "depth", "acc", "weights"):
            assert_close(out[lvl][key], ref[lvl][key], 1e-4, f"render from a loaded checkpoint: {lvl}/{key}")
    gt = 0.5 + 0.5 * torch.sin(3.0 * o + 2.0 * d)
    for _ in range(2):
        ts.step(rays.to(DEV), gt.to(DEV), rng={k: v.to(DEV) for k, v in rng.items()})
    trained = render(m)
    assert float((trained["fine"]["rgb"] - out["fine"]["rgb"]).abs().max()) > 1e-5, "two Adam steps must move the render"
    p2 = save_ckpt(m, os.path.join(tmp_path, "epoch=2.ckpt"), model_name="nerf", epoch=2, global_step=12,
                   optimizer=ts.optimizer)
    blob = torch.load(p2, map_location="cpu")
    assert blob["epoch"] == 2 and float(blob["hn_optimizer"]["step"].reshape(-1)[0]) == 2.0
    moved = sum(float((blob["state_dict"]["nerf." + k] - v).abs().max()) > 0 for k, v in sd.items())
    assert moved >= 60, f"only {moved} tensors differ from the loaded weights after two steps"
    fresh = models.NerfModel(EMB, n_samples_coarse=nc, n_samples_fine=nf, noise_std=None, **KW).to(DEV)
    HN.ParamArena(fresh.parameters())
    load_ckpt(fresh, p2, "nerf")
    again = render(fresh)
    for lvl in ("coarse", "fine"):
        for key in ("rgb", "depth", "weights"):
            assert torch.equal(again[lvl][key], trained[lvl][key]), f"reloaded model renders differently: {lvl}/{key}"


@pytest.mark.parametrize("use_graph", [False, True])
def test_forked_weight_gradient_schedule_matches_serial(use_graph):
    """functional.set_wgrad_overlap(True) (bench.py --fork-wgrad): each program's weight-gradient launch on a side
    stream behind its own backward-data kernel, joined at the end of backward — eagerly and as parallel branches of a
    captured HIP graph.  Off by default (measured slower), but it must compute the same
    gradients as the one batched launch: to 1e-5 of the buffer's scale (float atomics in a different order)."""
    from hypernerf_torch_amd.graphs import GraphedStep
    from hypernerf_torch_amd.hypernerf import model_utils
    from hypernerf_torch_amd.losses import MSELoss
    m, _ = small_model(71, 16, 16, noise_std=None, precision="bf16")
    arena = HN.ParamArena(m.parameters())
    _, _, _, rays = ray_rows(71, 64)
    rays = rays.to(DEV)
    gt = H.uniform(71, "gt", (64, 3), 0, 1).to(DEV)
    rng = {"t_rand": H.uniform(71, "t", (64, 16), 0, 1).to(DEV), "u": H.uniform(71, "u", (64, 16), 0, 1).to(DEV)}
    loss_fn = MSELoss()

    def fwd_bwd():
        out = m(model_utils.prepare_ray_dict(rays), {}, rng=rng)
        F.backward(loss_fn(out, gt))

    grads = {}
    try:
        for forked in (False, True):
            F.set_wgrad_overlap(forked)
            arena.zero_grad()
            if use_graph:
                g = GraphedStep(fwd_bwd, warmup=2, mutates_params=False)
                arena.zero_grad()
                g()
            else:
                fwd_bwd()
            torch.cuda.synchronize()
            assert not F._FORKED and not F._PENDING
            grads[forked] = arena.grad.clone()
    finally:
        F.set_wgrad_overlap(False)
    scale = float(grads[False].abs().max())
    assert scale > 0
    assert float((grads[True] - grads[False]).abs().max()) <= 1e-5 * scale


@pytest.mark.parametrize("precision", ["bf16", "fp32", "bf16s8"])
@pytest.mark.parametrize("size", [(64, 16, 16), (300, 64, 64), (37, 24, 20)])
def test_weight_gradients_through_partial_slabs_are_reproducible_and_equal_the_atomic_flush(precision, size):
    """Round 5 (machine.WGRAD_PARTIALS, hn_mlp_wgrad_reduce): the jobs of the batched weight-gradient launch store their
    dW rectangles and bias sums as partial slabs and ONE reduce launch adds every gradient element once, in a fixed order.
    The GLO table's gradient goes the same way (per-block rows stored by the backward machine, summed per table row).
    (1) Against the rounds-1-4 flush (every job adds its rectangle, every block its table row, with float atomics): the
    same gradients to 1e-5 of the buffer's scale.  (2) Run to run: the WHOLE gradient buffer BIT-identical (the atomic
    flush differs in about half of the elements from run to run) — the opt-in 8-bit stash excepted, which keeps the
    atomics of its bias sums and of the table, and sample counts that are not multiples of 32 (blocks spanning rays),
    whose table gradient takes the two-launch hn_embed_backward path."""
    from hypernerf_torch_amd import machine
    from hypernerf_torch_amd.hypernerf import model_utils
    from hypernerf_torch_amd.losses import MSELoss
    b, nc, nf = size
    m, _ = small_model(87, nc, nf, noise_std=None, precision=precision)
    arena = HN.ParamArena(m.parameters())
    _, _, _, rays = ray_rows(87, b)
    rays = rays.to(DEV)
    gt = H.uniform(87, "gt", (b, 3), 0, 1).to(DEV)
    rng = {"t_rand": H.uniform(87, "t", (b, nc), 0, 1).to(DEV), "u": H.uniform(87, "u", (b, nf), 0, 1).to(DEV)}
    loss_fn = MSELoss()

    def grads():
        arena.zero_grad()
        out = m(model_utils.prepare_ray_dict(rays), {}, rng=rng)
        F.backward(loss_fn(out, gt))
        torch.cuda.synchronize()
        return arena.grad.clone()
    before = machine.WGRAD_PARTIALS
    try:
        machine.WGRAD_PARTIALS = 0
        atomic = grads()
        machine.WGRAD_PARTIALS = 1
        runs = [grads() for _ in range(3)]
    finally:
        machine.WGRAD_PARTIALS = before
    scale = float(atomic.abs().max())
    assert scale > 0 and float((runs[0] - atomic).abs().max()) <= 1e-5 * scale
    for name, p in m.named_parameters():
        off = (p.grad.data_ptr() - arena.grad.data_ptr()) // 4
        sl = slice(off, off + p.numel())
        ragged = nc % 32 != 0 or nf % 32 != 0      # blocks that span rays: the table's gradient takes hn_embed_backward (atomics)
        if (precision == "bf16s8" and (name.endswith("embed.weight") or name.endswith(".bias"))) or \
                (ragged and name.endswith("embed.weight")):
            assert float((runs[1][sl] - runs[0][sl]).abs().max()) <= 1e-5 * scale, name
        else:
            assert torch.equal(runs[1][sl], runs[0][sl]) and torch.equal(runs[2][sl], runs[0][sl]), name


def _fwd_bwd_in(precision, seed=83, b=64, nc=16, nf=16):
    """One forward + backward of the small model in `precision`: (outputs, gradient buffer, name -> (offset, numel))."""
    from hypernerf_torch_amd.hypernerf import model_utils
    from hypernerf_torch_amd.losses import MSELoss
    m, _ = small_model(seed, nc, nf, noise_std=None, precision=precision)
    arena = HN.ParamArena(m.parameters())
    _, _, _, rays = ray_rows(seed, b)
    gt = H.uniform(seed, "gt", (b, 3), 0, 1).to(DEV)
    rng = {"t_rand": H.uniform(seed, "t", (b, nc), 0, 1).to(DEV), "u": H.uniform(seed, "u", (b, nf), 0, 1).to(DEV)}
    out = m(model_utils.prepare_ray_dict(rays.to(DEV)), {}, rng=rng)
    arena.zero_grad()
    F.backward(MSELoss()(out, gt))
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    return {k: out["fine"][k].detach().clone() for k in ("rgb", "depth", "weights")}, arena.grad.clone(), grads


@pytest.mark.parametrize("b,nc,nf", [(64, 16, 16), (50, 24, 20)])       # the second: ragged last block and last tile
def test_eight_bit_stash_changes_nothing_but_the_weight_gradients(b, nc, nf):
    """precision 'bf16s8' (HN_MODE_BF16_S8, include/hn_kernels.h — OPT-IN, never what bench.py's headline runs): the
    training stash, which only the weight-gradient kernel reads, is kept as e4m3 (layer inputs) / e5m2 (2^16-scaled
    layer gradients).  The forward machine and the backward-data machine compute what they compute in 'bf16' mode:
    outputs bit-identical, embedding gradients (reduced inside the backward machine: they carry the power-of-two scale
    and lose it again, exactly) equal up to the order of their float atomics.  The weight gradients carry the 8-bit
    rounding of every (point, feature) term as (nearly) zero-mean noise: finite, within 15 % relative L2 of the bf16 mode's
    over the whole buffer at 64 rays (measured 7.6 %; 3.9 % at 1024 rays x 128 samples — tools/s8_check.py), and not
    identical (the mode must actually be in effect)."""
    out16, g16, by16 = _fwd_bwd_in("bf16", b=b, nc=nc, nf=nf)
    out8, g8, by8 = _fwd_bwd_in("bf16s8", b=b, nc=nc, nf=nf)
    for k in out16:
        assert torch.equal(out16[k], out8[k]), k
    assert bool(torch.isfinite(g8).all())
    emb = [n for n in by16 if "embed" in n]
    mlp = [n for n in by16 if "embed" not in n]
    assert emb and mlp
    for n in emb:
        scale = float(by16[n].abs().max())
        assert float((by8[n] - by16[n]).abs().max()) <= 1e-5 * scale + 1e-12, n
    a = torch.cat([by16[n].flatten() for n in mlp])
    b_ = torch.cat([by8[n].flatten() for n in mlp])
    rel = float((b_ - a).norm() / a.norm())
    assert 1e-4 < rel < 0.15, rel
    # no bias to speak of: the error's component ALONG the gradient — what would act like a change of the learning
    # rate — stays within a few per cent of it (measured -2.7 % / -0.5 % here, -0.4 % at 1024 rays x 128 samples)
    assert abs(float(((b_ - a) * a).sum() / (a * a).sum())) < 0.06


def test_eight_bit_stash_trains_and_stand_alone_modules_fall_back():
    """TrainStep in 'bf16s8' (captured graph): the loss falls as it does in 'bf16' (same data, same draws, 30 steps).  A stand-alone module with a wide output has no 8-bit build (MlpRunner.effective_mode): it runs in plain
    bf16 mode, gradients equal to the 'bf16' run's up to the order of the float atomics."""
    from hypernerf_torch_amd.hypernerf import modules
    losses = {}
    for prec in ("bf16", "bf16s8"):
        m, _ = small_model(91, 16, 16, noise_std=None, precision=prec)
        _, _, _, rays = ray_rows(91, 128)
        rgbs = H.uniform(91, "rgbs", (128, 3), 0.1, 0.9).to(DEV)
        ts = TrainStep(m, lr=2e-3, use_graph=True)
        ls = [float(ts.step(rays.to(DEV), rgbs)["train/loss"]) for _ in range(30)]
        assert all(math.isfinite(v) for v in ls)
        losses[prec] = ls
    assert losses["bf16s8"][-1] < 0.7 * losses["bf16s8"][0]
    # (trajectories at this learning rate are chaotic — the two modes end a factor 2 apart in either direction from
    # seed to seed; the comparison of distributions is tools/psnr_parity.py's job, profiles/r03_psnr_parity_s8.json)
    assert losses["bf16"][-1] < 0.7 * losses["bf16"][0]
    assert losses["bf16s8"][-1] < 4.0 * losses["bf16"][-1]
    grads = {}
    for prec in ("bf16", "bf16s8"):
        HN.set_precision(prec)
        mm = modules.MLP(in_ch=20, out_ch=17, depth=3, width=64)
        load_hash(mm, 92)
        mm = mm.to(DEV)
        x = H.uniform(92, "x", (100, 20), -1, 1).to(DEV)
        (mm(x) * H.uniform(92, "g", (100, 17), -1, 1).to(DEV)).sum().backward()
        grads[prec] = torch.cat([p.grad.flatten() for p in mm.parameters()])
    scale = float(grads["bf16"].abs().max())
    assert float((grads["bf16s8"] - grads["bf16"]).abs().max()) <= 1e-5 * scale


@pytest.mark.parametrize("precision,use_graph,chunk", [("fp32", False, 1 << 20), ("bf16", True, 1 << 20), ("bf16", False, 32)],
                         ids=["fp32-eager", "bf16-graph", "bf16-eager-two-chunks"])
def test_reduce_fused_with_adam_equals_reduce_then_adam(precision, use_graph, chunk):
    """Round 6: on one GPU the launch that completes the gradient also applies the optimizer (hn_mlp_wgrad_reduce_adam,
    optim.ArenaAdam(fuse_reduce=True): TrainStep's default without a process group).  Same sums in the same order, the
    same Adam arithmetic operation for operation: parameters, both moments, the zeroed gradient buffer and the step
    counter are BIT-identical to the two-launch form after every one of 4 steps — also with the batch in two chunks
    (the first chunk's reduce is completed by the plain launch, the last one's is fused).  `finish_gradients()` between
    backward and step() yields the complete gradient."""
    from hypernerf_torch_amd import optim, machine
    _, _, _, rays = ray_rows(61, 64)
    rgbs = H.uniform(61, "rgbs", (64, 3), 0.1, 0.9).to(DEV)
    state = {}
    for fuse in (False, True):
        old = optim.FUSE_REDUCE
        optim.FUSE_REDUCE = fuse
        try:
            m, _ = small_model(61, nc=32, nf=32, precision=precision)
            m.use_stratified_sampling = False           # no random draws: both runs see the same samples
            ts = TrainStep(m, lr=2e-3, use_graph=use_graph, chunk=chunk)
            assert ts.optimizer.fuse_reduce == fuse
            names = []
            L_launch = HN._lib.launch

            def spy(name, *a, **k):
                names.append(name)
                return L_launch(name, *a, **k)
            if not use_graph:
                HN._lib.launch = spy
                for mod in (machine, optim):
                    mod.L.launch = spy
            snaps = []
            try:
                for _ in range(4):
                    ts.step(rays.to(DEV), rgbs)
                    o = ts.optimizer
                    snaps.append([t.clone() for t in (ts.arena.data, o.exp_avg, o.exp_avg_sq, ts.arena.grad, o.step_count)])
            finally:
                HN._lib.launch = L_launch
                for mod in (machine, optim):
                    mod.L.launch = L_launch
            state[fuse] = snaps
            if not use_graph:
                n_chunks = -(-64 // chunk)
                if fuse:
                    assert names.count("hn_mlp_wgrad_reduce_adam") == 4 and names.count("hn_adam_step") == 0
                    assert names.count("hn_mlp_wgrad_reduce") == 4 * (n_chunks - 1)
                else:
                    assert names.count("hn_mlp_wgrad_reduce_adam") == 0 and names.count("hn_adam_step") == 4
        finally:
            optim.FUSE_REDUCE = old
    for k, (a, b) in enumerate(zip(state[False], state[True])):
        for what, x, y in zip(("parameters", "exp_avg", "exp_avg_sq", "gradient buffer", "step counter"), a, b):
            assert torch.equal(x, y), f"step {k}: {what} differ between reduce -> Adam and the fused launch"
    assert float(state[True][-1][4]) == 4.0 and not bool(state[True][-1][3].any())
    # the complete gradient on request: backward, finish_gradients() == the two-launch form's gradient
    grads = {}
    for fuse in (False, True):
        m, _ = small_model(61, nc=32, nf=32, precision=precision)
        m.use_stratified_sampling = False
        arena = HN.ParamArena(m.parameters())
        opt = HN.ArenaAdam(arena, lr=1e-3, fuse_reduce=fuse)
        from hypernerf_torch_amd.hypernerf import model_utils
        from hypernerf_torch_amd.losses import MSELoss
        out = m(model_utils.prepare_ray_dict(rays.to(DEV)), {})
        F.backward(MSELoss()(out, rgbs))
        opt.finish_gradients()
        grads[fuse] = arena.grad.clone()
        opt.step()
    assert torch.equal(grads[False], grads[True]) and bool(grads[True].any())
