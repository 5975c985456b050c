"""Test infrastructure: one small training run of the CPU oracle (oracle/hypernerf_oracle.py + torch.optim.Adam — the
reference's training step, train.py:147-163 with utils.get_optimizer's Adam, utils/__init__.py:29-31) next to the same
run on the HIP path, for the bf16-training-vs-reference PSNR statement (tests/test_gpu_training.py, tests/psnr_vs_oracle.py).

Everything a run needs is derived from (seed, sizes): initial weights (hash-filled state dict), the analytic scene,
the ray batch of every step and its random draws (a seeded torch.Generator on the CPU) — so the CPU worker processes and
the GPU process reproduce the same data without exchanging tensors.  The held-out PSNR of BOTH final parameter sets is
computed by the same evaluator (the fp32 oracle, deterministic branch): what differs between the two numbers is the
training trajectory only.  CPU-only: safe to run in spawned worker processes next to a process that holds the GPU."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

KW = dict(hyper_slice_method="bendy_sheet", use_nerf_embed=True, use_alpha_cond=True)
EMB = {"warp": list(range(100)), "camera": [0], "appearance": list(range(100)), "time": list(range(100))}


def usable_cores() -> int:
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box of the pool shows
    256 CPUs in its mask under a 16-CPU quota: a thread pool sized by the mask is heavily over-subscribed)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


SCENE_FREQ = 1.0        # spatial frequency scale of the analytic scene (lower = easier)


def scene(n, gen, n_img=8, freq=None):
    """n rays of a smooth analytic dynamic scene: colour depends on origin, direction and (through the image id) time."""
    f = SCENE_FREQ if freq is None else freq
    o = torch.rand(n, 3, generator=gen) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    idx = torch.randint(0, n_img, (n,), generator=gen)
    t = idx.float()[:, None] / n_img
    col = 0.5 + 0.5 * torch.sin(f * (2.0 * o + 1.5 * d) + 2 * math.pi * t * torch.tensor([1.0, 0.5, 0.25]))
    return o, d, idx, col


def lr_at(it, steps, lr, lr_end):
    """Exponential decay from lr to lr_end over the run (lr_end None: constant)."""
    return lr if not lr_end else lr * (lr_end / lr) ** (it / max(1, steps - 1))


def batches(seed, steps, b, nc, nf, noise_std, freq=None):
    """The run's data: [(o, d, idx, gt, rng)] per step + the held-out set, from one generator."""
    g = torch.Generator().manual_seed(1000 + seed)
    pool = scene(4096, g, freq=freq)
    held = scene(512, g, freq=freq)
    out = []
    for _ in range(steps):
        sel = torch.randint(0, 4096, (b,), generator=g)
        rng = {"t_rand": torch.rand(b, nc, generator=g), "u": torch.rand(b, nf, generator=g),
               "noise_coarse": torch.randn(b, nc, 1, generator=g) * noise_std,
               "noise_fine": torch.randn(b, nc + nf, 1, generator=g) * noise_std}
        out.append(tuple(x[sel] for x in pool) + (rng,))
    return out, held


def initial_state(seed, nc, nf):
    import hashprng as H
    from hypernerf_torch_amd.hypernerf import models
    m = models.NerfModel(EMB, n_samples_coarse=nc, n_samples_fine=nf, noise_std=None, **KW)
    sd = m.state_dict()
    return H.fill_state_dict({k: tuple(v.shape) for k, v in sd.items()}, 500 + seed)


def heldout_psnr(sd, held, nc, nf):
    """fp32 oracle, deterministic branch (mid-bin coarse samples, evenly spaced u, no noise), fine render."""
    from oracle import hypernerf_oracle as O
    o, d, idx, col = held
    cfg = O.ModelCfg(n_samples_coarse=nc, n_samples_fine=nf, noise_std=None, **KW)
    n = o.shape[0]
    rng = {"t_rand": torch.full((n, nc), 0.5), "u": torch.linspace(0, 1, nf + 2)[1:-1].expand(n, nf).contiguous()}
    with torch.no_grad():
        p = {k: v.detach().clone().float() for k, v in sd.items()}
        out = O.nerf_model_forward(p, cfg, o, d, idx, rng)
        mse = ((out["fine"]["rgb"] - col) ** 2).mean()
    return float(-10.0 * torch.log10(mse))


def cpu_run(args):
    """One oracle training run.  args = (seed, steps, b, nc, nf, lr, noise_std, threads[, lr_end, freq]) ->
    (seed, held-out PSNR, loss curve)."""
    seed, steps, b, nc, nf, lr, noise_std, threads = args[:8]
    lr_end = args[8] if len(args) > 8 else None
    freq = args[9] if len(args) > 9 else None
    torch.set_num_threads(max(1, int(threads)))
    from oracle import hypernerf_oracle as O
    data, held = batches(seed, steps, b, nc, nf, noise_std, freq)
    sd = initial_state(seed, nc, nf)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(p.values()), lr=lr, eps=1e-8)
    cfg = O.ModelCfg(n_samples_coarse=nc, n_samples_fine=nf, noise_std=noise_std, **KW)
    losses = []
    for it, (o, d, idx, gt, rng) in enumerate(data):
        opt.param_groups[0]["lr"] = lr_at(it, steps, lr, lr_end)
        opt.zero_grad()
        loss = O.mse_loss(O.nerf_model_forward(p, cfg, o, d, idx, rng), gt)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return seed, heldout_psnr({k: v.detach() for k, v in p.items()}, held, nc, nf), losses


def gpu_run(seed, steps, b, nc, nf, lr, noise_std, precision, dev="cuda:0", use_graph=True, lr_end=None, freq=None):
    """The same run through TrainStep on the HIP path in `precision` -> (held-out PSNR by the fp32 oracle, loss curve)."""
    import hypernerf_torch_amd as HN
    from hypernerf_torch_amd.hypernerf import models
    from hypernerf_torch_amd.training import TrainStep
    HN.set_precision(precision)
    data, held = batches(seed, steps, b, nc, nf, noise_std, freq)
    m = models.NerfModel(EMB, n_samples_coarse=nc, n_samples_fine=nf, noise_std=noise_std, **KW)
    m.load_state_dict(initial_state(seed, nc, nf))
    m = m.to(dev)
    ts = TrainStep(m, lr=lr, eps=1e-8, use_graph=use_graph)
    losses = []
    for it, (o, d, idx, gt, rng) in enumerate(data):
        ts.optimizer.param_groups[0]["lr"] = lr_at(it, steps, lr, lr_end)      # uploaded by step() (sync_hyper)
        rays = torch.cat([o, d, torch.zeros(b, 1), torch.ones(b, 1), idx.float()[:, None]], dim=1).to(dev)
        log = ts.step(rays, gt.to(dev), rng={k: v.to(dev) for k, v in rng.items()})
        losses.append(log["train/loss"])
    losses = [float(x) for x in torch.stack(losses).cpu()]
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    return heldout_psnr(sd, held, nc, nf), losses
