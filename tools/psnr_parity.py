"""PSNR parity of the bf16 throughput mode against the fp32 parity mode (which matches the reference's CPU path to
1e-4, tests/test_gpu_model.py): train the config-2 model on a synthetic dynamic scene in both modes from the same
initialisation, ray batches and random draws, and compare the held-out PSNR of the fine render.

    python tools/psnr_parity.py [steps] [rays_per_step]

The scene is analytic (no dataset in this image): rays hit a unit sphere of colour that depends on the hit point
and, through the image id, on time; the rest is a background gradient.  Prints one JSON line."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import hypernerf_torch_amd as HN
from hypernerf_torch_amd.hypernerf import model_utils
from hypernerf_torch_amd.hypernerf.models import NerfModel
from hypernerf_torch_amd.losses import MSELoss, psnr
from gpu_common import EMB

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = "cuda:0"


def scene(n, gen):
    """n rays (n,9) towards a moving sphere + their colours (n,3)."""
    o = torch.rand(n, 3, generator=gen) * 0.2 + torch.tensor([-0.1, -0.1, -1.1])
    tgt = (torch.rand(n, 3, generator=gen) - 0.5) * torch.tensor([1.2, 1.2, 0.0])
    d = torch.nn.functional.normalize(tgt - o, dim=-1)
    img = torch.randint(0, 100, (n,), generator=gen)
    t = img.float() / 100.0
    c = torch.stack([0.25 * torch.sin(2 * math.pi * t), 0.25 * torch.cos(2 * math.pi * t), torch.zeros(n)], -1)
    oc = o - c
    bq = (oc * d).sum(-1)
    disc = bq * bq - ((oc * oc).sum(-1) - 0.3 ** 2)
    hit = disc > 0
    th = -bq - torch.sqrt(disc.clamp_min(0))
    p = o + th[:, None] * d
    nrm = torch.nn.functional.normalize(p - c, dim=-1)
    col_s = 0.5 + 0.5 * torch.stack([nrm[:, 0], nrm[:, 1], torch.sin(3 * nrm[:, 2] + 2 * math.pi * t)], -1)
    col_b = torch.stack([0.5 + 0.4 * d[:, 0], 0.5 + 0.4 * d[:, 1], 0.3 + 0.0 * d[:, 2]], -1)
    col = torch.where(hit[:, None], col_s, col_b).clamp(0, 1)
    rays = torch.cat([o, d, torch.zeros(n, 1), torch.full((n, 1), 2.0), img.float()[:, None]], 1)
    return rays, col


def run(precision, seed=0):
    HN.set_precision(precision)
    torch.manual_seed(0)
    m = NerfModel(EMB, near=0.2, far=2.0, n_samples_coarse=64, n_samples_fine=64, noise_std=1.0,
                  hyper_slice_method="bendy_sheet", use_warp=True, use_nerf_embed=True, use_alpha_cond=True,
                  view_fourier_dim=6).to(dev)
    arena = HN.ParamArena(m.parameters())
    opt = torch.optim.Adam([arena.flat_param], lr=1e-3, eps=1e-8, fused=True)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.1 ** (1.0 / steps))
    g = torch.Generator().manual_seed(7)
    train_rays, train_col = scene(65536, g)
    test_rays, test_col = scene(8192, g)
    train_rays, train_col, test_rays, test_col = (x.to(dev) for x in (train_rays, train_col, test_rays, test_col))
    loss_fn = MSELoss()
    gd = torch.Generator(device=dev).manual_seed(11 + seed)
    extra = {'nerf_alpha': None, 'warp_alpha': None, 'hyper_alpha': None, 'hyper_sheet_alpha': None}
    t0 = time.perf_counter()
    for it in range(steps):
        sel = torch.randint(0, train_rays.shape[0], (B,), device=dev, generator=gd)
        rng = {"t_rand": torch.rand(B, 64, device=dev, generator=gd), "noise_coarse": torch.randn(B, 64, 1, device=dev, generator=gd),
               "u": torch.rand(B, 64, device=dev, generator=gd), "noise_fine": torch.randn(B, 128, 1, device=dev, generator=gd)}
        out = m(model_utils.prepare_ray_dict(train_rays[sel]), extra, rng=rng)
        loss = loss_fn(out, train_col[sel])
        arena.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        if os.environ.get("HN_PSNR_DEBUG") and it % 1000 == 0:
            import resource
            print(f"[{precision} seed {seed} it {it}] cuda alloc {torch.cuda.memory_allocated() >> 20} MiB reserved "
                  f"{torch.cuda.memory_reserved() >> 20} MiB host maxrss {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10} MiB "
                  f"fds {len(os.listdir('/proc/self/fd'))}", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m.eval()
    vals = []
    with torch.no_grad():
        for i in range(0, test_rays.shape[0], 2048):
            rng = {"t_rand": torch.full((2048, 64), 0.5, device=dev), "u": torch.linspace(0, 1, 66, device=dev)[1:-1].expand(2048, 64).contiguous()}
            old = m.noise_std
            m.noise_std = None
            out = m(model_utils.prepare_ray_dict(test_rays[i:i + 2048]), extra, rng=rng)
            m.noise_std = old
            vals.append(((out["fine"]["rgb"] - test_col[i:i + 2048]) ** 2).mean())
    mse = torch.stack(vals).mean()
    return float(-10.0 * torch.log10(mse)), float(loss.detach()), dt


if __name__ == "__main__":
    res = {}
    seeds = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0"])]
    for prec in os.environ.get("HN_PSNR_MODES", "fp32,bf16").split(","):     # e.g. HN_PSNR_MODES=bf16: one mode only
        runs = [run(prec, sd) for sd in seeds]          # same init; the seed changes ray batches and draws
        res[prec] = {"test_psnr_db": [round(r[0], 3) for r in runs],
                     "mean_psnr_db": round(sum(r[0] for r in runs) / len(runs), 3),
                     "final_train_loss": [r[1] for r in runs], "train_seconds": round(sum(r[2] for r in runs), 2)}
    if "bf16" in res and "fp32" in res:
        res["psnr_gap_db"] = round(res["bf16"]["mean_psnr_db"] - res["fp32"]["mean_psnr_db"], 3)
    res["seeds"] = seeds
    res["config"] = f"{steps} steps x {B} rays x (64+64) samples, Adam lr 1e-3 -> 1e-4, synthetic moving-sphere scene"
    print(json.dumps(res))
