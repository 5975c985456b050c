"""GPU parity tests (MI355X) for the per-ray kernels and the hardware layout probe.  HIP path vs the CPU
oracle on identical seeded inputs; bit-exact for sample depths and searchsorted indices."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import hashprng as H
import hypernerf_torch_amd as HN
from gpu_common import DEV, EMB, assert_close, assert_grad_close, rays_for
from hypernerf_torch_amd import _lib as L
from hypernerf_torch_amd import functional as F
from hypernerf_torch_amd.hypernerf import model_utils as MU
from oracle import hypernerf_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rho(i, h):
    return (i & 3) + 8 * (i >> 2) + 4 * h


def test_probe_mfma_layout():
    """The MFMA lane maps and the LDS-DMA addressing the kernels are built on, checked on real hardware."""
    lib = L.load()
    o1 = torch.zeros(2048, device=DEV)
    o2 = torch.zeros(1024, device=DEV)
    o3 = torch.zeros(512, device=DEV)
    o3[256:] = torch.arange(256, device=DEV, dtype=torch.float32) * 3.0 + 1.0
    L.check(lib.hn_probe_mfma(L.ptr(o1), L.ptr(o2), L.ptr(o3), L.stream_handle()), "probe")
    torch.cuda.synchronize()
    a = o1.cpu().numpy()
    d1, d2 = a[:1024].reshape(64, 16), a[1024:].reshape(64, 16)
    f = o2.cpu().numpy().reshape(64, 16)
    for l in range(64):
        col, h = l & 31, l >> 5
        for q in range(16):
            i = rho(q, h)
            assert d1[l, q] == i, ("bf16 C/D row map", l, q, d1[l, q], i)
            assert d2[l, q] == (col & 15), ("bf16 A/B k pairing", l, q, d2[l, q])
            assert abs(f[l, q] - (i + (i + 1000) * 0.5 * col)) < 1e-3, ("f32 mfma", l, q, f[l, q])
    g = o3.cpu().numpy()
    assert np.array_equal(g[:256], g[256:]), "global_load_lds: LDS image must be lane-linear"


def test_sample_along_rays_bitexact():
    o, d, _ = rays_for(3, 37)
    t = H.uniform(3, "t", (37, 64), 0, 1)
    z_ref, p_ref = O.sample_along_rays(o, d, 64, 0.0, 1.0, t)
    z, p = MU.sample_along_rays(o.to(DEV), d.to(DEV), 64, 0.0, 1.0, True, False, t_rand=t.to(DEV))
    assert torch.equal(z.cpu(), z_ref), "stratified z must be bit-exact"
    assert torch.equal(p.cpu(), p_ref), "points must be bit-exact"
    z_ref, p_ref = O.sample_along_rays(o, d, 16, 0.5, 4.0, t[:, :16].contiguous(), lindisp=True)
    z, p = MU.sample_along_rays(o.to(DEV), d.to(DEV), 16, 0.5, 4.0, True, True, t_rand=t[:, :16].contiguous().to(DEV))
    assert torch.equal(z.cpu(), z_ref) and torch.equal(p.cpu(), p_ref)
    z_ref, _ = O.sample_along_rays(o, d, 16, 0.0, 1.0, None)
    z, _ = MU.sample_along_rays(o.to(DEV), d.to(DEV), 16, 0.0, 1.0, False, False)
    assert torch.equal(z.cpu(), z_ref)


def test_golden_g08(golden_dir):
    g = np.load(os.path.join(golden_dir, "g08_sample.npz"))
    o, d = torch.from_numpy(g["o"]).to(DEV), torch.from_numpy(g["d"]).to(DEV)
    z, p = MU.sample_along_rays(o, d, 16, 0.0, 1.0, True, False, t_rand=torch.from_numpy(g["t_rand"]).to(DEV))
    assert np.array_equal(z.cpu().numpy(), g["z_strat"]) and np.array_equal(p.cpu().numpy(), g["p_strat"])


@pytest.mark.parametrize("s", [8, 64, 128, 192, 70])
@pytest.mark.parametrize("variant", [0, 1])
def test_composite_vs_oracle(s, variant):
    b = 19
    o, d, _ = rays_for(5, b)
    rgb = H.uniform(5, "rgb", (b, s, 3), 0, 1)
    raw = H.uniform(5, "raw", (b, s), -3, 6)
    raw[0] = -30.0
    raw[1] = 25.0
    noise = H.normal(5, "noise", (b, s)) * 0.5
    z, _ = torch.sort(H.uniform(5, "z", (b, s), 0, 1), dim=-1)
    warped = H.uniform(5, "wp", (b, s, 7), -1, 1)
    rgb_t, raw_t = rgb.clone().requires_grad_(True), raw.clone().requires_grad_(True)
    if variant == 0:
        ref = O.volumetric_rendering(rgb_t, torch.nn.functional.softplus(raw_t + noise), z, d, white_bg=False)
        refs = [ref["rgb"], ref["depth"], ref["acc"], ref["weights"], ref["med_depth"]]
    else:
        deltas = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1) * torch.norm(d[:, None, :], dim=-1)
        alphas = 1 - torch.exp(-deltas * torch.relu(raw_t + noise))
        shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
        w = alphas * torch.cumprod(shifted, -1)[:, :-1]
        refs = [(w[..., None] * rgb_t).sum(-2) + 1 - w.sum(1)[:, None], (w * z).sum(-1), w.sum(1), w]
    rgb_g = rgb.to(DEV).requires_grad_(True)
    raw_g = raw.to(DEV).requires_grad_(True)
    outs = F.composite(rgb_g, raw_g, noise.to(DEV), z.to(DEV), d.to(DEV), warped.to(DEV) if variant == 0 else None,
                       variant=variant, white_bg=(variant == 1), sample_at_infinity=True, want_median=(variant == 0))
    names = ["rgb", "depth", "acc", "weights", "med_depth"]
    for i, r in enumerate(refs):
        assert_close(outs[i], r, 2e-5, f"composite {names[i]} S={s} v={variant}")
    if variant == 0:
        _, di = O.median_depth_index(ref["weights"])
        mp = torch.gather(warped, -2, di[..., None, None])[:, 0, 0]
        assert_close(outs[5], mp, 1e-6, "med_points")
    gr = [H.uniform(6, f"g{i}", tuple(refs[i].shape), -1, 1) for i in range(4)]
    sum((r * g).sum() for r, g in zip(refs[:4], gr)).backward()
    sum((outs[i] * gr[i].to(DEV)).sum() for i in range(4)).backward()
    assert_grad_close(rgb_g.grad, rgb_t.grad, 5e-5, "d rgb")
    assert_grad_close(raw_g.grad, raw_t.grad, 5e-5, "d raw")


def test_per_ray_kernels_random_sizes_fuzz():
    """Compositing (HyperNeRF variant) and inverse-CDF sampling on 30 seeded random (rays, coarse, fine) sizes — 1..300
    rays, 2..200 coarse and 1..200 fine samples, rays with all-zero, one-hot and huge densities — against the oracle:
    composite outputs 2e-5, its gradients 5e-5 / 2e-4, fine-sample indices / samples / merged sorted depths BIT-exact."""
    rs = np.random.RandomState(4321)
    for case in range(30):
        b = int(rs.choice([1, 2, 3, 4, 5, 17, 64, 129, 300]))
        nc = int(rs.choice([2, 3, 5, 8, 16, 33, 64, 100, 128, 200]))
        nf = int(rs.choice([1, 2, 7, 16, 64, 65, 128, 200]))
        seed = 500 + case
        what = f"fuzz {case}: rays {b} coarse {nc} fine {nf}"
        o, d, _ = rays_for(seed, b)
        rgb = H.uniform(seed, "rgb", (b, nc, 3), 0, 1)
        raw = H.uniform(seed, "raw", (b, nc), -3, 6)
        raw[0] = -40.0                                   # softplus -> 0: a ray with (almost) all-zero weights
        if b > 1:
            raw[1] = -40.0
            raw[1, nc // 2] = 30.0                       # one-hot
        if b > 2:
            raw[2] = 30.0                                # everything absorbed by the first sample
        z, _ = torch.sort(H.uniform(seed, "z", (b, nc), 0, 1), dim=-1)
        warped = H.uniform(seed, "wp", (b, nc, 7), -1, 1)
        rgb_t, raw_t = rgb.clone().requires_grad_(True), raw.clone().requires_grad_(True)
        ref = O.volumetric_rendering(rgb_t, torch.nn.functional.softplus(raw_t), z, d, white_bg=False)
        rgb_g, raw_g = rgb.to(DEV).requires_grad_(True), raw.to(DEV).requires_grad_(True)
        outs = F.composite(rgb_g, raw_g, None, z.to(DEV), d.to(DEV), warped.to(DEV), variant=0, white_bg=False,
                           sample_at_infinity=True, want_median=True)
        for i, k in enumerate(["rgb", "depth", "acc", "weights", "med_depth"]):
            assert_close(outs[i], ref[k], 2e-5, f"{what} composite {k}")
        gr = [H.uniform(seed, f"g{i}", tuple(ref[k].shape), -1, 1) for i, k in enumerate(["rgb", "depth", "acc", "weights"])]
        sum((ref[k] * g).sum() for k, g in zip(["rgb", "depth", "acc", "weights"], gr)).backward()
        sum((outs[i] * gr[i].to(DEV)).sum() for i in range(4)).backward()
        assert_grad_close(rgb_g.grad, rgb_t.grad, 5e-5, what + " d rgb")
        # 2e-4: in the smallest cases every ray is one of the degenerate ones and the whole gradient is ~1e-4
        assert_grad_close(raw_g.grad, raw_t.grad, 2e-4, what + " d raw")
        if nc >= 3:
            w = ref["weights"].detach()
            u = H.uniform(seed, "u", (b, nf), 0, 1)
            mid = 0.5 * (z[:, 1:] + z[:, :-1])
            z_ref, p_ref, inds_ref = O.sample_pdf(mid, w[:, 1:-1], o, d, z, u)
            z_all, pts, inds, zs = F.sample_pdf(outs[3].detach(), z.to(DEV), u.to(DEV), o.to(DEV), d.to(DEV))
            wg = outs[3].detach().cpu()
            if torch.equal(wg, w):                      # same fp32 weights in -> bit-identical sampling out
                assert torch.equal(inds.cpu(), inds_ref), what + " indices"
                assert torch.equal(z_all.cpu(), z_ref), what + " merged depths"
                assert torch.equal(pts.cpu(), p_ref), what + " points"
            else:                                       # weights differ in the last ulp: sample from the oracle's
                z_all, pts, inds, zs = F.sample_pdf(w.to(DEV), z.to(DEV), u.to(DEV), o.to(DEV), d.to(DEV))
                assert torch.equal(inds.cpu(), inds_ref), what + " indices (oracle weights)"
                assert torch.equal(z_all.cpu(), z_ref), what + " merged depths (oracle weights)"
            assert bool((z_all[:, 1:] >= z_all[:, :-1]).all()), what + " sortedness"


def test_sample_pdf_bitexact_indices():
    b, nc, nf = 333, 64, 128
    o, d, _ = rays_for(7, b)
    z, _ = torch.sort(H.uniform(7, "z", (b, nc), 0, 1), dim=-1)
    w = H.uniform(7, "w", (b, nc), 0, 1) ** 4
    w[0] = 0.0
    w[1] = 0.0
    w[1, 20] = 1.0
    u = H.uniform(7, "u", (b, nf), 0, 1)
    u[2, 0] = 0.0
    mid = 0.5 * (z[:, 1:] + z[:, :-1])
    z_ref, p_ref, inds_ref = O.sample_pdf(mid, w[:, 1:-1], o, d, z, u)
    zs_ref, _ = O.piecewise_constant_pdf(mid, w[:, 1:-1], u)
    z_all, pts, inds, zs = F.sample_pdf(w.to(DEV), z.to(DEV), u.to(DEV), o.to(DEV), d.to(DEV))
    assert torch.equal(inds.cpu(), inds_ref), "searchsorted indices must be bit-exact"
    assert torch.equal(zs.cpu(), zs_ref), "fine samples must be bit-exact"
    assert torch.equal(z_all.cpu(), z_ref), "merged sorted depths must be bit-exact"
    assert torch.equal(pts.cpu(), p_ref)
    # general form (explicit bins / weights), no merge
    zs2 = MU.piecewise_constant_pdf(mid.to(DEV), w[:, 1:-1].contiguous().to(DEV), nf, True, u=u.to(DEV))
    assert torch.equal(zs2.cpu(), zs_ref)


def test_golden_g10(golden_dir):
    g = np.load(os.path.join(golden_dir, "g10_pdf.npz"))
    T = lambda k: torch.from_numpy(g[k]).to(DEV)
    z_all, pts = MU.sample_pdf(T("bins"), T("w"), T("o"), T("d"), T("z"), 16, True, u=T("u"))
    _, _, inds, zs = F.sample_pdf(T("w"), T("z"), T("u"), bins=T("bins"), merge=False)
    assert np.array_equal(inds.cpu().numpy(), g["inds"]), "indices vs the reference itself"
    np.testing.assert_allclose(zs.cpu().numpy(), g["z_samples"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(z_all.cpu().numpy(), g["z_all"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(pts.cpu().numpy(), g["pts"], rtol=1e-6, atol=1e-6)


def test_legacy_sample_pdf_wrapper(golden_dir):
    """`models.rendering.sample_pdf(bins, weights, N_importance, det, eps)` — the nerf_pl wrapper itself
    (models/rendering.py:14-55): explicit `u` against the reference's samples of g10, the deterministic branch
    (u = linspace(0, 1, N)) against the fixture's `z_samples_det` and the oracle, the random branch by its range and
    sortedness per bin, and the unsupported `eps`."""
    from hypernerf_torch_amd.models import rendering as R
    g = np.load(os.path.join(golden_dir, "g10_pdf.npz"))
    T = lambda k: torch.from_numpy(g[k]).to(DEV)
    zs = R.sample_pdf(T("bins"), T("w"), 16, u=T("u"))
    np.testing.assert_allclose(zs.cpu().numpy(), g["z_samples"], rtol=1e-6, atol=1e-6)
    zd = R.sample_pdf(T("bins"), T("w"), 16, det=True)
    np.testing.assert_allclose(zd.cpu().numpy(), g["z_samples_det"], rtol=1e-6, atol=1e-6)
    u_det = torch.linspace(0, 1, 16).expand(8, 16).contiguous()
    zo, _ = O.piecewise_constant_pdf(torch.from_numpy(g["bins"]), torch.from_numpy(g["w"]), u_det)
    assert torch.equal(zd.cpu(), zo)
    torch.manual_seed(5)
    zr = R.sample_pdf(T("bins"), T("w"), 64)
    assert zr.shape == (8, 64)
    assert bool((zr >= T("bins")[:, :1]).all()) and bool((zr <= T("bins")[:, -1:]).all())
    with pytest.raises(NotImplementedError):
        R.sample_pdf(T("bins"), T("w"), 16, eps=1e-3)


def test_embed_and_posenc():
    tab = H.uniform(9, "tab", (100, 8), -1, 1)
    idx = torch.from_numpy((H.uniform01(9, "i", 57) * 100).astype(np.int64))
    tg = tab.to(DEV).requires_grad_(True)
    out = F.embed_lookup(tg, idx.to(DEV))
    assert torch.equal(out.cpu(), tab[idx])
    g = H.uniform(9, "g", (57, 8), -1, 1)
    (out * g.to(DEV)).sum().backward()
    tr = tab.clone().requires_grad_(True)
    (tr[idx] * g).sum().backward()
    assert_grad_close(tg.grad, tr.grad, 1e-6, "embedding grad")
    x = H.uniform(9, "x", (11, 5, 3), -2, 2)
    xg = x.to(DEV).requires_grad_(True)
    freqs = (2.0 ** torch.arange(10)).float().to(DEV)
    y = F.posenc(xg, freqs, True)
    xr = x.clone().requires_grad_(True)
    yr = O.posenc_orig(xr, 10)
    assert_close(y, yr, 2e-6, "posenc_orig")
    gy = H.uniform(9, "gy", tuple(yr.shape), -1, 1)
    (y * gy.to(DEV)).sum().backward()
    (yr * gy).sum().backward()
    assert_grad_close(xg.grad, xr.grad, 1e-5, "posenc grad")


def test_cpu_tensor_fails_loudly():
    o, d, _ = rays_for(1, 4)
    with pytest.raises(L.HnError):
        MU.sample_along_rays(o, d, 8, 0.0, 1.0, True, False)


@pytest.mark.gpu
@pytest.mark.parametrize("ndc", [False, True])
def test_generate_rays_vs_golden_and_oracle(golden_dir, ndc):
    """hn_generate_rays against the reference's own ray_utils outputs (tests/golden/g15_rays.npz) and, at a real
    image size, against the oracle.  Tolerance 2e-6 of the tensor scale (3-term dot products in a different order)."""
    import numpy as np
    from hypernerf_torch_amd import functional as F
    from oracle import hypernerf_oracle as O
    g = np.load(os.path.join(golden_dir, "g15_rays.npz"))
    hh, ww, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    c2w = torch.from_numpy(g["c2w"])
    rays = F.generate_rays(hh, ww, focal, c2w.to(DEV), near=0.0, far=1.0, ndc=ndc, image_id=5)
    assert rays.shape == (hh * ww, 9)
    ro, rd = (g["ndc_o"], g["ndc_d"]) if ndc else (g["rays_o"], g["rays_d"])
    assert_close(rays[:, 0:3], torch.from_numpy(ro), 2e-6, "origins")
    assert_close(rays[:, 3:6], torch.from_numpy(rd), 2e-6, "directions")
    assert torch.equal(rays[:, 6:9].cpu(), torch.tensor([0.0, 1.0, 5.0]).expand(hh * ww, 3))
    big = F.generate_rays(378, 504, 407.5, c2w.to(DEV), near=0.2, far=1.5, ndc=ndc)
    ref = O.image_rays(378, 504, 407.5, c2w, 0.2, 1.5, ndc)
    assert big.shape == ref.shape == (378 * 504, 8)
    assert_close(big, ref, 2e-6, "full image rays")


@pytest.mark.gpu
@pytest.mark.parametrize("weight_decay", [0.0, 1e-2])
def test_arena_adam_matches_torch_adam(weight_decay):
    """hn_adam_step (ArenaAdam) against torch.optim.Adam over 6 steps on the same gradients: parameters to 1e-6 of
    their scale, moments likewise; the gradient buffer is cleared by the step; the device step counter counts."""
    import hypernerf_torch_amd as HN
    torch.manual_seed(5)
    shapes = [(37, 19), (19,), (5, 3, 2), (1,), (130, 64)]
    p1 = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    p2 = [torch.nn.Parameter(p.detach().clone()) for p in p1]
    ref = torch.optim.Adam(p1, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=weight_decay)
    arena = HN.ParamArena(p2)
    opt = HN.ArenaAdam(arena, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=weight_decay)
    for it in range(6):
        for a, b in zip(p1, p2):
            g = torch.randn_like(a) * (0.1 + it)
            a.grad = g.clone()
            b.grad.copy_(g)
        ref.step()
        opt.step()
        assert float(arena.grad.abs().max()) == 0.0
        for a, b in zip(p1, p2):
            assert_close(b, a, 5e-6, f"step {it} parameters")
    assert float(opt.step_count) == 6.0
    st = ref.state[p1[0]]
    o0, n0 = arena.offsets[0], p1[0].numel()
    assert_close(opt.exp_avg[o0:o0 + n0].view_as(p1[0]), st["exp_avg"], 5e-6, "exp_avg")
    assert_close(opt.exp_avg_sq[o0:o0 + n0].view_as(p1[0]), st["exp_avg_sq"], 5e-6, "exp_avg_sq")


def test_random_draws_one_launch_statistics_and_replay():
    """hn_random_fill (the render step's draws in one launch; replaces torch.rand / torch.randn of model_utils.py:31,
    226, 300-317): ranges, moments and a Kolmogorov-Smirnov distance of both distributions, independence across
    buffers and launches, reproducibility from the seed, ragged sizes, and a HIP-graph replay drawing fresh numbers
    (the offset lives on the device and is advanced by the kernel itself)."""
    from scipy import stats
    F.seed_draws(1234)
    shapes = [((1024, 64), "uniform"), ((1024, 64, 1), "normal"), ((1024, 64), "uniform"), ((1024, 128, 1), "normal"),
              ((7,), "normal"), ((5, 3), "uniform")]
    a = F.random_draws(shapes, DEV)
    assert [tuple(t.shape) for t in a] == [s for s, _ in shapes]
    u0, n0, u1, n1 = (t.cpu().double().reshape(-1).numpy() for t in a[:4])
    for u in (u0, u1):
        assert u.min() >= 0.0 and u.max() < 1.0
        assert abs(u.mean() - 0.5) < 4 * (1 / 12 / u.size) ** 0.5 and abs(u.var() - 1 / 12) < 2e-3
        assert stats.kstest(u, "uniform").statistic < 1.63 / u.size ** 0.5            # 1 % level
    for n in (n0, n1):
        assert np.isfinite(n).all() and abs(n.mean()) < 4 / n.size ** 0.5 and abs(n.var() - 1.0) < 2e-2
        assert stats.kstest(n, "norm").statistic < 1.63 / n.size ** 0.5
        assert np.abs(n).max() < 6.0
    assert abs(np.corrcoef(u0, u1)[0, 1]) < 0.02 and abs(np.corrcoef(n0, n1[:n0.size])[0, 1]) < 0.02
    assert abs(np.corrcoef(n0[:-1], n0[1:])[0, 1]) < 0.02                            # the two Box-Muller outputs of a pair
    b = F.random_draws(shapes, DEV)
    assert not torch.equal(a[0], b[0]) and abs(np.corrcoef(u0, b[0].cpu().double().reshape(-1).numpy())[0, 1]) < 0.02
    F.seed_draws(1234)
    c = F.random_draws(shapes, DEV)
    assert all(torch.equal(x, y) for x, y in zip(a, c)), "same seed, same sequence"
    F.seed_draws(99)
    assert not torch.equal(F.random_draws(shapes, DEV)[0], a[0])
    # graph replay: the captured launch must not repeat its numbers
    F.seed_draws(5)
    F.random_draws(shapes[:2], DEV)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = F.random_draws(shapes[:2], DEV)
    seen = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        seen.append([t.clone() for t in outs])
    assert not torch.equal(seen[0][0], seen[1][0]) and not torch.equal(seen[1][1], seen[2][1])


def test_model_draws_are_one_launch_and_reproducible():
    """NerfModel.forward draws t_rand, u and both levels' noise through ONE launch (no ATen RNG kernel left in the step) —
    since round 6 the step head, hn_render_prologue, which also packs the stale weight streams and places the coarse
    samples (hn_random_fill / hn_pack_units_multi / hn_sample_along_rays with HN_PROLOGUE=0: the same numbers, the Philox
    counters do not depend on the launch that hosts them); torch.manual_seed + seed_draws reproduces a stochastic forward
    exactly; supplied draws are honoured (the fixtures' path) and HN_FAST_DRAWS=0 falls back to torch's generator."""
    from hypernerf_torch_amd.hypernerf import models as M
    HN.set_precision("fp32")
    m = M.NerfModel(EMB, n_samples_coarse=16, n_samples_fine=16, noise_std=1.0, hyper_slice_method="bendy_sheet",
                    use_nerf_embed=True, use_alpha_cond=True, view_fourier_dim=6).to(DEV)
    o, d, idx = rays_for(3, 24)
    rays = {"origins": o.to(DEV), "directions": d.to(DEV), "viewdirs": None,
            "metadata": {k: idx.to(DEV) for k in ("warp", "camera", "appearance", "time")}}
    L.KERNEL_TIMES = {}
    try:
        with torch.no_grad():
            F.seed_draws(7)
            a = m(rays, {})
        names = [k.split("[")[0] for k in L.collect_kernel_times()]
    finally:
        L.KERNEL_TIMES = None
    assert names.count("hn_render_prologue") == 1 and names.count("hn_random_fill") == 0
    assert names.count("hn_sample_along_rays") == 0 and names.count("hn_pack_units_multi") == 0 and names.count("hn_pack_units") == 0
    # the same forward with the head as separate launches: bit-identical (same draws, same sample arithmetic, same packing)
    F.PROLOGUE = False
    L.KERNEL_TIMES = {}
    try:
        from hypernerf_torch_amd import machine as MM
        MM.note_parameters_changed()            # make the weight streams stale again: this pass packs them itself
        with torch.no_grad():
            F.seed_draws(7)
            a2 = m(rays, {})
        names2 = [k.split("[")[0] for k in L.collect_kernel_times()]
    finally:
        L.KERNEL_TIMES = None
        F.PROLOGUE = True
    assert names2.count("hn_random_fill") == 1 and names2.count("hn_sample_along_rays") == 1 and "hn_render_prologue" not in names2
    for lvl in ("coarse", "fine"):
        for k in ("rgb", "weights", "points", "warped_points", "depth"):
            assert torch.equal(a[lvl][k], a2[lvl][k]), f"step head as one launch vs separate launches: {lvl}/{k}"
    with torch.no_grad():
        b = m(rays, {})
        F.seed_draws(7)
        c = m(rays, {})
    assert not torch.equal(a["fine"]["rgb"], b["fine"]["rgb"]), "a second forward must draw new numbers"
    assert torch.equal(a["fine"]["rgb"], c["fine"]["rgb"]) and torch.equal(a["coarse"]["weights"], c["coarse"]["weights"])
    # supplied draws win over the generator
    rng = {"t_rand": H.uniform(3, "t", (24, 16), 0, 1).to(DEV), "u": H.uniform(3, "u", (24, 16), 0, 1).to(DEV),
           "noise_coarse": H.normal(3, "n1", (24, 16, 1)).to(DEV), "noise_fine": H.normal(3, "n2", (24, 32, 1)).to(DEV)}
    with torch.no_grad():
        x = m(rays, {}, rng=rng)
        y = m(rays, {}, rng=rng)
    assert torch.equal(x["fine"]["rgb"], y["fine"]["rgb"])


def test_ls_bench_handoff_pipeline_checks_out():
    """tools/ls_bench.hip (round 4, step A: a persistent pipeline of workgroups, each keeping one layer's W^T and dW in
    registers and handing dZ to the next CU through a ring — write-through `sc1` stores, `sc1` LDS-DMA loads, published /
    consumed counters) checks every word of its result against plain reference kernels.  Run here at a small size in
    the cross-XCD and the XCD-local form so that the hand-off recipe DESIGN.md section 8.0 documents stays a tested one
    (the timing is the tool's business, not this test's)."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "ls_bench")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(exe + ".hip"):
        from hypernerf_torch_amd import _lib
        subprocess.run([_lib.hipcc_path(), "--offload-arch=gfx950", "-O3", "-o", exe, exe + ".hip"], check=True,
                       capture_output=True, timeout=300)
    for args in (["131072", "3", "2", "2", "85", "16", "1"], ["262144", "8", "2", "2", "32", "16", "1"],
                 ["262144", "8", "2", "4", "32", "16", "1"], ["131072", "8", "2", "3", "32", "16", "1"]):
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and "check ok" in out.stdout and "SPIN TIMEOUT" not in out.stdout, (args, out.stdout[-600:])


# ---- a level in two parts (round 5: the fine level re-uses the coarse level's warp / sheet results) ----------------
@pytest.mark.parametrize("b,nc,nf", [(333, 64, 128), (17, 64, 64), (5, 8, 8), (3, 33, 65), (64, 128, 200)])
def test_sample_pdf_split_permutation(b, nc, nf):
    """hn_sample_pdf_split: the same depths, points, indices and samples as hn_sample_pdf BIT for bit, plus the merge
    permutation (sorted position -> entry of cat(z, z_samples), equal depths in that order) and the new samples' points."""
    o, d, _ = rays_for(7, b)
    z, _ = torch.sort(H.uniform(7, "z", (b, nc), 0, 1), dim=-1)
    w = H.uniform(7, "w", (b, nc), 0, 1) ** 4
    w[0] = 0.0
    u = H.uniform(7, "u", (b, nf), 0, 1)
    if nf > 2:
        u[1, 1] = u[1, 0]                       # two equal new samples
    a = F.sample_pdf(w.to(DEV), z.to(DEV), u.to(DEV), o.to(DEV), d.to(DEV))
    s = F.sample_pdf(w.to(DEV), z.to(DEV), u.to(DEV), o.to(DEV), d.to(DEV), split=True)
    for x, y, name in zip(a, s[:4], ["z_all", "pts", "inds", "z_samples"]):
        assert torch.equal(x, y), name
    perm, pts_new = s[4].cpu().long(), s[5].cpu()
    cat = torch.cat([z, s[3].cpu()], dim=1)
    assert torch.equal(torch.sort(perm, dim=1)[0], torch.arange(nc + nf).expand(b, -1)), "perm is a permutation per ray"
    assert torch.equal(torch.gather(cat, 1, perm), s[0].cpu()), "z_all == cat(z, z_samples)[perm]"
    same = s[0].cpu()[:, 1:] == s[0].cpu()[:, :-1]
    assert bool((perm[:, 1:][same] > perm[:, :-1][same]).all()), "equal depths keep the order of cat(z, z_samples)"
    zs = s[3].cpu()
    ref_new = o[:, None, :] + zs[..., None] * d[:, None, :]          # unfused mul + add, as the sorted points
    assert torch.equal(pts_new, ref_new), "points of the new samples"


@pytest.mark.parametrize("nc,nf", [(64, 64), (64, 128), (8, 8), (33, 70), (128, 200)])
def test_composite_two_parts_equals_one_part(nc, nf):
    """hn_composite_* reading a level as two parts through the permutation == the same level gathered into sorted
    order first: forward outputs, the sorted warped rows and every gradient BIT for bit."""
    b, s = 23, nc + nf
    o, d, _ = rays_for(9, b)
    rgb = H.uniform(9, "rgb", (b, s, 3), 0, 1)
    raw = H.uniform(9, "raw", (b, s), -3, 6)
    noise = H.normal(9, "noise", (b, s)) * 0.5
    warped = H.uniform(9, "wp", (b, s, 7), -1, 1)
    z_old, _ = torch.sort(H.uniform(9, "z", (b, nc), 0, 1), dim=-1)
    z_new = H.uniform(9, "zn", (b, nf), 0, 1)
    z_all, perm = torch.sort(torch.cat([z_old, z_new], 1), dim=1, stable=True)
    keep = (H.uniform(9, "keep", (b, s), 0, 1) > 0.2).float()
    g = lambda t, p: torch.gather(t, 1, p if t.dim() == 2 else p[..., None].expand(-1, -1, t.shape[-1]))
    # one part: everything already in sorted order
    rgb1_, raw1_ = g(rgb, perm).to(DEV).requires_grad_(True), g(raw, perm).to(DEV).requires_grad_(True)
    one = F.composite(rgb1_, raw1_, noise.to(DEV), z_all.to(DEV), d.to(DEV), g(warped, perm).to(DEV), variant=0,
                      sample_at_infinity=True, want_median=True, dust_threshold=0.01, keep=keep.to(DEV), noise_scale=0.7)
    # two parts in their own order
    parts = [t.to(DEV).requires_grad_(True) for t in (rgb[:, :nc], raw[:, :nc], rgb[:, nc:], raw[:, nc:])]
    two = F.composite(parts[0], parts[1], noise.to(DEV), z_all.to(DEV), d.to(DEV), warped[:, :nc].contiguous().to(DEV),
                      variant=0, sample_at_infinity=True, want_median=True, dust_threshold=0.01, keep=keep.to(DEV),
                      noise_scale=0.7, rgb1=parts[2], raw1=parts[3], warped1=warped[:, nc:].contiguous().to(DEV),
                      perm=perm.int().to(DEV))
    for i, name in enumerate(["rgb", "depth", "acc", "weights", "med_depth", "med_points"]):
        assert torch.equal(one[i], two[i]), name
    assert torch.equal(two[6].cpu(), g(warped, perm)), "sorted warped rows"
    gr = [H.uniform(10, f"g{i}", tuple(one[i].shape), -1, 1).to(DEV) for i in range(4)]
    sum((one[i] * gr[i]).sum() for i in range(4)).backward()
    sum((two[i] * gr[i]).sum() for i in range(4)).backward()
    d_rgb = torch.zeros(b, s, 3).scatter_(1, perm[..., None].expand(-1, -1, 3), rgb1_.grad.cpu())
    d_raw = torch.zeros(b, s).scatter_(1, perm, raw1_.grad.cpu())
    assert torch.equal(torch.cat([parts[0].grad, parts[2].grad], 1).cpu(), d_rgb), "d rgb"
    assert torch.equal(torch.cat([parts[1].grad, parts[3].grad], 1).cpu(), d_raw), "d raw"


def test_sample_pdf_merge_falls_back_for_unsorted_or_non_finite_depths():
    """Round 6: the sampler merges by RANK when the level's own depths arrive sorted and every value is a number, and
    keeps the (depth, position) bitonic sort for anything else.  Both must give `sort(cat(z, z_samples))` with equal
    depths in cat order: sorted input (rank path) and the same rays with two depths swapped (sort path) against
    torch.sort(stable=True), indices and samples bit-identical between the two (they do not depend on the merge); rays
    whose weights are NaN run through without a fault."""
    b, nc, nf = 37, 64, 96
    o, d, _ = rays_for(21, b)
    z, _ = torch.sort(H.uniform(21, "z", (b, nc), 0, 1), dim=-1)
    z[3, 10] = z[3, 11]                                   # a tie among the level's own depths
    w = H.uniform(21, "w", (b, nc), 0, 1) ** 3
    u = H.uniform(21, "u", (b, nf), 0, 1)
    u[5, 7] = u[5, 3]                                     # two equal new samples
    zu = z.clone()
    zu[:, [20, 40]] = zu[:, [40, 20]]                     # no longer sorted: the bitonic path
    for zin, name in ((z, "sorted input (rank merge)"), (zu, "unsorted input (bitonic sort)")):
        z_all, pts, inds, zs, perm, pts_new = F.sample_pdf(w.to(DEV), zin.to(DEV), u.to(DEV), o.to(DEV), d.to(DEV), split=True)
        cat = torch.cat([zin, zs.cpu()], dim=1)
        ref, ref_perm = torch.sort(cat, dim=1, stable=True)
        assert torch.equal(z_all.cpu(), ref), name + ": merged depths"
        assert torch.equal(perm.cpu().long(), ref_perm), name + ": merge permutation (equal depths in cat order)"
        assert torch.equal(pts.cpu(), o[:, None, :] + ref[..., None] * d[:, None, :]), name + ": points"
        z_all2, _, inds2, zs2 = F.sample_pdf(w.to(DEV), zin.to(DEV), u.to(DEV), o.to(DEV), d.to(DEV))
        assert torch.equal(z_all2, z_all) and torch.equal(inds2, inds) and torch.equal(zs2, zs), name + ": with / without payload"
    wn = w.clone()
    wn[2] = float("nan")
    out = F.sample_pdf(wn.to(DEV), z.to(DEV), u.to(DEV), o.to(DEV), d.to(DEV), split=True)
    torch.cuda.synchronize()
    ok = torch.ones(b, dtype=torch.bool)
    ok[2] = False
    ref = torch.sort(torch.cat([z, out[3].cpu()], dim=1), dim=1, stable=True)[0]
    assert torch.equal(out[0].cpu()[ok], ref[ok]), "the rays beside a NaN ray are untouched"


def test_composite_then_pdf_equals_the_two_launches():
    """hn_composite_sample_pdf (round 6: the coarse level's compositing and the fine level's inverse-CDF sampling as one
    launch, the weights handed over in LDS) against hn_composite_forward followed by hn_sample_pdf_split: every output of
    both, bit for bit, with and without the split outputs, and the compositing gradients."""
    for b, nc, nf, split in ((29, 64, 64, True), (5, 33, 70, False), (64, 128, 200, True)):
        o, d, _ = rays_for(23, b)
        rgb = H.uniform(23, "rgb", (b, nc, 3), 0, 1)
        raw = H.uniform(23, "raw", (b, nc), -3, 6)
        noise = H.normal(23, "noise", (b, nc))
        warped = H.uniform(23, "wp", (b, nc, 7), -1, 1)
        z, _ = torch.sort(H.uniform(23, "z", (b, nc), 0, 1), dim=-1)
        u = H.uniform(23, "u", (b, nf), 0, 1)
        args = lambda: (rgb.to(DEV).requires_grad_(True), raw.to(DEV).requires_grad_(True))
        r1, a1 = args()
        sep = F.composite(r1, a1, noise.to(DEV), z.to(DEV), d.to(DEV), warped.to(DEV), noise_scale=0.5)
        pdf = F.sample_pdf(sep[3], z.to(DEV), u.to(DEV), o.to(DEV), d.to(DEV), split=split)
        r2, a2 = args()
        fused = F.composite(r2, a2, noise.to(DEV), z.to(DEV), d.to(DEV), warped.to(DEV), noise_scale=0.5,
                            then_pdf=dict(u=u.to(DEV), origins=o.to(DEV), directions=d.to(DEV), split=split))
        n_pdf = 6 if split else 4
        assert len(fused) == len(sep) + n_pdf
        for i, (x, y) in enumerate(zip(sep, fused[:len(sep)])):
            assert torch.equal(x, y), f"compositing output {i}"
        for i, (x, y) in enumerate(zip(pdf, fused[len(sep):])):
            assert torch.equal(x, y), f"sampler output {i} ({b} x {nc} + {nf}, split {split})"
        g = H.uniform(23, "g", (b, 3), -1, 1).to(DEV)
        (sep[0] * g).sum().backward()
        (fused[0] * g).sum().backward()
        assert torch.equal(r1.grad, r2.grad) and torch.equal(a1.grad, a2.grad)
