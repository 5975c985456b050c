"""Diagnostic: find reads of uninitialised device memory.  The caching allocator is 'poisoned' — big buffers filled
with NaN bit patterns are allocated and released, so that every later torch.empty() returns NaN-filled memory — and the
model is run again: any output or gradient that differs from the clean run (or is not finite) was computed from memory
nobody wrote."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import hashprng as H                                              # noqa: E402
import hypernerf_torch_amd as HN                                  # noqa: E402
from gpu_common import DEV, EMB, load_hash, rays_for              # noqa: E402
from hypernerf_torch_amd.hypernerf import models                  # noqa: E402

CASES = {"bendy_cond": dict(hyper_slice_method="bendy_sheet", use_nerf_embed=True, use_alpha_cond=True),
         "axis": dict(hyper_slice_method="axis_aligned_plane", hyper_slice_out_dim=8, use_nerf_embed=True, use_alpha_cond=True),
         "nowarp_cond": dict(use_warp=False, hyper_slice_method=None, use_nerf_embed=True, use_alpha_cond=True),
         # BASELINE config 5: SE3Field warp (own program + hn_se3_apply) in front of the gathered template
         "se3_axis": dict(hyper_slice_method="axis_aligned_plane", hyper_slice_out_dim=8, use_nerf_embed=True,
                          use_alpha_cond=True)}


_LDS = [None]


def poison_lds(pattern):
    """Fill every CU's LDS with `pattern` (tools/lds_poison.hip, built by __graft_entry__.build()); a no-op if the
    helper library is not there."""
    import ctypes
    if _LDS[0] is None:
        path = os.path.join(ROOT, "tools", "liblds_poison.so")
        if not os.path.exists(path):        # not shipped with this snapshot: hipcc is on every GPU box
            import subprocess
            try:
                subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-o", path,
                                os.path.join(ROOT, "tools", "lds_poison.hip")], check=True, capture_output=True, timeout=300)
            except (OSError, subprocess.SubprocessError):
                pass
        _LDS[0] = ctypes.CDLL(path) if os.path.exists(path) else False
    if _LDS[0]:
        rc = _LDS[0].lds_poison(ctypes.c_uint32(pattern), 2, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        torch.cuda.synchronize()
    return bool(_LDS[0])


def poison(pattern):
    poison_lds(pattern)
    bufs = []
    for mb in (2048, 1024, 512, 256, 128, 64, 64, 32, 32, 16, 16, 8, 8, 4, 4, 2, 2, 1, 1):
        t = torch.empty(mb << 18, dtype=torch.int32, device=DEV)
        t.fill_(pattern)
        bufs.append(t)
    small = [torch.full((n,), pattern, dtype=torch.int32, device=DEV) for n in (128, 512, 2048, 8192, 32768) for _ in range(8)]
    torch.cuda.synchronize()
    del bufs, small


def run(m, rays, rng, gt, arena_mode):
    for p in m.parameters():
        if arena_mode:
            p.grad.zero_()
        else:
            p.grad = None
    out = m(rays, {}, rng=rng)
    loss = ((out["coarse"]["rgb"] - gt) ** 2).mean() + ((out["fine"]["rgb"] - gt) ** 2).mean()
    loss.backward()
    torch.cuda.synchronize()
    res = {f"{l}/{k}": out[l][k].detach().clone() for l in out for k in ("rgb", "depth", "acc", "weights", "warped_points")}
    res.update({"d " + k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    return res


def probe(precisions=("bf16", "fp32"), cases=tuple(CASES), arena_modes=(False, True),
          sizes=((96, 32, 32), (40, 8, 8), (100, 16, 24), (13, 7, 5), (1024, 64, 64)), verbose=True):
    """Returns the list of (description, tensor name, error, scale) that differ between a clean run and runs on a
    poisoned allocator (or between repeated runs: a race shows up the same way)."""
    bad = []
    for prec in precisions:
        for case in cases:
            kw = CASES[case]
            for arena_mode in arena_modes:
                for (b, nc, nf) in sizes:
                    HN.set_precision(prec)
                    m = models.NerfModel(EMB, n_samples_coarse=nc, n_samples_fine=nf, noise_std=0.5, **kw)
                    if case == "se3_axis":
                        from hypernerf_torch_amd.hypernerf import warping
                        m.warp_field = warping.SE3Field(in_ch=3)
                    load_hash(m, 77)
                    m = m.to(DEV)
                    if arena_mode:
                        arena = HN.ParamArena(m.parameters())      # noqa: F841 (keeps the views alive)
                    o, d, idx = rays_for(77, b)
                    rays = {"origins": o.to(DEV), "directions": d.to(DEV), "viewdirs": None,
                            "metadata": {k: idx.to(DEV) for k in ("warp", "camera", "appearance", "time")}}
                    rng = {"t_rand": H.uniform(77, "t", (b, nc), 0, 1).to(DEV), "u": H.uniform(77, "u", (b, nf), 0, 1).to(DEV),
                           "noise_coarse": (H.normal(77, "n1", (b, nc, 1)) * 0.5).to(DEV),
                           "noise_fine": (H.normal(77, "n2", (b, nc + nf, 1)) * 0.5).to(DEV)}
                    gt = H.uniform(77, "gt", (b, 3), 0, 1).to(DEV)
                    clean = run(m, rays, rng, gt, arena_mode)
                    # NaN / huge / 1000.0 as fp32 words; as bf16 pairs: (NaN, 0) / (3.4e38, 3.4e38) / (1000, 0)
                    for pattern in (0x7fc00000, 0x7f7f7f7f, 0x447a0000):
                        poison(pattern)
                        got = run(m, rays, rng, gt, arena_mode)
                        for k, v in got.items():
                            ref = clean[k]
                            scale = float(ref.abs().max()) + 1e-30
                            err = float((v - ref).abs().max()) if torch.isfinite(v).all() else float("inf")
                            # forward tensors: no atomics anywhere on their path -> bit-identical from run to run;
                            # gradients: float atomics land in a different order -> 1e-4 of the tensor's scale
                            tol = 0.0 if not k.startswith("d ") else 1e-4 * scale
                            if not err <= tol:
                                what = f"{prec} {case} arena={arena_mode} b={b} nc={nc} nf={nf} pattern={pattern:#x}"
                                bad.append((what, k, err, scale))
                                if verbose:
                                    print(f"UNINIT? {what}: {k} err {err:.3g} scale {scale:.3g}")
    HN.set_precision("bf16")
    return bad


if __name__ == "__main__" and len(sys.argv) == 1:
    print("suspicious tensors:", len(probe()))


def repeat_probe(case="bendy_cond", prec="bf16", size=(1024, 64, 64), n=200, arena_mode=True):
    """Race hunt: the same step `n` times on identical inputs.  Forward tensors must repeat bit for bit; gradients to
    1e-5 of their scale (float atomics).  Returns the deviations found."""
    kw = CASES[case]
    b, nc, nf = size
    HN.set_precision(prec)
    m = models.NerfModel(EMB, n_samples_coarse=nc, n_samples_fine=nf, noise_std=0.5, **kw)
    if case == "se3_axis":
        from hypernerf_torch_amd.hypernerf import warping
        m.warp_field = warping.SE3Field(in_ch=3)
    load_hash(m, 78)
    m = m.to(DEV)
    if arena_mode:
        arena = HN.ParamArena(m.parameters())      # noqa: F841
    o, d, idx = rays_for(78, b)
    rays = {"origins": o.to(DEV), "directions": d.to(DEV), "viewdirs": None,
            "metadata": {k: idx.to(DEV) for k in ("warp", "camera", "appearance", "time")}}
    rng = {"t_rand": H.uniform(78, "t", (b, nc), 0, 1).to(DEV), "u": H.uniform(78, "u", (b, nf), 0, 1).to(DEV),
           "noise_coarse": (H.normal(78, "n1", (b, nc, 1)) * 0.5).to(DEV),
           "noise_fine": (H.normal(78, "n2", (b, nc + nf, 1)) * 0.5).to(DEV)}
    gt = H.uniform(78, "gt", (b, 3), 0, 1).to(DEV)
    first = run(m, rays, rng, gt, arena_mode)
    bad = []
    for it in range(n):
        got = run(m, rays, rng, gt, arena_mode)
        for k, v in got.items():
            ref = first[k]
            scale = float(ref.abs().max()) + 1e-30
            err = float((v - ref).abs().max()) if torch.isfinite(v).all() else float("inf")
            tol = 0.0 if not k.startswith("d ") else 1e-5 * scale
            if not err <= tol:
                bad.append((it, k, err, scale))
    HN.set_precision("bf16")
    return bad


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "repeat":
    for case in CASES:
        for prec, n, size in (("bf16", 200, (1024, 64, 64)), ("fp32", 40, (1024, 64, 64)), ("bf16", 300, (40, 8, 8)),
                              ("bf16s8", 200, (1024, 64, 64)), ("bf16s8", 300, (100, 16, 24)),
                              ("bf16", 300, (100, 16, 24)), ("fp32", 100, (100, 16, 24))):
            bad = repeat_probe(case, prec, size, n)
            print(case, prec, size, "runs", n, "deviations", len(bad), bad[:4])


def legacy_probe(prec="bf16", n_rays=50, ns=16, ni=24, verbose=True):
    """The same for the nerf_pl path (models/rendering.render_rays, two NeRF networks)."""
    from hypernerf_torch_amd.models import nerf as legacy_nerf
    from hypernerf_torch_amd.models import rendering as legacy_rendering
    HN.set_precision(prec)
    nets = [legacy_nerf.NeRF(), legacy_nerf.NeRF()]
    for i, n in enumerate(nets):
        load_hash(n, 90 + i)
        n.to(DEV)
    emb = [legacy_nerf.Embedding(3, 10), legacy_nerf.Embedding(3, 4)]
    o, d, _ = rays_for(91, n_rays)
    rays = torch.cat([o, d, torch.full((n_rays, 1), 2.0), torch.full((n_rays, 1), 6.0)], 1).to(DEV)
    rng = {"perturb_rand": H.uniform(91, "p", (n_rays, ns), 0, 1).to(DEV), "noise_coarse": H.normal(91, "n1", (n_rays, ns)).to(DEV),
           "u": H.uniform(91, "u", (n_rays, ni), 0, 1).to(DEV), "noise_fine": H.normal(91, "n2", (n_rays, ns + ni)).to(DEV)}
    gt = H.uniform(91, "gt", (n_rays, 3), 0, 1).to(DEV)

    def once():
        for n in nets:
            for p in n.parameters():
                p.grad = None
        res = legacy_rendering.render_rays(nets, emb, rays, N_samples=ns, N_importance=ni, perturb=1.0, noise_std=1.0, rng=rng)
        loss = ((res["rgb_coarse"] - gt) ** 2).mean() + ((res["rgb_fine"] - gt) ** 2).mean()
        loss.backward()
        torch.cuda.synchronize()
        out = {k: v.detach().clone() for k, v in res.items()}
        for i, n in enumerate(nets):
            out.update({f"d net{i}.{k}": p.grad.detach().clone() for k, p in n.named_parameters() if p.grad is not None})
        return out

    clean = once()
    bad = []
    for pattern in (0x7fc00000, 0x7f7f7f7f, 0x447a0000):
        poison(pattern)
        got = once()
        for k, v in got.items():
            ref = clean[k]
            scale = float(ref.abs().max()) + 1e-30
            err = float((v - ref).abs().max()) if torch.isfinite(v).all() else float("inf")
            tol = 0.0 if not k.startswith("d ") else 1e-4 * scale
            if not err <= tol:
                bad.append((f"legacy {prec} pattern={pattern:#x}", k, err, scale))
                if verbose:
                    print("UNINIT?", bad[-1])
    HN.set_precision("bf16")
    return bad


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "legacy":
    for prec in ("bf16", "fp32"):
        for shape in ((50, 16, 24), (256, 64, 64), (7, 5, 3)):
            print(prec, shape, "suspicious:", len(legacy_probe(prec, *shape)))
