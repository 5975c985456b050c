"""GPU parity tests (MI355X): the HIP kernels against the reference's own unit-level outputs (tests/golden/),
covering the fixtures that round 1 only ran on the oracle — G1 (posenc incl. the JAX-style quirks), G3 (depth-0 and
skips=[2] MLPs), G4 (GLO lookup with (B,) and (B,1) indices), G9 (volumetric rendering: sample_at_infinity x white
background, sigma = 0 and sigma = 1e4 rows, median depth), G13/G14 (loss, PSNR), G16 (filter_sigma) — plus the
persistent multi-iteration path of the machine kernels (> 1024 workgroup tiles).

fp32 (parity) mode, north-star tolerance 1e-4; every comparison is ELEMENT-wise relative with an absolute floor
(|a-b| <= tol * max(|ref|, floor)), the floor stated per tensor."""
import os

import numpy as np
import pytest
import torch

import hashprng as H
import hypernerf_torch_amd as HN
from gpu_common import oracle_threads, DEV, EMB, assert_close, assert_rel_close, load_hash, rays_for
from hypernerf_torch_amd import functional as F
from hypernerf_torch_amd import losses
from hypernerf_torch_amd.hypernerf import model_utils as MU
from hypernerf_torch_amd.hypernerf import models, modules
from hypernerf_torch_amd.models import nerf as legacy_nerf
from oracle import hypernerf_oracle as O

pytestmark = pytest.mark.gpu


def G(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def T(a):
    return torch.from_numpy(np.asarray(a)).to(DEV)


@pytest.fixture(autouse=True)
def fp32_mode():
    old = HN.get_precision()
    HN.set_precision("fp32")
    yield
    HN.set_precision(old)


def test_g01_posenc_hip(golden_dir):
    """hn_posenc against posenc_orig (N = 4, 6, 7, 10; 2-D and 3-D inputs), the legacy Embedding and the JAX-style
    posenc with its non-integer scales and cos-as-shifted-sin (model_utils.py:234-246, 255-274; nerf.py:4-38).
    Floor 1e-2: sin(2^9 x) of an fp32 argument is only defined to ~|x| 2^9 2^-24 = 6e-5 absolute."""
    g = G(golden_dir, "g01_posenc")
    for n in (4, 6, 7, 10):
        freqs = (2.0 ** torch.arange(n)).float().to(DEV)
        for tag in ("2", "3"):
            y = F.posenc(T(g[f"x{tag}_{n}"]), freqs, True)
            assert_rel_close(y, g[f"y{tag}_{n}"], 1e-4, 1e-2, f"posenc_orig N={n} {tag}-D")
    for n in (10, 4):
        emb = legacy_nerf.Embedding(3, n)
        assert_rel_close(emb(T(g["xe"])), g[f"ye_{n}"], 1e-4, 1e-2, f"legacy Embedding N={n}")
    for key, lo, hi, ident in (("yj_id", 0, 8, True), ("yj", 0, 8, False), ("yj_24", 2, 6, False)):
        scales = (2.0 ** torch.linspace(float(lo), float(hi), steps=hi - lo)).to(DEV)
        y = F.posenc(T(g["xj"]), scales, ident, jax_cos=True)
        assert_rel_close(y, g[key], 1e-4, 1e-2, f"posenc (JAX style) {key}")


def test_g01_through_the_model_utils_names(golden_dir):
    """The drop-in names of hypernerf/model_utils.py:234-274 (`posenc_orig`, `posenc`) with the reference's
    signatures, on the reference's own G1 vectors; `log_scale=False` and the gradient against the oracle/autograd."""
    g = G(golden_dir, "g01_posenc")
    for n in (4, 6, 7, 10):
        for tag in ("2", "3"):
            assert_rel_close(MU.posenc_orig(T(g[f"x{tag}_{n}"]), n), g[f"y{tag}_{n}"], 1e-4, 1e-2,
                             f"model_utils.posenc_orig N={n} {tag}-D")
            assert MU.get_posenc_ch_orig(g[f"x{tag}_{n}"].shape[-1], n) == g[f"y{tag}_{n}"].shape[-1]
    for key, lo, hi, ident in (("yj_id", 0, 8, True), ("yj", 0, 8, False), ("yj_24", 2, 6, False)):
        assert_rel_close(MU.posenc(T(g["xj"]), lo, hi, use_identity=ident), g[key], 1e-4, 1e-2,
                         f"model_utils.posenc {key}")
        assert MU.get_posenc_ch(g["xj"].shape[-1], lo, hi, ident) == g[key].shape[-1]
    x = H.uniform(3, "pe_lin", (50, 3), -1, 1)
    bands = torch.linspace(0, 4, 5)
    ref = torch.cat([x] + [f(b * x) for b in bands for f in (torch.sin, torch.cos)], -1)     # model_utils.py:238-246
    assert_rel_close(MU.posenc_orig(x.to(DEV), 5, log_scale=False), ref, 1e-4, 1e-2, "posenc_orig log_scale=False")
    xr = x.clone().requires_grad_(True)
    gw = H.normal(3, "pe_g", (50, 3 * 13))
    (O.posenc_orig(xr, 6) * gw).sum().backward()
    xd = x.clone().to(DEV).requires_grad_(True)
    (MU.posenc_orig(xd, 6) * gw.to(DEV)).sum().backward()
    assert_close(xd.grad, xr.grad, 1e-4, "d posenc_orig / dx")


@pytest.mark.parametrize("inf,wb", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_g09_depth_index_names(golden_dir, inf, wb):
    """model_utils.compute_depth_index / compute_depth_map / compute_opaqueness_mask (model_utils.py:319-362) on the
    reference's own weights: its depth index and median depth (G9), exact; the mask against the oracle's restatement,
    also for a threshold no ray reaches (all-zero mask, index 0, depth 0) and (B, R, S)-shaped weights."""
    g = G(golden_dir, "g09_volrend")
    w, z = T(g[f"weights_inf{inf}_wb{wb}"]), T(g["z"])
    idx = MU.compute_depth_index(w)
    assert idx.dtype == torch.int64 and np.array_equal(idx.cpu().numpy(), g[f"dindex_inf{inf}_wb{wb}"])
    assert np.array_equal(MU.compute_depth_map(w, z).cpu().numpy(), g[f"med_depth_inf{inf}_wb{wb}"])
    mask_ref, idx_ref = O.median_depth_index(w.cpu())
    assert torch.equal(MU.compute_opaqueness_mask(w).cpu(), mask_ref.float()) and torch.equal(idx.cpu(), idx_ref)
    for thr in (0.1, 0.9, 5.0):
        m_ref, i_ref = O.median_depth_index(w.cpu(), thr)
        assert torch.equal(MU.compute_opaqueness_mask(w, thr).cpu(), m_ref.float()), thr
        assert torch.equal(MU.compute_depth_index(w, thr).cpu(), i_ref), thr
        assert torch.equal(MU.compute_depth_map(w, z, thr).cpu(), (m_ref.float() * z.cpu()).sum(-1)), thr
    w3 = H.uniform(5, "w3", (6, 7, 130), 0, 0.02)
    z3 = torch.sort(H.uniform(5, "z3", (6, 7, 130), 0, 1), dim=-1)[0]
    m_ref, i_ref = O.median_depth_index(w3)
    # the scan order differs from ATen's sequential cumsum: exclude rays whose running sum passes within 1e-6 of 0.5
    cs = torch.cumsum(w3.double(), -1)
    safe = ((cs - 0.5).abs().min(dim=-1)[0] > 1e-6)
    assert safe.float().mean() > 0.9
    assert torch.equal(MU.compute_depth_index(w3.to(DEV)).cpu()[safe], i_ref[safe])
    assert torch.equal(MU.compute_depth_map(w3.to(DEV), z3.to(DEV)).cpu()[safe], (m_ref.float() * z3).sum(-1)[safe])
    # differentiable w.r.t. z_vals like the reference's sum(mask * z_vals): the gradient IS the mask
    zg = z.clone().requires_grad_(True)
    d = MU.compute_depth_map(w, zg)
    assert np.array_equal(d.detach().cpu().numpy(), g[f"med_depth_inf{inf}_wb{wb}"])
    d.sum().backward()
    assert torch.equal(zg.grad.cpu(), mask_ref.float().expand_as(zg))


@pytest.mark.parametrize("name,kw", [
    ("d0", dict(in_ch=128, out_ch=3, depth=0, width=128)),
    ("skip2", dict(in_ch=20, out_ch=5, depth=5, width=32, skips=[2])),
    ("warp", dict(in_ch=71, out_ch=3, depth=6, width=128)),
    ("sheet", dict(in_ch=53, out_ch=4, depth=6, width=64)),
    ("trunk", dict(in_ch=115, out_ch=256, depth=8, width=256, output_activation=torch.nn.ReLU())),
    ("rgb", dict(in_ch=167, out_ch=3, depth=4, width=128, output_activation=torch.nn.Sigmoid())),
])
def test_g03_mlp_hip(golden_dir, name, kw):
    """modules.MLP on the machine against the reference's MLP.forward (modules.py:116-127): every G3 case,
    including depth 0 (which still builds one hidden layer, modules.py:99-101) and a skip after layer 2."""
    g = G(golden_dir, "g03_mlp")
    m = modules.MLP(**kw)
    assert sorted(m.state_dict().keys()) == g["keys_" + name].tolist()
    load_hash(m, 3)
    y = m.to(DEV)(T(g["x_" + name]))
    assert_rel_close(y, g["y_" + name], 1e-4, 1e-1 if name != "rgb" else 1e-2, f"G3 {name}")


def test_g04_glo_hip(golden_dir):
    """GLOEmbed with (B,) and (B,1) indices (modules.py:131-167): exact (a gather)."""
    g = G(golden_dir, "g04_glo")
    e = modules.GLOEmbed(num_embeddings=100, embedding_dim=8)
    load_hash(e, 4)
    e = e.to(DEV)
    idx = T(g["idx"])
    y_flat, y_col = e(idx), e(idx[:, None])
    assert y_flat.shape == g["y_flat"].shape and y_col.shape == g["y_col"].shape
    assert np.array_equal(y_flat.detach().cpu().numpy(), g["y_flat"])
    assert np.array_equal(y_col.detach().cpu().numpy(), g["y_col"])


@pytest.mark.parametrize("inf", [True, False])
@pytest.mark.parametrize("wb", [True, False])
def test_g09_volumetric_rendering_hip(golden_dir, inf, wb):
    """hn_composite (activated-density variant) against the reference's volumetric_rendering for all four
    sample_at_infinity x white_background combinations, incl. the sigma = 0 ray, the sigma = 1e4 ray and the late
    surface, and the median-depth index (model_utils.py:43-107, 319-362).  Floor 1e-3 (weights of empty space)."""
    g = G(golden_dir, "g09_volrend")
    tag = f"inf{int(inf)}_wb{int(wb)}"
    r = MU.volumetric_rendering(T(g["rgb"]), T(g["sigma"]), T(g["z"]), T(g["d"]), use_white_background=wb,
                                sample_at_infinity=inf)
    for k in ("rgb", "depth", "acc", "weights", "med_depth"):
        assert_rel_close(r[k], g[f"{k}_{tag}"], 1e-4, 1e-3, f"G9 {k} {tag}")
    # the median depth is a gathered z value: the reference's own, exactly (same median index)
    assert np.array_equal(r["med_depth"].cpu().numpy(), g[f"med_depth_{tag}"].reshape(-1))


def test_g13_loss_and_psnr_hip(golden_dir):
    """losses.MSELoss / psnr on the GPU against the reference's losses.py:9-14 and metrics.py:4-13, value and
    gradient (the gradient of mean((a-gt)^2) is 2 (a-gt) / numel)."""
    g = G(golden_dir, "g13_misc")
    a, b, gt = (T(g[k]).clone().requires_grad_(k != "gt") for k in ("a", "b", "gt"))
    lf = losses.MSELoss()
    l1 = lf({"coarse": {"rgb": a}}, gt)
    assert abs(float(l1.detach()) - float(g["loss_c"])) <= 1e-6 * max(1.0, float(g["loss_c"]))
    l2 = lf({"coarse": {"rgb": a}, "fine": {"rgb": b}}, gt)
    assert abs(float(l2.detach()) - float(g["loss_cf"])) <= 1e-6 * max(1.0, float(g["loss_cf"]))
    l2.backward()
    n = a.numel()
    assert_rel_close(a.grad, (2.0 * (g["a"] - g["gt"]) / n), 1e-5, 1e-6, "d loss / d coarse rgb")
    assert_rel_close(b.grad, (2.0 * (g["b"] - g["gt"]) / n), 1e-5, 1e-6, "d loss / d fine rgb")
    p = losses.psnr(b.detach(), gt)
    assert abs(float(p) - float(g["psnr_f"])) <= 1e-5 * abs(float(g["psnr_f"]))


@pytest.mark.parametrize("tag", ["none", "dust", "box", "both"])
def test_g16_filter_sigma_hip(golden_dir, tag):
    """filter_sigma (models.py:35-63) as the compositing kernel applies it, against the reference's
    filter_sigma -> volumetric_rendering on the same samples; and the host-side models.filter_sigma itself."""
    g = G(golden_dir, "g16_filter")
    box = tuple(float(v) for v in g["box"])
    opts = {"none": None, "dust": {"dust_threshold": 0.5}, "box": {"bounding_box": box},
            "both": {"dust_threshold": 0.5, "bounding_box": box}}[tag]
    pts, sigma = T(g["pts"]), T(g["sigma"])
    f = models.filter_sigma(pts, sigma, opts)
    assert np.array_equal(f.cpu().numpy(), g["sigma_" + tag])
    dust = opts.get("dust_threshold") if opts and "dust_threshold" in opts else None
    keep = None
    if opts and "bounding_box" in opts:
        keep = ((pts[..., 0] >= box[0]) & (pts[..., 0] <= box[1]) & (pts[..., 1] >= box[2]) & (pts[..., 1] <= box[3])
                & (pts[..., 2] >= box[4]) & (pts[..., 2] <= box[5])).float()
    outs = F.composite(T(g["rgb"]), sigma, None, T(g["z"]), T(g["d"]), None, variant=2, white_bg=False,
                       sample_at_infinity=True, want_median=True, dust_threshold=dust, keep=keep)
    for i, k in enumerate(("rgb", "depth", "acc", "weights", "med_depth")):
        assert_rel_close(outs[i], g[f"{k}_{tag}"], 1e-4, 1e-3, f"G16 {k} {tag}")


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_persistent_loop_more_than_1024_tiles(precision):
    """hn_grid_for caps the grid at 1024 workgroups; beyond that a workgroup runs several tiles (re-staging its
    sources and restarting the weight stream each time).  2048 rays x 192 samples = 393,216 points = 3072 fp32 tiles /
    1536 bf16 tiles of the translation field, forward and backward, against the oracle."""
    from hypernerf_torch_amd.hypernerf import warping
    HN.set_precision(precision)
    tf = warping.TranslationField(in_ch=3, in_ch_embed=8)
    sd = load_hash(tf, 5)
    b, s = 2048, 192
    pts = H.uniform(21, "pts", (b, s, 3), -1.2, 1.2)
    emb = H.uniform(21, "emb", (b, 8), -0.5, 0.5)
    tp = {"w." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    er = emb.clone().requires_grad_(True)
    torch.set_num_threads(oracle_threads(32))
    y_ref = O.translation_field(tp, "w", pts, er[:, None, :].expand(b, s, 8))
    gsel = H.uniform(22, "g", (b, s, 3), -1, 1)
    (y_ref * gsel).sum().backward()
    tf = tf.to(DEV)
    eg = emb.to(DEV).requires_grad_(True)
    y = tf.warp(pts.to(DEV), eg, None)
    (y * gsel.to(DEV)).sum().backward()
    tol, gtol = (1e-4, 2e-3) if precision == "fp32" else (1e-2, 2.5e-1)     # bf16: the bound of test_gpu_model.GTOL
    assert_close(y, y_ref, tol, f"warp forward, {b * s} points", elementwise=precision == "fp32")
    # the LAST tile (served by a workgroup's later iteration) specifically
    assert_close(y[-64:], y_ref[-64:], tol, "warp forward, last rays", elementwise=precision == "fp32")
    from gpu_common import assert_grad_close
    assert_grad_close(eg.grad, er.grad, gtol, "d embed", frobenius=True)
    for k, prm in tf.named_parameters():
        assert_grad_close(prm.grad, tp["w." + k].grad, gtol, "d " + k, frobenius=True)
