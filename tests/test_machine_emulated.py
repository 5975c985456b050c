"""Host "compiler" of the MLP machine checked WITHOUT a GPU: the op programs, packing tables, slot layout
and dW job lists produced by hypernerf_torch_amd.machine are executed by a lane-level numpy emulation of the
device algorithms (tests/emulator.py, documented gfx950 MFMA lane maps) and compared with the CPU oracle
and torch autograd.  Both numeric modes' LAYOUTS are emulated (values stay fp64, so tolerance is 1e-9)."""
import numpy as np
import pytest
import torch

import emulator as E
import hashprng as H
import hypernerf_torch_amd  # noqa: F401
from hypernerf_torch_amd import _lib as L
from hypernerf_torch_amd import machine
from hypernerf_torch_amd.hypernerf import models, modules, warping
from oracle import hypernerf_oracle as O

EMB = {"warp": list(range(100)), "camera": [0], "appearance": list(range(100)), "time": list(range(100))}


def load_hash(module, seed):
    sd = module.state_dict()
    module.load_state_dict(H.fill_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed))
    return {k: v.double() for k, v in module.state_dict().items()}


def np_params(prog):
    return [p.detach().double().numpy() for p in prog.params]


def check_grads(prog, flat, torch_params, named):
    offs, _ = prog.grad_offsets()
    for prm, off in zip(prog.params, offs):
        name = [k for k, v in named.items() if v is prm][0]
        ref = torch_params[name].grad
        got = flat[off:off + prm.numel()].reshape(prm.shape)
        ref = np.zeros(prm.shape) if ref is None else ref.numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-10, err_msg=name)


@pytest.mark.parametrize("bf16", [True, False])
def test_translation_field_emulated(bf16):
    torch.manual_seed(0)
    tf = warping.TranslationField(in_ch=3, in_ch_embed=8, depth=6, hidden_channels=64)
    sd = load_hash(tf, 11)
    b, s = 5, 8
    n = b * s
    pts = H.uniform(1, "pts", (n, 3), -1, 1).double()
    emb = H.uniform(1, "emb", (b, 8), -0.5, 0.5).double()
    call = tf._call(True, False, True)
    prog = call.program
    mode = E.Mode(bf16)
    tables = prog.host_tables(1 if bf16 else 0)
    params = np_params(prog)
    srcs = [(pts.numpy(), False), (emb.numpy(), True), None, None]
    outs, stash = E.run_forward(prog, mode, tables, params, srcs, n, s, [3])
    # oracle
    tp = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    embt = emb.clone().requires_grad_(True)
    y = O.translation_field({"w." + k: v for k, v in tp.items()}, "w", pts.view(b, s, 3),
                            embt[:, None, :].expand(b, s, 8))
    np.testing.assert_allclose(outs[0].reshape(b, s, 3), y.detach().numpy(), rtol=1e-9, atol=1e-11)
    g = H.uniform(2, "g", (n, 3), -1, 1).double()
    (y.reshape(n, 3) * g).sum().backward()
    bsrcs = srcs + [(g.numpy(), False)]
    dsrc = E.run_backward(prog, mode, tables, params, bsrcs, n, s, stash)
    # embed gradient: dsrc columns -> per-ray sums
    cols = {c: sl for (si, c), sl in prog.dsrc_map.items() if si == 1}
    got = np.stack([dsrc[:, cols[c]].reshape(b, s).sum(1) for c in range(8)], axis=1)
    np.testing.assert_allclose(got, embt.grad.numpy(), rtol=1e-8, atol=1e-10)
    jobs = prog.wgrad_jobs(1 if bf16 else 0, n)
    _, gtot = prog.grad_offsets()
    flat = E.run_wgrad(prog, mode, jobs, stash, gtot)
    check_grads(prog, flat, tp, dict(tf.named_parameters()))


@pytest.mark.parametrize("bf16", [True, False])
def test_nerf_mlp_emulated(bf16):
    torch.manual_seed(0)
    nm = modules.NerfMLP(in_ch=6, trunk_depth=4, trunk_width=64, rgb_branch_depth=2, rgb_branch_width=32,
                         hidden_activation=torch.nn.ReLU(), skips=[2], rgb_activation=torch.nn.Sigmoid(),
                         alpha_condition_dim=4, rgb_condition_dim=5, alpha_brach_width=32)
    # bottleneck is trunk_width//2 = 32 = rgb_branch_width
    sd = load_hash(nm, 5)
    b, s = 3, 12
    n = b * s          # 36 points: one full block + a partial one
    x = H.uniform(3, "x", (b, s, 6), -1, 1).double()
    ac = H.uniform(3, "ac", (b, 4), -0.5, 0.5).double()
    rc = H.uniform(3, "rc", (b, 5), -1, 1).double()
    call = nm._call(True, 4, True, 5, True)
    prog = call.program
    mode = E.Mode(bf16)
    tables = prog.host_tables(1 if bf16 else 0)
    params = np_params(prog)
    srcs = [(x.reshape(n, 6).numpy(), False), (ac.numpy(), True), (rc.numpy(), True), None]
    outs, stash = E.run_forward(prog, mode, tables, params, srcs, n, s, [3, 1])
    tp = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xt, act, rct = (t.clone().requires_grad_(True) for t in (x, ac, rc))
    rgb, alpha = O.nerf_mlp({"n." + k: v for k, v in tp.items()}, "n", xt, act, rct, trunk_depth=4, rgb_depth=2,
                            skips=(2,))
    np.testing.assert_allclose(outs[0].reshape(b, s, 3), rgb.detach().numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(outs[1].reshape(b, s, 1), alpha.detach().numpy(), rtol=1e-9, atol=1e-11)
    g_rgb = H.uniform(4, "g_rgb", (n, 3), -1, 1).double()
    g_a = H.uniform(4, "g_a", (n, 1), -1, 1).double()
    ((rgb.reshape(n, 3) * g_rgb).sum() + (alpha.reshape(n, 1) * g_a).sum()).backward()
    bsrcs = srcs + [(g_rgb.numpy(), False), (g_a.numpy(), False), (outs[0], False)]
    dsrc = E.run_backward(prog, mode, tables, params, bsrcs, n, s, stash)
    cx = {c: sl for (si, c), sl in prog.dsrc_map.items() if si == 0}
    got_x = np.stack([dsrc[:, cx[c]] for c in range(6)], axis=1)
    np.testing.assert_allclose(got_x, xt.grad.reshape(n, 6).numpy(), rtol=1e-8, atol=1e-10)
    for si, ref, width in ((1, act.grad, 4), (2, rct.grad, 5)):
        cols = {c: sl for (s2, c), sl in prog.dsrc_map.items() if s2 == si}
        got = np.stack([dsrc[:, cols[c]].reshape(b, s).sum(1) for c in range(width)], axis=1)
        np.testing.assert_allclose(got, ref.numpy(), rtol=1e-8, atol=1e-10)
    jobs = prog.wgrad_jobs(1 if bf16 else 0, n)
    _, gtot = prog.grad_offsets()
    flat = E.run_wgrad(prog, mode, jobs, stash, gtot)
    check_grads(prog, flat, tp, dict(nm.named_parameters()))


@pytest.mark.parametrize("bf16", [True, False])
def test_wide_output_mlp_emulated(bf16):
    m = modules.MLP(in_ch=10, out_ch=40, depth=3, width=32, skips=[0], output_activation=torch.nn.ReLU())
    sd = load_hash(m, 9)
    n = 33
    x = H.uniform(5, "x", (n, 10), -1, 1).double()
    call = m._call(True)
    prog = call.program
    mode = E.Mode(bf16)
    tables = prog.host_tables(1 if bf16 else 0)
    params = np_params(prog)
    srcs = [(x.numpy(), False), None, None, None]
    outs, stash = E.run_forward(prog, mode, tables, params, srcs, n, 1, [40])
    tp = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xt = x.clone().requires_grad_(True)
    y = O.mlp({"m." + k: v for k, v in tp.items()}, "m", xt, depth=3, skips=(0,), out_act="relu")
    np.testing.assert_allclose(outs[0], y.detach().numpy(), rtol=1e-9, atol=1e-11)
    g = H.uniform(6, "g", (n, 40), -1, 1).double()
    (y * g).sum().backward()
    dsrc = E.run_backward(prog, mode, tables, params, srcs + [(g.numpy(), False), (outs[0], False)], n, 1, stash)
    cx = {c: sl for (si, c), sl in prog.dsrc_map.items() if si == 0}
    np.testing.assert_allclose(np.stack([dsrc[:, cx[c]] for c in range(10)], axis=1), xt.grad.numpy(),
                               rtol=1e-8, atol=1e-10)
    jobs = prog.wgrad_jobs(1 if bf16 else 0, n)
    _, gtot = prog.grad_offsets()
    check_grads(prog, E.run_wgrad(prog, mode, jobs, stash, gtot), tp, dict(m.named_parameters()))


@pytest.mark.parametrize("bf16", [True, False])
def test_wide_raw_input_direct_features_emulated(bf16):
    """> HN_MAX_COMPS raw input channels: the surplus identity features are read directly (HN_FEAT_ID_DIRECT)."""
    m = modules.MLP(in_ch=45, out_ch=3, depth=2, width=32, skips=[7])
    sd = load_hash(m, 13)
    n = 40
    x = H.uniform(7, "x", (n, 45), -1, 1).double()
    call = m._call(False)
    prog = call.program
    assert len(prog.comp_map) == 32
    mode = E.Mode(bf16)
    tables = prog.host_tables(1 if bf16 else 0)
    params = np_params(prog)
    srcs = [(x.numpy(), False), None, None, None]
    outs, stash = E.run_forward(prog, mode, tables, params, srcs, n, 1, [3])
    tp = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y = O.mlp({"m." + k: v for k, v in tp.items()}, "m", x, depth=2, skips=(7,))
    np.testing.assert_allclose(outs[0], y.detach().numpy(), rtol=1e-9, atol=1e-11)
    g = H.uniform(8, "g", (n, 3), -1, 1).double()
    (y * g).sum().backward()
    E.run_backward(prog, mode, tables, params, srcs + [(g.numpy(), False), (outs[0], False)], n, 1, stash)
    jobs = prog.wgrad_jobs(1 if bf16 else 0, n)
    _, gtot = prog.grad_offsets()
    check_grads(prog, E.run_wgrad(prog, mode, jobs, stash, gtot), tp, dict(m.named_parameters()))


@pytest.mark.parametrize("bf16", [True, False])
def test_fused_level_program_emulated(bf16):
    """The whole level as ONE program (warp field -> hyper sheet -> template; models.NerfModel._level_call): heads
    publish their results as staged components, the template encodes them, the backward walks the three chains in
    reverse and feeds the warp / sheet heads from the source-gradient accumulators; the GLO table is gathered by ray
    index.  Small layer widths (the program structure is what is under test), against the oracle + torch autograd:
    outputs, every weight gradient and the table gradient."""
    from hypernerf_torch_amd.hypernerf import models
    torch.manual_seed(0)
    emb = {"warp": list(range(12)), "camera": [0], "appearance": list(range(12)), "time": list(range(12))}
    m = models.NerfModel(emb, n_samples_coarse=8, n_samples_fine=8, hyper_slice_method="bendy_sheet",
                         use_nerf_embed=True, use_alpha_cond=True, xyz_fourier_dim=2, hyper_fourier_dim=1,
                         view_fourier_dim=1)
    m.warp_field = warping.TranslationField(in_ch=3, in_ch_embed=8, depth=6, hidden_channels=32)
    m.hyper_sheet_mlp = modules.HyperSheetMLP(out_ch=4, in_ch_embed=8, depth=6, width=32)
    m.nerf_mlps_coarse = modules.NerfMLP(in_ch=27, trunk_depth=4, trunk_width=64, rgb_branch_depth=2,
                                         rgb_branch_width=32, hidden_activation=torch.nn.ReLU(), skips=[2],
                                         rgb_activation=torch.nn.Sigmoid(), alpha_condition_dim=8, rgb_condition_dim=9,
                                         alpha_brach_width=32)
    sd = load_hash(m, 21)
    b, s = 5, 8
    n = b * s                      # 40 points: one full block and a partial one
    pts = H.uniform(7, "pts", (b, s, 3), -1, 1).double()
    dirs = H.uniform(7, "dirs", (b, 3), -1, 1).double()
    idx = torch.tensor([3, 11, 0, 3, 7])
    call = m._level_call("coarse")
    prog = call.program
    assert [len(c) for c in prog.chains()] == [7, 7, 10]
    mode = E.Mode(bf16)
    tables = prog.host_tables(1 if bf16 else 0)
    params = np_params(prog)
    table = sd["warp_embed.embed.weight"]
    srcs = [(pts.reshape(n, 3).numpy(), False), (dirs.numpy(), True), (table.numpy(), True, idx.numpy()), None]
    outs, stash = E.run_forward(prog, mode, tables, params, srcs, n, s, [7, 3, 1])
    tp = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    e = O.glo_embed(tp["warp_embed.embed.weight"], idx)
    ee = e[:, None, :].expand(b, s, 8)
    warped = torch.cat([O.translation_field(tp, "warp_field", pts, ee), O.hyper_sheet(tp, "hyper_sheet_mlp", pts, ee)], -1)
    feat = torch.cat([O.posenc_orig(warped[..., :3], 2), O.posenc_orig(warped[..., 3:], 1)], -1)
    rgb, alpha = O.nerf_mlp(tp, "nerf_mlps_coarse", feat, e, O.posenc_orig(dirs, 1), trunk_depth=4, rgb_depth=2,
                            skips=(2,))
    np.testing.assert_allclose(outs[0].reshape(b, s, 7), warped.detach().numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(outs[1].reshape(b, s, 3), rgb.detach().numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(outs[2].reshape(b, s, 1), alpha.detach().numpy(), rtol=1e-9, atol=1e-11)
    g_rgb = H.uniform(8, "g_rgb", (n, 3), -1, 1).double()
    g_a = H.uniform(8, "g_a", (n, 1), -1, 1).double()
    g_w = H.uniform(8, "g_w", (n, 7), -1, 1).double()       # an external gradient on warped_points as well
    ((rgb.reshape(n, 3) * g_rgb).sum() + (alpha.reshape(n, 1) * g_a).sum() + (warped.reshape(n, 7) * g_w).sum()).backward()
    bsrcs = srcs[:3] + [(outs[0], False), (g_rgb.numpy(), False), (g_a.numpy(), False), (outs[1], False),
                        (g_w.numpy(), False)]
    dsrc = E.run_backward(prog, mode, tables, params, bsrcs, n, s, stash)
    cols = {c: sl for (si, c), sl in prog.dsrc_map.items() if si == 2}
    rows = np.stack([dsrc[:, cols[c]].reshape(b, s).sum(1) for c in range(8)], axis=1)
    d_table = np.zeros(table.shape)
    np.add.at(d_table, idx.numpy(), rows)
    np.testing.assert_allclose(d_table, tp["warp_embed.embed.weight"].grad.numpy(), rtol=1e-8, atol=1e-10)
    mask, col = prog.embed_fold(2)
    for c, sl in cols.items():
        assert col[sl] == c and (mask >> ((sl & 3) + 4 * (sl >> 3))) & 1
    jobs = prog.wgrad_jobs(1 if bf16 else 0, n)
    _, gtot = prog.grad_offsets()
    flat = E.run_wgrad(prog, mode, jobs, stash, gtot)
    check_grads(prog, flat, tp, dict(m.named_parameters()))
    # without the external gradient the optional source is simply absent
    bsrcs[7] = None
    for v in tp.values():
        v.grad = None
    e2 = O.glo_embed(tp["warp_embed.embed.weight"], idx)[:, None, :].expand(b, s, 8)
    w2 = torch.cat([O.translation_field(tp, "warp_field", pts, e2), O.hyper_sheet(tp, "hyper_sheet_mlp", pts, e2)], -1)
    f2 = torch.cat([O.posenc_orig(w2[..., :3], 2), O.posenc_orig(w2[..., 3:], 1)], -1)
    r2, a2 = O.nerf_mlp(tp, "nerf_mlps_coarse", f2, e2[:, 0], O.posenc_orig(dirs, 1), trunk_depth=4, rgb_depth=2, skips=(2,))
    ((r2.reshape(n, 3) * g_rgb).sum() + (a2.reshape(n, 1) * g_a).sum()).backward()
    stash2 = E.run_forward(prog, mode, tables, params, srcs, n, s, [7, 3, 1])[1]
    E.run_backward(prog, mode, tables, params, bsrcs, n, s, stash2)
    flat2 = E.run_wgrad(prog, mode, jobs, stash2, gtot)
    check_grads(prog, flat2, tp, dict(m.named_parameters()))


@pytest.mark.parametrize("bf16", [True, False])
def test_fine_level_in_two_parts_emulated(bf16):
    """Round 5, NerfModel.REUSE_COARSE on the host compiler + the lane-level emulation (no GPU): the fine level as
    (i) the fine TEMPLATE alone over the coarse level's warped points (`_template_reuse_call`: source 0 = the coarse
    program's `warped_points` output, GLO conditions gathered by ray) + (ii) the whole level program over the NEW samples
    only, against the reference's structure — the level program over every fine sample.  Same rgb / alpha / warped
    points for every sample; and with (i)'s source gradient handed to the coarse program as the external gradient on its
    `warped_points` output, the SUM of the weight gradients and the GLO-table gradient equal the full path's."""
    torch.manual_seed(0)
    emb = {"warp": list(range(12)), "camera": [0], "appearance": list(range(12)), "time": list(range(12))}
    m = models.NerfModel(emb, n_samples_coarse=8, n_samples_fine=8, hyper_slice_method="bendy_sheet",
                         use_nerf_embed=True, use_alpha_cond=True, xyz_fourier_dim=2, hyper_fourier_dim=1,
                         view_fourier_dim=1)
    m.warp_field = warping.TranslationField(in_ch=3, in_ch_embed=8, depth=6, hidden_channels=32)
    m.hyper_sheet_mlp = modules.HyperSheetMLP(out_ch=4, in_ch_embed=8, depth=6, width=32)
    for lvl in ("coarse", "fine"):
        setattr(m, f"nerf_mlps_{lvl}", modules.NerfMLP(
            in_ch=27, trunk_depth=4, trunk_width=64, rgb_branch_depth=2, rgb_branch_width=32,
            hidden_activation=torch.nn.ReLU(), skips=[2], rgb_activation=torch.nn.Sigmoid(), alpha_condition_dim=8,
            rgb_condition_dim=9, alpha_brach_width=32))
    sd = load_hash(m, 23)
    b, s = 5, 8
    n = b * s
    old = H.uniform(9, "old", (b, s, 3), -1, 1).double()
    new = H.uniform(9, "new", (b, s, 3), -1, 1).double()
    dirs = H.uniform(9, "dirs", (b, 3), -1, 1).double()
    idx = torch.tensor([3, 11, 0, 3, 7])
    table = sd["warp_embed.embed.weight"].numpy()
    mode = E.Mode(bf16)
    mi = 1 if bf16 else 0
    pc, pf, pr = (m._level_call("coarse").program, m._level_call("fine").program,
                  m._template_reuse_call("fine", 4, False, True).program)
    assert [len(c) for c in pr.chains()] == [10] and pr.n_src == 4
    # source 0 of the reuse program: xyz then the sheet's 4 hyper coordinates, every one of them with a gradient row
    assert sorted(c for (si, c) in pr.dsrc_map if si == 0) == list(range(7))
    tabs = {id(p): p.host_tables(mi) for p in (pc, pf, pr)}
    prm = {id(p): np_params(p) for p in (pc, pf, pr)}

    def fwd(prog, pts, spr, widths, src0_is_warped=False):
        k = pts.shape[0]
        srcs = [(pts, False), (dirs.numpy(), True), (table, True, idx.numpy())] + ([] if src0_is_warped else [None])
        outs, stash = E.run_forward(prog, mode, tabs[id(prog)], prm[id(prog)], srcs, k, spr, widths)
        return srcs, outs, stash
    g = {k_: H.uniform(10, k_, shp, -1, 1).double().numpy() for k_, shp in
         (("c_rgb", (n, 3)), ("c_a", (n, 1)), ("o_rgb", (n, 3)), ("o_a", (n, 1)), ("n_rgb", (n, 3)), ("n_a", (n, 1)))}

    def bwd(prog, srcs, outs, stash, g_rgb, g_a, g_warped, k, spr, rgb_i, warped_out):
        bs = list(srcs[:3]) + [(outs[0], False) if warped_out else None, (g_rgb, False), (g_a, False), (outs[rgb_i], False),
                               (g_warped, False) if g_warped is not None else None]
        dsrc = E.run_backward(prog, mode, tabs[id(prog)], prm[id(prog)], bs, k, spr, stash)
        _, gtot = prog.grad_offsets()
        flat = E.run_wgrad(prog, mode, prog.wgrad_jobs(mi, k), stash, gtot)
        cols = {c: sl for (si, c), sl in prog.dsrc_map.items() if si == 2}
        d_tab = np.zeros(table.shape)
        np.add.at(d_tab, idx.numpy(), np.stack([dsrc[:, cols[c]].reshape(-1, spr).sum(1) for c in range(8)], axis=1))
        return dsrc, flat, d_tab

    def by_param(prog, flat, acc):
        offs, _ = prog.grad_offsets()
        for p_, off in zip(prog.params, offs):
            acc[id(p_)] = acc.get(id(p_), 0.0) + flat[off:off + p_.numel()]
        return acc
    # ---- the coarse level (both paths): forward once
    sc, oc, stc = fwd(pc, old.reshape(n, 3).numpy(), s, [7, 3, 1])
    # ---- reference structure: the fine program over [old | new] of every ray
    both = torch.cat([old, new], 1).reshape(2 * n, 3).numpy()
    sf, of_, stf = fwd(pf, both, 2 * s, [7, 3, 1])
    g_rgb_f = np.concatenate([g["o_rgb"].reshape(b, s, 3), g["n_rgb"].reshape(b, s, 3)], 1).reshape(2 * n, 3)
    g_a_f = np.concatenate([g["o_a"].reshape(b, s, 1), g["n_a"].reshape(b, s, 1)], 1).reshape(2 * n, 1)
    _, flat_f, tab_f = bwd(pf, sf, of_, stf, g_rgb_f, g_a_f, None, 2 * n, 2 * s, 1, True)
    _, flat_c, tab_c = bwd(pc, sc, oc, stc, g["c_rgb"], g["c_a"], None, n, s, 1, True)
    full = by_param(pc, flat_c, by_param(pf, flat_f, {}))
    # ---- two parts: (ii) the fine program over the new samples, (i) the fine template over the coarse warped points
    sn, on, stn = fwd(pf, new.reshape(n, 3).numpy(), s, [7, 3, 1])
    so, oo, sto = fwd(pr, oc[0], s, [3, 1], src0_is_warped=True)
    fr = of_[1].reshape(b, 2 * s, 3)
    np.testing.assert_allclose(oo[0].reshape(b, s, 3), fr[:, :s], rtol=1e-12, atol=1e-14)        # rgb of the old samples
    np.testing.assert_allclose(on[1].reshape(b, s, 3), fr[:, s:], rtol=1e-12, atol=1e-14)        # ... of the new ones
    fa = of_[2].reshape(b, 2 * s, 1)
    np.testing.assert_allclose(oo[1].reshape(b, s, 1), fa[:, :s], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(on[2].reshape(b, s, 1), fa[:, s:], rtol=1e-12, atol=1e-14)
    fw = of_[0].reshape(b, 2 * s, 7)
    np.testing.assert_allclose(oc[0].reshape(b, s, 7), fw[:, :s], rtol=1e-12, atol=1e-14)        # the warp is the coarse level's
    np.testing.assert_allclose(on[0].reshape(b, s, 7), fw[:, s:], rtol=1e-12, atol=1e-14)
    _, flat_n, tab_n = bwd(pf, sn, on, stn, g["n_rgb"], g["n_a"], None, n, s, 1, True)
    bs_o = list(so[:3]) + [None, (g["o_rgb"], False), (g["o_a"], False), (oo[0], False)]
    dsrc_o = E.run_backward(pr, mode, tabs[id(pr)], prm[id(pr)], bs_o, n, s, sto)
    _, gt_o = pr.grad_offsets()
    flat_o = E.run_wgrad(pr, mode, pr.wgrad_jobs(mi, n), sto, gt_o)
    cols_o = {c: sl for (si, c), sl in pr.dsrc_map.items() if si == 2}
    tab_o = np.zeros(table.shape)
    np.add.at(tab_o, idx.numpy(), np.stack([dsrc_o[:, cols_o[c]].reshape(-1, s).sum(1) for c in range(8)], axis=1))
    g_warped = np.stack([dsrc_o[:, pr.dsrc_map[(0, c)]] for c in range(7)], axis=1)              # d L / d warped_points (old)
    _, flat_c2, tab_c2 = bwd(pc, sc, oc, stc, g["c_rgb"], g["c_a"], g_warped, n, s, 1, True)
    parts = by_param(pc, flat_c2, by_param(pf, flat_n, by_param(pr, flat_o, {})))
    assert parts.keys() == full.keys()
    for k_ in full:
        np.testing.assert_allclose(parts[k_], full[k_], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(tab_c2 + tab_n + tab_o, tab_c + tab_f, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("bf16", [True, False])
def test_axis_aligned_level_and_gathered_template_emulated(bf16):
    """axis_aligned_plane (models.py:533-534): the hyper coordinates are the ray's GLO row.  (a) the fused level
    [warp | template] encodes them from the gathered source, (b) the stand-alone template program with the table
    gathered in-kernel (levels whose warp runs outside the program: SE3Field).  Outputs, weight gradients and the
    table gradient (conditions + hyper coordinates + warp input, one row per ray) against the oracle."""
    from hypernerf_torch_amd.hypernerf import models
    torch.manual_seed(0)
    emb = {"warp": list(range(12)), "camera": [0], "appearance": list(range(12)), "time": list(range(12))}
    m = models.NerfModel(emb, n_samples_coarse=8, n_samples_fine=8, hyper_slice_method="axis_aligned_plane",
                         hyper_slice_out_dim=8, use_nerf_embed=True, use_alpha_cond=True, xyz_fourier_dim=2,
                         hyper_fourier_dim=1, view_fourier_dim=1)
    m.warp_field = warping.TranslationField(in_ch=3, in_ch_embed=8, depth=6, hidden_channels=32)
    m.nerf_mlps_coarse = modules.NerfMLP(in_ch=15 + 24, trunk_depth=4, trunk_width=64, rgb_branch_depth=2,
                                         rgb_branch_width=32, hidden_activation=torch.nn.ReLU(), skips=[2],
                                         rgb_activation=torch.nn.Sigmoid(), alpha_condition_dim=8, rgb_condition_dim=9,
                                         alpha_brach_width=32)
    sd = load_hash(m, 23)
    b, s = 5, 8
    n = b * s
    pts = H.uniform(7, "pts", (b, s, 3), -1, 1).double()
    dirs = H.uniform(7, "dirs", (b, 3), -1, 1).double()
    idx = torch.tensor([3, 11, 0, 3, 7])
    mode = E.Mode(bf16)
    table = sd["warp_embed.embed.weight"]
    g_rgb = H.uniform(8, "g_rgb", (n, 3), -1, 1).double()
    g_a = H.uniform(8, "g_a", (n, 1), -1, 1).double()

    def reference(tp, with_warp):
        e = O.glo_embed(tp["warp_embed.embed.weight"], idx)
        ee = e[:, None, :].expand(b, s, 8)
        xyz = O.translation_field(tp, "warp_field", pts, ee) if with_warp else pts
        feat = torch.cat([O.posenc_orig(xyz, 2), O.posenc_orig(ee, 1)], -1)
        rgb, alpha = O.nerf_mlp(tp, "nerf_mlps_coarse", feat, e, O.posenc_orig(dirs, 1), trunk_depth=4, rgb_depth=2,
                                skips=(2,))
        return xyz, rgb, alpha

    def table_grad(prog, dsrc):
        cols = {c: sl for (si, c), sl in prog.dsrc_map.items() if si == 2}
        assert sorted(cols) == list(range(8))
        rows = np.stack([dsrc[:, cols[c]].reshape(b, s).sum(1) for c in range(8)], axis=1)
        d_table = np.zeros(table.shape)
        np.add.at(d_table, idx.numpy(), rows)
        return d_table

    # (a) fused level
    assert m._can_fuse_level(True, False, {})
    call = m._level_call("coarse")
    assert call.fill_from_gather == (0, 3) and call.dst_widths == [11, 3, 1]
    prog = call.program
    tables = prog.host_tables(1 if bf16 else 0)
    params = np_params(prog)
    srcs = [(pts.reshape(n, 3).numpy(), False), (dirs.numpy(), True), (table.numpy(), True, idx.numpy()), None]
    outs, stash = E.run_forward(prog, mode, tables, params, srcs, n, s, [11, 3, 1])
    tp = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xyz, rgb, alpha = reference(tp, True)
    np.testing.assert_allclose(outs[0].reshape(b, s, 11)[..., :3], xyz.detach().numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(outs[1].reshape(b, s, 3), rgb.detach().numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(outs[2].reshape(b, s, 1), alpha.detach().numpy(), rtol=1e-9, atol=1e-11)
    ((rgb.reshape(n, 3) * g_rgb).sum() + (alpha.reshape(n, 1) * g_a).sum()).backward()
    bsrcs = srcs[:3] + [(outs[0], False), (g_rgb.numpy(), False), (g_a.numpy(), False), (outs[1], False), None]
    dsrc = E.run_backward(prog, mode, tables, params, bsrcs, n, s, stash)
    np.testing.assert_allclose(table_grad(prog, dsrc), tp["warp_embed.embed.weight"].grad.numpy(), rtol=1e-8, atol=1e-10)
    _, gtot = prog.grad_offsets()
    check_grads(prog, E.run_wgrad(prog, mode, prog.wgrad_jobs(1 if bf16 else 0, n), stash, gtot), tp,
                dict(m.named_parameters()))

    # (b) template alone, table gathered in-kernel, gradient w.r.t. the spatial points returned
    call = m._template_gather_call("coarse", True, True)
    prog = call.program
    tables = prog.host_tables(1 if bf16 else 0)
    params = np_params(prog)
    srcs = [(pts.reshape(n, 3).numpy(), False), (dirs.numpy(), True), (table.numpy(), True, idx.numpy())]
    outs, stash = E.run_forward(prog, mode, tables, params, srcs, n, s, [3, 1])
    tp = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pts_g = pts.clone().requires_grad_(True)
    e = O.glo_embed(tp["warp_embed.embed.weight"], idx)
    feat = torch.cat([O.posenc_orig(pts_g, 2), O.posenc_orig(e[:, None, :].expand(b, s, 8), 1)], -1)
    rgb, alpha = O.nerf_mlp(tp, "nerf_mlps_coarse", feat, e, O.posenc_orig(dirs, 1), trunk_depth=4, rgb_depth=2, skips=(2,))
    np.testing.assert_allclose(outs[0].reshape(b, s, 3), rgb.detach().numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(outs[1].reshape(b, s, 1), alpha.detach().numpy(), rtol=1e-9, atol=1e-11)
    ((rgb.reshape(n, 3) * g_rgb).sum() + (alpha.reshape(n, 1) * g_a).sum()).backward()
    bsrcs = srcs + [None, (g_rgb.numpy(), False), (g_a.numpy(), False), (outs[0], False)]
    dsrc = E.run_backward(prog, mode, tables, params, bsrcs, n, s, stash)
    np.testing.assert_allclose(table_grad(prog, dsrc), tp["warp_embed.embed.weight"].grad.numpy(), rtol=1e-8, atol=1e-10)
    pcols = [prog.dsrc_map[(0, c)] for c in range(3)]
    np.testing.assert_allclose(dsrc[:, pcols].reshape(b, s, 3), pts_g.grad.numpy(), rtol=1e-8, atol=1e-10)
    _, gtot = prog.grad_offsets()
    flat = E.run_wgrad(prog, mode, prog.wgrad_jobs(1 if bf16 else 0, n), stash, gtot)
    check_grads(prog, flat, {k: v for k, v in tp.items() if k.startswith("nerf_mlps_coarse")},
                {k: v for k, v in m.named_parameters() if k.startswith("nerf_mlps_coarse")})


@pytest.mark.parametrize("bf16", [True, False])
def test_se3_field_program_emulated(bf16):
    """SE3Field as ONE program (warping.py:212-225): encoder -> trunk -> [w_net.linears.0 ; v_net.linears.0] as one
    row-stacked 128 -> 256 layer -> two logit layers reading a window (one half) of that activation each.  Outputs
    (w | v), every weight gradient (the stacked layer's go to two different parameters) and the gradient w.r.t. the
    points, against the oracle's MLPs + autograd."""
    torch.manual_seed(0)
    f = warping.SE3Field(in_ch=3)
    sd = load_hash(f, 13)
    n = 40
    pts = H.uniform(9, "pts", (n, 3), -1, 1).double()
    call = f._field_call(True)
    prog = call.program
    assert [len(c) for c in prog.chains()] == [10]
    mode = E.Mode(bf16)
    tables = prog.host_tables(1 if bf16 else 0)
    params = np_params(prog)
    srcs = [(pts.numpy(), False), None, None, None]
    outs, stash = E.run_forward(prog, mode, tables, params, srcs, n, 1, [6])
    tp = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pt = pts.clone().requires_grad_(True)
    # the encoder with the program's own (fp32) scale constants, evaluated in fp64 like the emulation
    scales = (2.0 ** torch.linspace(0.0, 8.0, steps=8)).double()
    xb = pt[..., None, :] * scales[:, None]
    h = torch.sin(torch.stack((xb, xb + 0.5 * 3.1415926), dim=-2)).reshape(n, -1)
    assert float((h - O.posenc_jax(pt, 0, 8, False)).abs().max()) < 1e-4       # = the oracle's encoder up to fp32 scales
    t = O.mlp(tp, "trunk", h, depth=6)
    w = O.mlp(tp, "w_net", t, depth=0)
    v = O.mlp(tp, "v_net", t, depth=0)
    np.testing.assert_allclose(outs[0][:, :3], w.detach().numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(outs[0][:, 3:], v.detach().numpy(), rtol=1e-9, atol=1e-12)
    g = H.uniform(10, "g", (n, 6), -1, 1).double()
    ((w * g[:, :3]).sum() + (v * g[:, 3:]).sum()).backward()
    bsrcs = srcs + [(g.numpy(), False)]
    dsrc = E.run_backward(prog, mode, tables, params, bsrcs, n, 1, stash)
    cx = {c: sl for (si, c), sl in prog.dsrc_map.items() if si == 0}
    got = np.stack([dsrc[:, cx[c]] for c in range(3)], axis=1)
    np.testing.assert_allclose(got, pt.grad.numpy(), rtol=1e-7, atol=1e-10)
    jobs = prog.wgrad_jobs(1 if bf16 else 0, n)
    _, gtot = prog.grad_offsets()
    flat = E.run_wgrad(prog, mode, jobs, stash, gtot)
    check_grads(prog, flat, tp, dict(f.named_parameters()))


@pytest.mark.parametrize("bf16", [True, False])
def test_legacy_nerf_aux_first_skip_emulated(bf16):
    """The nerf_pl NeRF (models/nerf.py:83-124) concatenates [input_xyz, h] BEFORE its skip layer: the generated
    features take the FIRST columns of that matrix and the running activation the ones after.  Small widths;
    embedded inputs as raw features; forward and every weight gradient."""
    from hypernerf_torch_amd.models import nerf as legacy_nerf
    torch.manual_seed(0)
    m = legacy_nerf.NeRF(D=4, W=64, in_channels_xyz=15, in_channels_dir=9, skips=[2])
    sd = load_hash(m, 17)
    n = 40
    x = H.uniform(11, "x", (n, 24), -1, 1).double()
    call = m._embedded_call(False)
    prog = call.program
    mode = E.Mode(bf16)
    tables = prog.host_tables(1 if bf16 else 0)
    params = np_params(prog)
    srcs = [(x.numpy(), False), None, None, None]
    outs, stash = E.run_forward(prog, mode, tables, params, srcs, n, 1, [4])
    tp = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y = O.legacy_nerf(tp, x, d=4, w=64, in_xyz=15, in_dir=9, skips=(2,))
    np.testing.assert_allclose(outs[0], y.detach().numpy(), rtol=1e-9, atol=1e-11)
    g = H.uniform(12, "g", (n, 4), -1, 1).double()
    (y * g).sum().backward()
    bsrcs = srcs + [(g.numpy(), False), (outs[0], False)]
    E.run_backward(prog, mode, tables, params, bsrcs, n, 1, stash)
    jobs = prog.wgrad_jobs(1 if bf16 else 0, n)
    _, gtot = prog.grad_offsets()
    flat = E.run_wgrad(prog, mode, jobs, stash, gtot)
    check_grads(prog, flat, tp, dict(m.named_parameters()))


def test_exact_argument_reduction_of_the_bf16_encoders():
    """The arithmetic behind hn_rev_split / hn_features4 (hn_mlp.hip), replayed in float32 numpy: x / 2pi staged as
    hi + lo (hi = fl(x c), lo = the FMA residual of that product + x (1/2pi - c)), feature argument =
    fract(f hi) + (f lo + phase).  For the power-of-two frequencies of posenc_orig the argument must agree with the
    exact f x / 2pi (mod 1) to ~1e-7 revolutions at EVERY octave, where the one-FMA form of rounds 1-2,
    fract(fl(f / 2pi) x + phase), is off by up to ~1e-5 at f = 512 — the error that carried 2/3 of the bf16 mode's
    gradient error."""
    rng = np.random.default_rng(5)
    x = rng.uniform(-1.0, 1.0, 200000).astype(np.float32)
    c_hi = np.float32(0.15915494309189535)
    c_lo = np.float32(0.15915494309189535 - float(c_hi))
    f32 = np.float32

    def fma(a, b, c):       # one rounding: the float32 product is exact in float64
        return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(np.float32)

    hi = (x * c_hi).astype(np.float32)
    lo = fma(x, c_lo, fma(x, c_hi, -hi))
    exact_rev = x.astype(np.float64) / (2.0 * np.pi)
    worst_new, worst_old = 0.0, 0.0
    for k in range(10):
        f = f32(2.0 ** k)
        big = (f * hi).astype(np.float32)                  # exact: a power of two times a float32
        assert np.array_equal(big.astype(np.float64), np.float64(f) * hi.astype(np.float64))
        arg_new = (big - np.floor(big)).astype(np.float32) + fma(lo, f, np.zeros_like(lo))
        want = np.float64(f) * exact_rev
        err_new = np.abs(((arg_new.astype(np.float64) - want + 0.5) % 1.0) - 0.5)
        scale = f32(float(f) * 0.15915494309189535)
        t_old = fma(x, scale, np.zeros_like(x))
        arg_old = (t_old - np.floor(t_old)).astype(np.float32)
        err_old = np.abs(((arg_old.astype(np.float64) - want + 0.5) % 1.0) - 0.5)
        worst_new, worst_old = max(worst_new, err_new.max()), max(worst_old, err_old.max())
        assert err_new.max() < 1.3e-7, (k, err_new.max())
    assert worst_old > 3e-6 and worst_old / worst_new > 25, (worst_old, worst_new)


def test_eight_bit_stash_layout_and_transposed_read():
    """HN_MODE_BF16_S8 (opt-in): a 32 x 32 tile is one 1-KiB unit; forward lane (point r, half h) stores its 16
    accumulator elements (features rho(i, h)) as 16 bytes at slot hn_stash8_slot(r, h).  The weight-gradient kernel's two
    ds_read_b64_tr_b8 per tile (offsets hn_dw8_offset, +512 for points 16..31) must hand lane (c, kg) — c = lane & 31,
    kg = lane >> 5 — the 8 consecutive points 16 mm + 8 kg .. + 7 of ONE feature, hn_dw8_feature(c), i.e. a valid MFMA
    operand up to a fixed feature permutation; that permutation is a bijection of the tile's 32 features (the epilogue
    undoes it in its addresses), and the reads are bank-conflict free (asserted inside the emulated instruction)."""
    assert sorted(E.stash8_slot(r, h) for r in range(32) for h in range(2)) == list(range(64))
    assert sorted(E.dw8_feature(c) for c in range(32)) == list(range(32))
    # tile[p][f] = a code of (point, feature) that fits 8 bits is not possible (1024 values): check in two passes
    for part in ("point", "feature"):
        img = np.zeros(1024)
        for r in range(32):
            for h in range(2):
                for i in range(16):
                    img[E.stash8_slot(r, h) * 16 + i] = r if part == "point" else E.rho(i, h)
        off = np.array([E.dw8_offset(l) for l in range(64)])
        for mm in range(2):
            got = E.ds_read_b64_tr_b8(img, off + 512 * mm)
            for lane in range(64):
                c, kg = lane & 31, lane >> 5
                if part == "point":
                    assert got[lane].tolist() == [16 * mm + 8 * kg + b for b in range(8)], (mm, lane, got[lane])
                else:
                    assert set(got[lane].tolist()) == {E.dw8_feature(c)}, (mm, lane, got[lane])


@pytest.mark.parametrize("launch_bytes", [None, 3.8e9, 8.5e10])
def test_wgrad_job_tables_cover_every_tile_product_once(launch_bytes):
    """Host logic of the batched weight-gradient launch (Program.wgrad_jobs, round 4: wave grids that use every wave a
    rectangle admits, job sizes that follow the launch's bytes, skip layers' two input segments as one rectangle with two
    X slots): whatever the sizing, every (layer part, dZ tile, input k-tile, point block) product is computed by exactly
    ONE job, every bias by exactly one, a job's k-tiles map onto the right stash slot, and every job satisfies the
    kernel's limits (<= 8 waves, <= 4 x 2 tiles per wave, a stage of `bps` blocks inside the build's ring stage (64 KiB; 8-bit stash 48 KiB), block ranges
    aligned to stages)."""
    m = models.NerfModel(EMB, n_samples_coarse=8, n_samples_fine=8, hyper_slice_method="bendy_sheet",
                         use_nerf_embed=True, use_alpha_cond=True)
    progs = [m._level_call("fine").program, warping.SE3Field(in_ch=3)._field_call(True).program]
    for prog in progs:
        for mode in (L.HN_MODE_BF16, L.HN_MODE_F32, L.HN_MODE_BF16_S8):
            for n_points in (196608, 1000):
                offs, _, _ = prog.layout(mode, n_points)
                nblk = (n_points + 31) // 32
                tile_kib = machine.mode_consts(mode)[1] // 1024
                stage_kib = 48 if mode == L.HN_MODE_BF16_S8 else L.WGRAD_MAX_STAGE_KB      # 2 x 64 KiB ring since round 5
                jobs = prog.wgrad_jobs(mode, n_points, job_bytes=machine.WGRAD_JOB_BYTES, launch_bytes=launch_bytes)
                assert len(jobs) > 0
                # what the jobs stream == what resolve_pending sizes a launch by (round 5: the launch's bytes are computed
                # from the programs of the pass, not carried over from the previous launch)
                streamed = int(((jobs["n_nt"] + jobs["n_kt"]).astype(np.int64) * (jobs["blk1"] - jobs["blk0"])).sum())
                assert streamed * tile_kib * 1024 == prog.wgrad_stream_bytes(mode, n_points)
                goffs0, _ = prog.grad_offsets()
                cut = sorted(goffs0)[len(goffs0) // 2]
                head = jobs[jobs["w_off"] >= cut]
                assert int(((head["n_nt"] + head["n_kt"]).astype(np.int64) * (head["blk1"] - head["blk0"])).sum()) \
                    * tile_kib * 1024 == prog.wgrad_stream_bytes(mode, n_points, goffs0, cut)
                slot_of = {int(o[0]): i for i, o in enumerate(offs)}
                seen, bias_seen = {}, {}
                for jb in jobs:
                    gn, gk, bps = int(jb["pad"]) & 255, (int(jb["pad"]) >> 8) & 255, (int(jb["pad"]) >> 16) & 255
                    n_nt, n_kt, n_kt1 = int(jb["n_nt"]), int(jb["n_kt"]), int(jb["n_kt1"])
                    assert 1 <= gn * gk <= 8 and gn <= n_nt and gk <= n_kt
                    assert -(-n_nt // gn) <= 4 and -(-n_kt // gk) <= 2
                    assert bps >= 1 and bps * (n_nt + n_kt) * tile_kib <= stage_kib
                    assert int(jb["blk0"]) % bps == 0 and 0 <= jb["blk0"] < jb["blk1"] <= nblk
                    assert 1 <= n_kt1 <= n_kt
                    assert int(jb["x_off"]) in slot_of and int(jb["z_off"]) in slot_of
                    if n_kt1 < n_kt:
                        assert int(jb["x2_off"]) in slot_of and jb["x2_t0"] + (n_kt - n_kt1) <= jb["x2_nt"]
                        assert jb["x_t0"] + n_kt1 <= jb["x_nt"]
                    else:
                        assert jb["x_t0"] + n_kt <= jb["x_nt"]
                    assert jb["z_t0"] + n_nt <= jb["z_nt"]
                    for i in range(n_nt):
                        row = int(jb["r0"]) + 32 * i
                        if row >= jb["r_end"]:
                            continue
                        for j in range(n_kt):
                            col = int(jb["c0"]) + 32 * j
                            if col >= jb["c_end"]:
                                continue
                            key = (int(jb["w_off"]), row, col)
                            iv = seen.setdefault(key, [])
                            iv.append((int(jb["blk0"]), int(jb["blk1"])))
                        if jb["b_off"] >= 0:
                            bias_seen.setdefault((int(jb["b_off"]), row), []).append((int(jb["blk0"]), int(jb["blk1"])))
                for table in (seen, bias_seen):
                    for key, iv in table.items():
                        iv.sort()
                        assert iv[0][0] == 0 and iv[-1][1] == nblk, (key, iv[:3])
                        assert all(a[1] == b[0] for a, b in zip(iv[:-1], iv[1:])), (key, iv[:4])
                # every weight element of every layer is covered
                goffs, _ = prog.grad_offsets()
                for ly in prog.layers:
                    for (w_id, b_id, row0, rows) in ly.parts:
                        for rt in range(0, rows, 32):
                            for ct in range(0, ly.in_features, 32):
                                assert (goffs[w_id], rt, ct) in seen, (ly.name, rt, ct)
                            if b_id >= 0:
                                assert (goffs[b_id], rt) in bias_seen, (ly.name, rt)
    # the wave grid itself: maximal active waves within the kernel's limits, for every rectangle shape
    for n_nt in range(1, 9):
        for n_kt in range(1, 9):
            gn, gk = machine.Program._wave_grid(n_nt, n_kt)
            assert gn * gk <= 8 and -(-n_nt // gn) <= 4 and -(-n_kt // gk) <= 2
            best = max(a * b for a in range(1, 9) for b in range(1, 8 // a + 1)
                       if a <= n_nt and b <= n_kt and -(-n_nt // a) <= 4 and -(-n_kt // b) <= 2)
            assert gn * gk == best, (n_nt, n_kt, gn, gk, best)
