"""CPU tests of the drop-in surface: the C-ABI library loads and exports every symbol include/hn_kernels.h
declares, struct mirrors have the C sizes, module constructors / state_dict keys / initialiser RNG order match
the reference's golden key lists, and ops refuse CPU tensors (no CPU fallback)."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest
import torch

import hypernerf_torch_amd as HN
from hypernerf_torch_amd import _lib as L
from hypernerf_torch_amd import machine
from hypernerf_torch_amd.hypernerf import model_utils, models, modules, warping
from hypernerf_torch_amd.models import nerf as legacy_nerf
from hypernerf_torch_amd.models import rendering as legacy_rendering

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMB = {"warp": list(range(100)), "camera": [0], "appearance": list(range(100)), "time": list(range(100))}


def test_library_exports_every_declared_symbol():
    HN.build()
    lib = L.load()
    header = open(os.path.join(ROOT, "include", "hn_kernels.h")).read()
    declared = set(re.findall(r"^int (hn_\w+)\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.hn_version() == 340


def test_abi_struct_sizes_match_c():
    lib = L.load()
    out = (ctypes.c_int32 * 8)()
    assert lib.hn_abi_sizes(out, 8) == 8
    c = list(out)
    assert c[0] == ctypes.sizeof(L.HnMlpArgs)
    assert c[1] == L.PACK_UNIT_DT.itemsize and c[2] == L.PACK_BIAS_DT.itemsize and c[3] == L.DWJOB_DT.itemsize
    assert c[4] == ctypes.sizeof(L.HnCompositeArgs)
    assert c[5] == L.FEAT_DT.itemsize and c[6] == ctypes.sizeof(L.HnSlot) and c[7] == ctypes.sizeof(L.HnSrc)


def test_argument_errors_are_reported_not_crashes():
    lib = L.load()
    a = L.HnMlpArgs()
    assert lib.hn_mlp_forward(ctypes.byref(a), None) < 0          # empty args: negative status, no launch
    assert lib.hn_mlp_backward(None, None) < 0
    c = L.HnCompositeArgs()
    assert lib.hn_composite_forward(ctypes.byref(c), None) < 0
    assert lib.hn_sample_pdf(None, 0, None, 0, None, 0, None, None, None, 0, 0, 0, None, None, None, None, None) < 0
    # the weight-gradient launches refuse job tables cut for a longer LDS stage than the library's ring holds (mode word
    # bits 8..15, KiB): -8 before anything else; the stage the host uses and "not stated" pass (no jobs: status 0)
    ok, none, bad = L.HN_MODE_BF16 | L.WGRAD_MAX_STAGE_KB << 8, L.HN_MODE_BF16, L.HN_MODE_BF16 | (L.WGRAD_MAX_STAGE_KB + 16) << 8
    assert lib.hn_mlp_wgrad_batched(bad, None, 0, None, None) == -8 and lib.hn_mlp_wgrad(bad, None, 0, None, None, None) == -8
    assert lib.hn_mlp_wgrad_batched_t(L.HN_MODE_F32 | 255 << 8, None, 0, None, None, None) == -8
    assert lib.hn_mlp_wgrad_batched(ok, None, 0, None, None) == 0 and lib.hn_mlp_wgrad_batched(none, None, 0, None, None) == 0
    assert lib.hn_mlp_wgrad_batched(L.HN_MODE_BF16_S8 | 16 << 8, None, 0, None, None) == 0      # bits 8.. = the dZ scale there


def test_workspace_query_matches_the_host_compiler():
    """hn_mlp_workspace_bytes (host arithmetic in the C library) against the layout the Python program compiler
    allocates: every slot is written by the forward or the backward program, so the larger of the two queries is
    exactly the allocation, in both modes and for ragged point counts."""
    lib = L.load()
    m = models.NerfModel(EMB, n_samples_coarse=8, n_samples_fine=8, hyper_slice_method="bendy_sheet",
                         use_nerf_embed=True, use_alpha_cond=True)
    progs = [m._level_call("fine").program, warping.SE3Field(in_ch=3)._field_call(True).program,
             legacy_nerf.NeRF().fused_call(legacy_nerf.Embedding(3, 10), legacy_nerf.Embedding(3, 4), False).program]
    for prog in progs:
        for mode in (L.HN_MODE_BF16, L.HN_MODE_F32, L.HN_MODE_BF16_S8):
            for n in (1, 32, 1000, 196608):
                fwd, bwd = prog.resolved_ops(mode, n)
                _, sb, mb = prog.layout(mode, n)
                got = []
                for back, ops in ((0, fwd), (1, bwd)):
                    ops = np.ascontiguousarray(ops, dtype=np.int32)
                    s_out, m_out = ctypes.c_int64(-1), ctypes.c_int64(-1)
                    rc = lib.hn_mlp_workspace_bytes(ops.ctypes.data_as(ctypes.c_void_p), len(ops), back, mode,
                                                    ctypes.c_int64(n), ctypes.byref(s_out), ctypes.byref(m_out))
                    assert rc == 0
                    assert 0 <= s_out.value <= sb and 0 <= m_out.value <= mb
                    got.append((s_out.value, m_out.value))
                assert max(g[0] for g in got) == sb and max(g[1] for g in got) == mb, (prog.name, mode, n, got, sb, mb)
        # the opt-in 8-bit stash: half the stash of the bf16 mode, the same masks, the same weight-gradient job grid
        # (same rectangles over the same point blocks; half the bytes per job)
        _, sb16, mb16 = prog.layout(L.HN_MODE_BF16, 196608)
        _, sb8, mb8 = prog.layout(L.HN_MODE_BF16_S8, 196608)
        assert sb8 * 2 == sb16 and mb8 == mb16
        j16 = prog.wgrad_jobs(L.HN_MODE_BF16, 196608, job_bytes=machine.WGRAD_JOB_BYTES)
        j8 = prog.wgrad_jobs(L.HN_MODE_BF16_S8, 196608, job_bytes=machine.WGRAD_JOB_BYTES // 2)
        cover = lambda j: sorted(zip(j["w_off"].tolist(), j["r0"].tolist(), j["c0"].tolist(), j["n_nt"].tolist(), j["n_kt"].tolist()))
        assert len(j8) > 0 and set(cover(j8)) == set(cover(j16))
        assert int(((j8["n_nt"] + j8["n_kt"]).astype(np.int64) * (j8["blk1"] - j8["blk0"])).sum()) == \
            int(((j16["n_nt"] + j16["n_kt"]).astype(np.int64) * (j16["blk1"] - j16["blk0"])).sum())
    assert machine.wgrad_mode_word(L.HN_MODE_BF16) == 1 | machine.WGRAD_STAGE_KB << 8
    assert machine.wgrad_mode_word(L.HN_MODE_F32) == 0 | machine.WGRAD_STAGE_KB << 8
    assert machine.wgrad_mode_word(L.HN_MODE_BF16_S8) == 2 | machine.DZ_SCALE_LOG2 << 8
    bad = np.zeros((1, 8), dtype=np.int32); bad[0, 0] = 99
    s_out, m_out = ctypes.c_int64(), ctypes.c_int64()
    assert lib.hn_mlp_workspace_bytes(bad.ctypes.data_as(ctypes.c_void_p), 1, 0, L.HN_MODE_BF16, ctypes.c_int64(32),
                                      ctypes.byref(s_out), ctypes.byref(m_out)) == -7
    assert lib.hn_mlp_workspace_bytes(None, 1, 0, L.HN_MODE_BF16, ctypes.c_int64(32), None, None) == -1


CASES = {
    "bendy": dict(hyper_slice_method="bendy_sheet", use_nerf_embed=False, use_alpha_cond=False),
    "bendy_cond": dict(hyper_slice_method="bendy_sheet", use_nerf_embed=True, use_alpha_cond=True),
    "bendy_rgbcond": dict(hyper_slice_method="bendy_sheet", use_nerf_embed=True, use_alpha_cond=True, use_rgb_cond=True),
    "nowarp": dict(use_warp=False, hyper_slice_method=None, use_nerf_embed=False, use_alpha_cond=False),
    "nowarp_cond": dict(use_warp=False, hyper_slice_method=None, use_nerf_embed=True, use_alpha_cond=True),
    "warp_noslice": dict(use_warp=True, hyper_slice_method=None, use_nerf_embed=False, use_alpha_cond=False,
                         hyper_slice_out_dim=0),
    "axis": dict(hyper_slice_method="axis_aligned_plane", hyper_slice_out_dim=8, use_nerf_embed=False,
                 use_alpha_cond=False),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_state_dict_keys_and_shapes_match_reference(golden_dir, case):
    g = np.load(os.path.join(golden_dir, f"g11_model_{case}_8_8.npz"))
    m = models.NerfModel(EMB, n_samples_coarse=8, n_samples_fine=8, view_fourier_dim=6, **CASES[case])
    sd = m.state_dict()
    assert sorted(sd.keys()) == g["keys"].tolist()
    assert [str(tuple(sd[k].shape)) for k in sorted(sd)] == g["shapes"].tolist()


def test_legacy_state_dict_keys(golden_dir):
    g = np.load(os.path.join(golden_dir, "g12_legacy_c_only.npz"))
    sd = legacy_nerf.NeRF().state_dict()
    assert sorted(sd.keys()) == g["keys"].tolist()
    assert [str(tuple(sd[k].shape)) for k in sorted(sd)] == g["shapes"].tolist()
    assert legacy_nerf.Embedding(3, 10).out_channels == 63 and legacy_nerf.Embedding(3, 4).out_channels == 27


def test_reference_constructor_errors():
    with pytest.raises(ValueError):
        models.NerfModel(EMB, use_nerf_embed=True, use_alpha_cond=False, use_rgb_cond=False)
    with pytest.raises(UnboundLocalError):
        models.NerfModel(EMB, n_samples_fine=0)                 # hypernerf/models.py:292-309
    with pytest.raises(UnboundLocalError):
        models.NerfModel(EMB, share_GLO=False)                  # hypernerf/models.py:167-186
    tf = warping.TranslationField(in_ch=3)
    with pytest.raises(Exception):
        tf(torch.zeros(2, 3), torch.zeros(2, 8), None, return_jacobian=True)


def test_unsupported_mlp_arguments_are_refused_at_construction():
    """Round 6 (verdict item 7): what the HIP MLP machine does not run — hidden activations other than ReLU, width > 256,
    Sigmoid on a wide output, unknown activations, a ReLU rgb head of NerfMLP — raises NotImplementedError in the
    CONSTRUCTOR (no GPU needed), not at the first forward; everything the reference itself builds constructs
    (hypernerf/modules.py:62-114, models.py:139-166)."""
    from hypernerf_torch_amd.hypernerf import modules
    for bad in (dict(hidden_activation=torch.nn.Tanh()), dict(hidden_activation=torch.nn.Sigmoid()), dict(width=512),
                dict(width=257), dict(out_ch=8, output_activation=torch.nn.Sigmoid()), dict(output_activation=torch.nn.GELU()),
                dict(in_ch=193), dict(out_ch=300)):
        kw = dict(in_ch=16, out_ch=3)
        kw.update(bad)
        with pytest.raises(NotImplementedError):
            modules.MLP(**kw)
    with pytest.raises(NotImplementedError):
        modules.NerfMLP(in_ch=63, rgb_activation=torch.nn.ReLU())
    # what the reference constructs, and the generalities the machine does run
    modules.MLP(in_ch=71, out_ch=128, depth=6, width=128)
    modules.MLP(in_ch=115, out_ch=256, depth=8, width=256, output_activation=torch.nn.ReLU())
    modules.MLP(in_ch=167, out_ch=3, depth=4, width=128, output_activation=torch.nn.Sigmoid())
    modules.MLP(in_ch=5, out_ch=40, depth=0, width=53, skips=[])
    modules.MLP(in_ch=5, out_ch=2, depth=3, width=24, output_activation=torch.nn.ReLU(), hidden_norm="anything")   # inert upstream too
    modules.NerfMLP(in_ch=63)
    modules.NerfMLP(in_ch=63, rgb_activation=torch.nn.Sigmoid())
    modules.HyperSheetMLP(in_ch=3, in_ch_embed=8)


def test_no_cpu_fallback():
    m = models.NerfModel(EMB, n_samples_coarse=8, n_samples_fine=8, hyper_slice_method="bendy_sheet")
    rays = {"origins": torch.zeros(4, 3), "directions": torch.ones(4, 3), "viewdirs": None,
            "metadata": {k: torch.zeros(4, dtype=torch.long) for k in ("warp", "camera", "appearance", "time")}}
    with pytest.raises(L.HnError):
        m(rays, {})
    with pytest.raises(L.HnError):
        legacy_rendering.render_rays([legacy_nerf.NeRF()], [legacy_nerf.Embedding(3, 10), legacy_nerf.Embedding(3, 4)],
                                     torch.zeros(4, 8))
    with pytest.raises(L.HnError):
        modules.MLP(8, 3, depth=2, width=32)(torch.zeros(5, 8))


def test_ray_dict_plumbing():
    rays = torch.arange(5 * 9, dtype=torch.float32).view(5, 9)
    rd = model_utils.prepare_ray_dict(rays)
    assert rd["viewdirs"] is None and torch.equal(rd["origins"], rays[:, :3]) and torch.equal(rd["directions"], rays[:, 3:6])
    assert rd["metadata"]["time"].dtype == torch.long and torch.equal(rd["metadata"]["warp"], rays[:, 8].long())
    part = model_utils.extract_rays_batch(rd, 1, 3)
    assert part["origins"].shape == (2, 3) and part["metadata"]["time"].shape == (2,)
    assert model_utils.get_posenc_ch_orig(3, 10) == 63 and model_utils.get_posenc_ch(3, 0, 8, False) == 48
    both = model_utils.concat_ray_batch([{"a": torch.zeros(2, 3)}, {"a": torch.ones(1, 3)}])
    assert both["a"].shape == (3, 3)


def test_init_matches_reference_rng_order():
    """Same torch seed -> same parameter values as a module built the reference's way (nn.Linear creation order,
    then xavier on hidden layers, then the output initialiser): hypernerf/modules.py:99-109."""
    torch.manual_seed(123)
    m = modules.MLP(in_ch=7, out_ch=3, depth=3, width=16, skips=[1])
    torch.manual_seed(123)
    lins = [torch.nn.Linear(7, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16 + 7, 16)]
    logit = torch.nn.Linear(16, 3)
    for l in lins:
        torch.nn.init.xavier_uniform_(l.weight)
    torch.nn.init.xavier_uniform_(logit.weight)
    for i, l in enumerate(lins):
        assert torch.equal(m.linears[i].weight, l.weight) and torch.equal(m.linears[i].bias, l.bias)
    assert torch.equal(m.logit_layer.weight, logit.weight)


def test_param_arena_views_and_detach():
    """ParamArena: p.data / p.grad become views of two flat buffers; values survive; one tensor to optimise;
    zero_grad(set_to_none=True) on a parameter detaches it (kernels then fall back to autograd gradients)."""
    lin = torch.nn.Linear(5, 3)
    emb = torch.nn.Embedding(7, 2)
    w0, b0, e0 = lin.weight.detach().clone(), lin.bias.detach().clone(), emb.weight.detach().clone()
    params = list(lin.parameters()) + list(emb.parameters())
    arena = HN.ParamArena(params + [lin.weight])        # duplicates are ignored
    assert arena.numel == 16 + 4 + 16 and arena.offsets == [0, 16, 20]
    assert torch.equal(lin.weight, w0) and torch.equal(lin.bias, b0) and torch.equal(emb.weight, e0)
    assert lin.weight.data_ptr() == arena.data.data_ptr() and lin.bias.data_ptr() == arena.data.data_ptr() + 64
    assert HN.ParamArena.lookup(params)[1] == [0, 16, 20]
    # ordinary autograd still accumulates into the views
    y = lin(torch.ones(2, 5)).sum() + emb(torch.tensor([1, 1, 3])).sum()
    y.backward()
    assert torch.allclose(arena.grad[:15], torch.full((15,), 2.0)) and float(arena.grad[15]) == 0.0   # padding
    assert torch.allclose(emb.weight.grad[1], torch.full((2,), 2.0))
    # one optimiser tensor; stepping it changes the module's weights and the arena version
    v = arena.version()
    torch.optim.SGD([arena.flat_param], lr=0.5).step()
    assert arena.version() != v
    assert torch.allclose(lin.weight, w0 - 1.0) and torch.allclose(emb.weight[3], e0[3] - 0.5)
    arena.zero_grad()
    assert float(lin.weight.grad.abs().sum()) == 0.0 and arena.attached(lin.weight) == 0
    lin.weight.grad = None
    assert arena.attached(lin.weight) is None and HN.ParamArena.lookup(params) is None


def test_load_ckpt_lightning_and_bare(tmp_path):
    """Reference checkpoints load unchanged: Lightning layout ({'state_dict': {'nerf.<name>': ...}}, train.py:48) and a
    bare prefixed state dict, `prefixes_to_ignore`, empty path = no-op (reference: utils/__init__.py:66-89)."""
    from hypernerf_torch_amd.utils import extract_model_state_dict, load_ckpt
    emb = {"warp": list(range(10)), "camera": [0], "appearance": list(range(10)), "time": list(range(10))}
    kw = dict(hyper_slice_method="bendy_sheet", use_nerf_embed=True, use_alpha_cond=True, view_fourier_dim=6)
    src = models.NerfModel(emb, **kw)
    dst = models.NerfModel(emb, **kw)
    blob = {"epoch": 3, "state_dict": {"nerf." + k: v.clone() for k, v in src.state_dict().items()}}
    blob["state_dict"]["loss.weight"] = torch.zeros(1)            # other modules of the Lightning system
    path = os.path.join(tmp_path, "epoch=3.ckpt")
    torch.save(blob, path)
    got = extract_model_state_dict(path, "nerf")
    assert set(got) == set(src.state_dict())
    load_ckpt(dst, "", "nerf")                                     # no-op
    assert not torch.equal(dst.warp_field.mlp.linears[0].weight, src.warp_field.mlp.linears[0].weight)
    arena = HN.ParamArena(dst.parameters())                        # loading goes through the arena views
    load_ckpt(dst, path, "nerf")
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v), k
    assert arena.attached(dst.warp_field.mlp.linears[0].weight) is not None
    # bare dict + ignored prefix
    bare = os.path.join(tmp_path, "bare.pt")
    torch.save({"nerf." + k: v + 1.0 for k, v in src.state_dict().items()}, bare)
    load_ckpt(dst, bare, "nerf", prefixes_to_ignore=["warp_field"])
    assert torch.equal(dst.warp_field.mlp.linears[0].weight, src.warp_field.mlp.linears[0].weight)
    assert torch.equal(dst.hyper_sheet_mlp.mlp.linears[0].weight, src.hyper_sheet_mlp.mlp.linears[0].weight + 1.0)


def test_save_ckpt_writes_the_reference_layout(tmp_path, golden_dir):
    """utils.save_ckpt (SURVEY.md §8 f3 'export back'): {'state_dict': {'nerf.<name>': fp32 CPU tensor}} with exactly
    the parameter names the reference's own model has (key list recorded from the reference in the g11 fixtures), so
    that the reference's `load_ckpt(nerf, path, model_name='nerf')` (utils/__init__.py:83-88: strip 'nerf.', update,
    load_state_dict) restores it; round trip through load_ckpt here, also from a model living in a ParamArena."""
    from hypernerf_torch_amd.utils import load_ckpt, save_ckpt
    g = np.load(os.path.join(golden_dir, "g11_model_bendy_cond_8_8.npz"))
    kw = dict(hyper_slice_method="bendy_sheet", use_nerf_embed=True, use_alpha_cond=True, view_fourier_dim=6)
    src = models.NerfModel(EMB, **kw)
    arena = HN.ParamArena(src.parameters())
    path = save_ckpt(src, os.path.join(tmp_path, "epoch=7.ckpt"), model_name="nerf", epoch=7, global_step=1234)
    blob = torch.load(path, map_location="cpu")
    assert blob["epoch"] == 7 and blob["global_step"] == 1234
    assert sorted(blob["state_dict"]) == ["nerf." + k for k in g["keys"].tolist()]
    for k, v in blob["state_dict"].items():
        assert v.dtype == torch.float32 and v.device.type == "cpu"
        assert v.untyped_storage().nbytes() == v.numel() * 4, "arena views must be exported as compact copies"
    # what the reference's loader does with it (utils/__init__.py:66-88), restated inline
    ref_side = {k[len("nerf") + 1:]: v for k, v in blob["state_dict"].items() if k.startswith("nerf")}
    dst = models.NerfModel(EMB, **kw)
    sd = dst.state_dict()
    sd.update(ref_side)
    dst.load_state_dict(sd)
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v), k
    dst2 = models.NerfModel(EMB, **kw)
    load_ckpt(dst2, path, "nerf")
    assert all(torch.equal(dst2.state_dict()[k], v) for k, v in src.state_dict().items())
    assert arena.attached(src.warp_field.mlp.linears[0].weight) is not None


def test_multistep_lr_matches_torch():
    """optim.MultiStepLR (the reference's 'steplr', utils/__init__.py:43-46) against torch's scheduler, epoch by
    epoch, including a resume through state_dict."""
    from hypernerf_torch_amd.optim import MultiStepLR

    class Dummy:                       # the only thing the scheduler touches
        def __init__(self, lr):
            self.param_groups = [{"lr": lr}]

    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.Adam([p], lr=5e-4)
    tsch = torch.optim.lr_scheduler.MultiStepLR(topt, milestones=[2, 5, 6], gamma=0.5)
    mine = Dummy(5e-4)
    sch = MultiStepLR(mine, [2, 5, 6], 0.5)
    for epoch in range(9):
        assert abs(mine.param_groups[0]["lr"] - topt.param_groups[0]["lr"]) < 1e-12, epoch
        topt.step(); tsch.step(); sch.step()
        if epoch == 3:
            state = sch.state_dict()
            mine = Dummy(123.0)
            sch = MultiStepLR(mine, [1], 0.1)
            sch.load_state_dict(state)
    assert sch.get_last_lr() == [mine.param_groups[0]["lr"]]


def test_bench_launch_plan():
    """`bench.py --gpus N` started plainly is the launcher (reference: Lightning spawns `devices=num_gpus` ranks
    itself, train.py:224-229): N child commands with the torch.distributed environment per rank, rendezvous on
    127.0.0.1, dmabuf IPC kept — and no GPU call in the parent (the plan is produced on this GPU-less machine)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1",
                          "--launch-plan"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    plan = json.loads(out.stdout.strip().splitlines()[-1])
    assert plan["n_ranks"] == 4 and len(plan["ranks"]) == 4
    ports = set()
    for r, rk in enumerate(plan["ranks"]):
        e = rk["env"]
        assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"]) == (str(r), str(r), "4")
        # the IPC variable is INHERITED by the ranks, not written by the launcher (round 5: opt-in)
        assert e["MASTER_ADDR"] == "127.0.0.1" and "HSA_ENABLE_IPC_MODE_LEGACY" not in e
        ports.add(e["MASTER_PORT"])
        assert rk["cmd"][1].endswith("bench.py") and "--launch-plan" not in rk["cmd"]
        assert rk["cmd"][2:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert len(ports) == 1
    # the 8-GPU node of BASELINE config 4: 8 commands, RANK == LOCAL_RANK == 0..7 (one rank per GPU, never two on one),
    # one rendezvous port, the caller's MASTER_PORT honoured; the IPC variable is written only on request
    # (HN_SET_IPC_ENV: "1" = the pool's known-good 0, any other value verbatim) — the ranks inherit the caller's otherwise
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod_plan", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    plan8 = bench.launch_plan(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], env={"MASTER_PORT": "29611"})
    assert len(plan8) == 8
    assert [e["LOCAL_RANK"] for _, e in plan8] == [str(i) for i in range(8)] == [e["RANK"] for _, e in plan8]
    assert {e["MASTER_PORT"] for _, e in plan8} == {"29611"} and {e["WORLD_SIZE"] for _, e in plan8} == {"8"}
    assert all(c[2:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] for c, _ in plan8)
    assert all("HSA_ENABLE_IPC_MODE_LEGACY" not in e for _, e in plan8)
    assert all("HSA_ENABLE_IPC_MODE_LEGACY" not in e
               for _, e in bench.launch_plan(2, [], env={"HSA_ENABLE_IPC_MODE_LEGACY": "1"}))      # inherited as it is
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for _, e in bench.launch_plan(2, [], env={"HN_SET_IPC_ENV": "1"}))
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "2" for _, e in bench.launch_plan(2, [], env={"HN_SET_IPC_ENV": "2"}))
    # a rank whose WORLD_SIZE disagrees with --gpus refuses to run (it would report a point of the wrong curve)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert bad.returncode == 4 and "WORLD_SIZE=1" in bad.stderr
    # more RCCL ranks than GPUs is refused (two ranks on one device hang RCCL); only the gloo debugging mode may fold
    fold = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                          env=dict({k: v for k, v in env.items() if k != "HN_DIST_BACKEND"}, WORLD_SIZE="2", RANK="1",
                                   LOCAL_RANK="1"), timeout=300)
    assert fold.returncode == 5 and "HN_DIST_BACKEND=gloo" in fold.stderr


def test_bench_also_block_host_logic(monkeypatch):
    """The `also` block the default `python bench.py` appends to its line (configs 3, 5, 1 + the eval image loop as child
    processes): entries are built from the children's own JSON lines, a child that fails or times out becomes an
    `error` / `skipped` entry instead of costing the headline, the flags every child gets keep it from recursing or
    timing the CPU oracle, and the block stops starting children when its time budget is spent."""
    import importlib.util
    import json
    import subprocess
    spec = importlib.util.spec_from_file_location("bench_mod_also", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = []

    def fake_run(cmd, capture_output, text, timeout):
        calls.append((cmd, timeout))
        script = os.path.basename(cmd[1])
        if script == "eval_bench.py":
            line = {"images_per_s": 13.5, "s_per_image": 0.074, "ray_samples_per_s": 3.3e8, "mfma_frac_of_2.5PF": 0.32,
                    "build": {"kernel_src_sha256": "abc"}}
            return subprocess.CompletedProcess(cmd, 0, "banner\n" + json.dumps(line) + "\n", "")
        cfg = cmd[cmd.index("--config") + 1]
        if cfg == "5":
            return subprocess.CompletedProcess(cmd, 3, "", "boom")
        if cfg == "1":
            raise subprocess.TimeoutExpired(cmd, timeout)
        line = {"value": 8.2e7, "unit": "ray-samples/s", "ms_per_step": 38.0, "dtype": "bf16", "steps": 5, "repeats": 3,
                "ms_per_step_spread": [37.9, 38.1], "config": {"workload": "w"}, "step_mfma_frac": 0.21,
                "roofline": {"kernel": "hn_wgrad_kernel<true>", "frac": 0.18}, "build": {"kernel_src_sha256": "abc"}}
        return subprocess.CompletedProcess(cmd, 0, json.dumps(line) + "\n", "")
    monkeypatch.setattr(subprocess, "run", fake_run)
    out = bench.also_block()
    assert set(out) >= {"config3", "config5", "config1", "render_image", "wall_s", "note"}
    assert out["config3"]["value"] == 8.2e7 and out["config3"]["frac"] == 0.18 and out["config3"]["kernel"].startswith("hn_wgrad")
    assert out["config3"]["build"]["kernel_src_sha256"] == "abc"
    assert "error" in out["config5"] and "boom" in out["config5"]["stderr_tail"]
    assert "skipped" in out["config1"]
    assert out["render_image"]["value"] == 13.5 and out["render_image"]["frac"] == 0.32
    for cmd, timeout in calls:
        assert timeout <= 40.0
        if os.path.basename(cmd[1]) == "bench.py":
            assert "--no-also" in cmd and "--no-cpu-baseline" in cmd
    # budget spent: no further child is started
    calls.clear()
    monkeypatch.setattr(bench, "ALSO_BUDGET_S", 0.0)
    out = bench.also_block()
    assert not calls and all("skipped" in out[k] for k in ("config3", "config5", "config1", "render_image"))


def test_pmc_traffic_is_tied_to_the_build(tmp_path, monkeypatch):
    """bench.py quotes a profiles/rNN_traffic_configC.json only when it was collected on the running kernels."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))

    class A:
        config, rays, nc, nf, precision = 2, 1024, 64, 64, "bf16"
    mine = L.build_id()["kernel_src_sha256"]
    base = {"config": 2, "rays": 1024, "nc": 64, "nf": 64, "bytes_per_step": 1.0, "per_kernel_launch": {}}
    (prof / "r02_traffic_config2.json").write_text(json.dumps(base))                       # unstamped: never quoted
    t, note = bench._pmc_traffic(A)
    assert t is None and "not quoted" in note
    (prof / "r03_traffic_config2.json").write_text(json.dumps(dict(base, build={"kernel_src_sha256": "0" * 16})))
    t, note = bench._pmc_traffic(A)
    assert t is None and "0000" in note
    (prof / "r04_traffic_config2.json").write_text(json.dumps(dict(base, build={"kernel_src_sha256": mine})))
    t, note = bench._pmc_traffic(A)
    assert t is not None and note is None and t["source"].endswith("r04_traffic_config2.json")
    lib = L.build_id().get("lib_sha256")
    if lib is not None:           # a comment-only edit of the sources: other source hash, the SAME built library -> quoted
        (prof / "r04_traffic_config2.json").unlink()
        (prof / "r05_traffic_config2.json").write_text(json.dumps(dict(base, build={"kernel_src_sha256": "1" * 16, "lib_sha256": lib})))
        t, note = bench._pmc_traffic(A)
        assert t is not None and note is None and t["source"].endswith("r05_traffic_config2.json")
    A.precision = "fp32"          # a bf16 collection says nothing about the fp32 mode's bytes
    assert bench._pmc_traffic(A)[0] is None


@pytest.mark.skipif(not os.path.isdir("/root/reference/utils"), reason="build container only: needs /root/reference")
def test_ckpt_round_trip_through_the_reference_loader():
    """SURVEY.md §8 f3, executed rather than argued: tests/golden/check_ckpt_with_reference.py imports the reference's
    own `load_ckpt` / `extract_model_state_dict` (utils/__init__.py:66-88) and round-trips checkpoints between the
    reference's modules and this package's (arena-backed) ones, both directions, NerfModel and legacy NeRF."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "check_ckpt_with_reference.py")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["ours_to_reference_tensors"] == rep["reference_to_ours_tensors"] == 94
    assert rep["legacy_ours_to_reference_tensors"] == rep["legacy_reference_to_ours_tensors"] == 24


def test_lr_schedules_match_the_reference(golden_dir):
    """optim.get_scheduler against learning-rate sequences recorded from the REFERENCE's own get_scheduler
    (utils/__init__.py:43-59 + utils/warmup_scheduler.py; tests/golden/make_lr_golden.py): steplr, cosine and the
    GradualWarmup wrapper around both — the hand-over quirks included.  'poly' raises NameError upstream (unpinned):
    checked against the formula the reference states."""
    import json
    import types
    from hypernerf_torch_amd import optim
    g = np.load(os.path.join(golden_dir, "g17_lr_schedules.npz"))
    cases = json.loads(str(g["cases"]))
    assert len(cases) >= 6
    for name, hp in cases.items():
        opt = types.SimpleNamespace(param_groups=[{"lr": 5e-4}])
        sch = optim.get_scheduler(types.SimpleNamespace(optimizer="adam", **hp), opt)
        lrs = [opt.param_groups[0]["lr"]]
        for _ in range(len(g[name]) - 1):
            sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        np.testing.assert_allclose(np.array(lrs), g[name], rtol=1e-9, atol=1e-15, err_msg=name)
        # resume mid-schedule (checkpoint at every epoch, warm-up hand-over included): a FRESH optimizer + scheduler
        # loaded with the saved state continues on the recorded sequence — the wrapped scheduler's state and the
        # learning rate the optimizer held travel with it
        for cut in range(1, len(g[name]) - 1):
            opt = types.SimpleNamespace(param_groups=[{"lr": 5e-4}])
            sch = optim.get_scheduler(types.SimpleNamespace(optimizer="adam", **hp), opt)
            for _ in range(cut):
                sch.step()
            saved = json.loads(json.dumps(sch.state_dict()))          # plain data: survives a checkpoint file
            opt2 = types.SimpleNamespace(param_groups=[{"lr": 5e-4}])
            sch2 = optim.get_scheduler(types.SimpleNamespace(optimizer="adam", **hp), opt2)
            sch2.load_state_dict(saved)
            assert opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"], (name, cut)
            rest = []
            for _ in range(len(g[name]) - 1 - cut):
                sch2.step()
                rest.append(opt2.param_groups[0]["lr"])
            np.testing.assert_allclose(np.array(rest), g[name][cut + 1:], rtol=1e-9, atol=1e-15, err_msg=f"{name} resumed at {cut}")
    opt = types.SimpleNamespace(param_groups=[{"lr": 5e-4}])
    sch = optim.get_scheduler(types.SimpleNamespace(lr_scheduler="poly", num_epochs=10, poly_exp=0.9, warmup_epochs=0), opt)
    for e in range(1, 11):
        sch.step()
        assert abs(opt.param_groups[0]["lr"] - 5e-4 * (1 - e / 10) ** 0.9) < 1e-15
    with pytest.raises(ValueError):
        optim.get_scheduler(types.SimpleNamespace(lr_scheduler="nope", warmup_epochs=0), opt)


def test_hot_kernels_stay_within_their_register_budget():
    """hipcc's own resource report for the machine kernels (gfx950 cross-compile, no GPU needed): 256 VGPRs at two
    waves per SIMD, and scratch within what the allocator has been seen to spill outside the tile loops.  A stray
    run-time index into the by-value kernel arguments once put all 2.5 KB of them on the stack of every forward kernel
    (2568 B/lane, -20 % throughput) without failing a single parity test: this is the test that fails then."""
    import subprocess
    cmd = [L.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-S",
           "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", os.devnull,
           os.path.join(L.CSRC, "hn_mlp.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    info, cur = {}, None
    for line in res.stderr.splitlines():
        m_ = re.search(r"Function Name: (\S+)", line)
        if m_:
            cur = info.setdefault(m_.group(1), {})
        for key in ("VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]"):
            m2 = re.search(re.escape(key) + r": (\d+)", line)
            if m2 and cur is not None and "AGPR" not in line.split(key)[0][-3:]:
                cur[key] = int(m2.group(1))
    bounds = {"_Z17hn_mlp_fwd_kernelILb1ELi2ELb0ELb1ELb0EEv9HnMlpArgs": 0, "_Z17hn_mlp_fwd_kernelILb1ELi3ELb0ELb1ELb0EEv9HnMlpArgs": 0,
              "_Z17hn_mlp_fwd_kernelILb1ELi2ELb0ELb0ELb0EEv9HnMlpArgs": 32, "_Z17hn_mlp_fwd_kernelILb1ELi3ELb0ELb0ELb0EEv9HnMlpArgs": 96,
              "_Z17hn_mlp_fwd_kernelILb1ELi2ELb1ELb1ELb0EEv9HnMlpArgs": 64, "_Z17hn_mlp_fwd_kernelILb1ELi3ELb1ELb1ELb0EEv9HnMlpArgs": 160,
              "_Z17hn_mlp_bwd_kernelILb1ELb0ELb0EEv9HnMlpArgs": 0, "_Z17hn_mlp_bwd_kernelILb1ELb1ELb0EEv9HnMlpArgs": 256, "_Z15hn_wgrad_kernelILb1ELb0EEv14HnDwBatchTable": 0,
              # the opt-in 8-bit-stash builds (HN_MODE_BF16_S8)
              "_Z17hn_mlp_fwd_kernelILb1ELi2ELb0ELb1ELb1EEv9HnMlpArgs": 0, "_Z17hn_mlp_fwd_kernelILb1ELi3ELb0ELb1ELb1EEv9HnMlpArgs": 0,
              "_Z17hn_mlp_bwd_kernelILb1ELb0ELb1EEv9HnMlpArgs": 0, "_Z15hn_wgrad_kernelILb1ELb1EEv14HnDwBatchTable": 0,
              # round 4: the fp32 weight-gradient build reloaded a spilled DMA address inside its stage loop (24-28 B of
              # scratch since round 2; a scratch reload waits for every LDS-DMA in front of it): offsets in KiB, no scratch
              "_Z15hn_wgrad_kernelILb0ELb0EEv14HnDwBatchTable": 0}
    for name, max_scratch in bounds.items():
        assert name in info, sorted(info)
        assert info[name]["VGPRs"] <= 256 and info[name]["Occupancy [waves/SIMD]"] == 2, (name, info[name])
        assert info[name]["ScratchSize [bytes/lane]"] <= max_scratch, (name, info[name])


def test_wgrad_ring_constants_mirror_the_kernel_defaults():
    """The host cuts hn_wgrad_kernel's jobs into LDS stages (machine.Program.wgrad_jobs); ring depth and pieces per wave
    are compile-time constants of the kernel.  _lib mirrors their defaults: a stage the host makes must fit the ring the
    library was built with (a stage longer than 8 x MAXSLOT KiB is silently never loaded — NaN gradients)."""
    src = open(os.path.join(L.CSRC, "hn_mlp.hip")).read()
    stages = int(re.search(r"#define HN_WGRAD_STAGES (\d+)", src).group(1))
    maxslot = int(re.search(r"#define HN_WGRAD_MAXSLOT (\d+)", src).group(1))
    if not os.environ.get("HN_WGRAD_STAGES") and not os.environ.get("HN_WGRAD_MAXSLOT"):
        assert (L.WGRAD_STAGES, L.WGRAD_MAXSLOT) == (stages, maxslot)
    assert L.WGRAD_STAGES * L.WGRAD_MAX_STAGE_KB <= 160 and L.WGRAD_MAX_STAGE_KB <= 8 * L.WGRAD_MAXSLOT
    from hypernerf_torch_amd import machine as M
    assert M.WGRAD_STAGE_KB <= L.WGRAD_MAX_STAGE_KB
    # every job of a config-2-sized launch: blocks per stage x units per block within the per-wave slot table and the ring
    m = models.NerfModel(EMB, n_samples_coarse=8, n_samples_fine=8, hyper_slice_method="bendy_sheet",
                         use_nerf_embed=True, use_alpha_cond=True)
    progs = [m._level_call("fine").program, warping.SE3Field(in_ch=3)._field_call(True).program,
             legacy_nerf.NeRF().fused_call(legacy_nerf.Embedding(3, 10), legacy_nerf.Embedding(3, 4), False).program]
    for prog in progs:
        for mode in (L.HN_MODE_BF16, L.HN_MODE_F32):
            tile_units = M.mode_consts(mode)[1] // 1024
            for jobs in (prog.wgrad_jobs(mode, 1000), prog.wgrad_jobs(mode, 196608, job_bytes=M.WGRAD_JOB_BYTES)):
                bps = (jobs["pad"] >> 16) & 255
                units = bps * (jobs["n_nt"] + jobs["n_kt"]) * tile_units
                assert (bps >= 1).all() and int(units.max()) <= 8 * L.WGRAD_MAXSLOT
                assert int(units.max()) * L.WGRAD_STAGES <= 160


def test_adam_rest_ranges_partition_the_arena():
    """hn_mlp_wgrad_reduce_adam steps every element of the arena exactly once only if the destinations of the reduce
    launch (32 x 32 tiles, bias records, the gathered table's rows) plus the `rest` ranges PARTITION it:
    machine.adam_rest_ranges proves that on the host.  Synthetic tables first (edge tiles, a masked table, an overlap that
    must be refused), then the real job tables of the config-2 programs laid out in a ParamArena."""
    from hypernerf_torch_amd import machine as M
    from hypernerf_torch_amd.arena import ParamArena
    # a 70 x 50 matrix at offset 8 (3 x 2 tiles, ragged edges), its 70 biases at 3600 (3 dZ tiles), a 10 x 8 table at 3700
    tiles = np.zeros(7, dtype=L.DWREDUCE_DT)
    k = 0
    for i in range(3):
        for j in range(2):
            tiles[k] = (0, 8, 50, 32 * i, 32 * j, 70, 50, 0, 1)
            k += 1
    tiles[6] = (0, 3600, 0, 0, 3, 70, 0, 0, 1)
    ranges, cov = M.adam_rest_ranges(tiles, 4000, (3700, 10, 8, 0b10111111))
    assert ranges is not None and int(cov.max()) == 1
    full = cov.astype(np.int64).copy()
    for a, n in ranges:
        assert 0 < n <= M.ADAM_REST_CHUNK
        full[a:a + n] += 1
    assert (full == 1).all()                                             # every element exactly once
    assert cov[8:8 + 3500].all() and cov[3600:3670].all() and not cov[:8].any() and not cov[3670:3700].any()
    assert not cov[3700 + 6:3700 + 80:8].any() and cov[3700:3700 + 6].all()      # column 6 of the table is masked out -> rest
    big = np.zeros(1, dtype=L.DWREDUCE_DT)
    big[0] = (0, 0, 0, 0, 200, 6400, 0, 0, 1)                           # a 6400-row bias: the rest behind it is chunked
    ranges, _ = M.adam_rest_ranges(big, 6400 + 5000)
    assert [int(n) for _, n in ranges] == [2048, 2048, 904] and int(ranges[0][0]) == 6400
    dup = np.concatenate([tiles, tiles[:1]])
    assert M.adam_rest_ranges(dup, 4000)[0] is None                     # an element that is a destination twice: no fusion
    # the real thing: the three programs of a config-2 step on one arena
    m = models.NerfModel(EMB, n_samples_coarse=64, n_samples_fine=64, hyper_slice_method="bendy_sheet",
                         use_nerf_embed=True, use_alpha_cond=True)
    arena = ParamArena(m.parameters())
    calls = [m._level_call("coarse"), m._level_call("fine"), m._template_reuse_call("fine", m.hyper_sheet_out_dim, False, True)]
    grp = []
    total = sum(c.program.wgrad_stream_bytes(L.HN_MODE_BF16, 65536) for c in calls)
    for c in calls:
        goffs = tuple(ParamArena.lookup(c.program.params)[1])
        # (MlpRunner.wgrad_tables' host half: it asks the runtime whether a stream capture is open, which needs a GPU)
        jh = np.ascontiguousarray(c.program.wgrad_jobs(L.HN_MODE_BF16, 65536, grad_offsets=goffs,
                                                       job_bytes=M.WGRAD_JOB_BYTES, launch_bytes=total))
        tl = M.slab_tiles(jh)
        jh["p_tile"] = np.cumsum(tl) - tl
        grp.append(M.ResolvedWgrad(L.HN_MODE_BF16, torch.zeros(1), len(jh), arena.grad, arena.grad, None, jobs_host=jh,
                                   owner=c.runner))
    red = M._reduce_tables(grp, "cpu")
    assert red[3]["one_buffer"]
    table = m.warp_embed.embed.weight
    emb = (ParamArena.lookup([table])[1][0], table.shape[0], table.shape[1], (1 << table.shape[1]) - 1)
    ranges, cov = M.adam_rest_ranges(red[3]["tiles_host"], arena.numel, emb)
    assert ranges is not None, "two destination records of one launch overlap"
    full = cov.astype(np.int64).copy()
    for a, n in ranges:
        full[a:a + n] += 1
    assert (full == 1).all()
    covered = set()
    for c in calls:
        covered |= {id(p) for p in c.program.params}
    for p, o in zip(arena.params, arena.offsets):
        if id(p) in covered or p is table:
            assert cov[o:o + p.numel()].all(), "a parameter of the launch's programs is not fully a destination"
        else:
            assert not cov[o:o + p.numel()].any()
    assert sum(int(n) for _, n in ranges) == arena.numel - int(cov.sum())


def test_png_writer_round_trip(tmp_path):
    """inference.write_png (what evaluate_images saves, eval.py:166) with the standard library only: signature, IHDR, CRCs
    and the pixels survive a round trip through read_png; sizes that are no multiple of anything."""
    from hypernerf_torch_amd.inference import read_png, write_png
    rs = np.random.RandomState(5)
    for h, w in ((1, 1), (7, 13), (64, 48)):
        img = rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        path = os.path.join(str(tmp_path), f"{h}x{w}.png")
        write_png(path, torch.from_numpy(img))
        raw = open(path, "rb").read()
        assert raw[:8] == b"\x89PNG\r\n\x1a\n" and raw[12:16] == b"IHDR" and raw[-8:-4] == b"IEND"
        assert np.array_equal(read_png(path), img)
    with pytest.raises(ValueError):
        write_png(os.path.join(str(tmp_path), "bad.png"), np.zeros((4, 4), dtype=np.uint8))


def test_ray_metadata_is_a_dict_that_converts_on_first_use():
    """model_utils.RayMetadata (round 6): what prepare_ray_dict hands out for fp32 ray rows on the GPU — the reference's
    four metadata keys over ONE int64 tensor (model_utils.py:389-398), converted on first access unless the model's step
    head has done it in its own launch.  Dict semantics a caller may rely on: item access, get with default, items /
    values, iteration, copy, rebuilding from pairs (DDP's input scatter); CPU rows keep the plain dict."""
    from hypernerf_torch_amd.hypernerf.model_utils import RayMetadata, prepare_ray_dict, extract_rays_batch
    col = torch.tensor([3.0, 7.9, 0.0, 99.0])
    md = RayMetadata(col)
    assert not md.converted() and set(md) == {"warp", "camera", "appearance", "time"} and len(md) == 4
    assert md.get("hyper_point") is None and md.get("hyper_point", 5) == 5 and "hyper_point" not in md
    idx = md["time"]
    assert md.converted() and idx.dtype == torch.int64 and idx.tolist() == [3, 7, 0, 99]       # truncation, as .type(torch.long)
    assert md["warp"] is idx and all(v is idx for v in md.values()) and dict(md.items())["camera"] is idx
    md2 = RayMetadata(col)
    buf = torch.tensor([1, 2, 3, 4])
    md2.set_converted(buf)                                  # what NerfModel.forward does before its step-head launch
    assert md2.converted() and md2["appearance"] is buf
    md3 = RayMetadata(list(md.items()))                    # rebuilt from pairs: a plain, converted dict
    assert md3.converted() and md3["warp"] is idx
    assert md.copy() == dict(md.items()) and type(md.copy()) is dict
    rays = torch.cat([torch.zeros(4, 8), col[:, None]], dim=1)
    rd = prepare_ray_dict(rays)                             # CPU rows: converted eagerly, plain dict (the reference's behaviour)
    assert type(rd["metadata"]) is dict and rd["metadata"]["warp"].tolist() == [3, 7, 0, 99]
    sub = extract_rays_batch({"origins": rays[:, :3], "directions": rays[:, 3:6], "viewdirs": None, "metadata": RayMetadata(col)}, 1, 3)
    assert sub["metadata"]["camera"].tolist() == [7, 0]
