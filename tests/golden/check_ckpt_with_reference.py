#!/usr/bin/env python3
"""Checkpoint round trip THROUGH THE REFERENCE'S OWN LOADER, build container only:

    python tests/golden/check_ckpt_with_reference.py

1. this package's `save_ckpt` writes a checkpoint of a NerfModel whose parameters live in a ParamArena; the
   reference's `utils.load_ckpt(nerf, path, model_name='nerf')` (utils/__init__.py:66-88 — imported from
   /root/reference, not restated) loads it into the REFERENCE's `hypernerf.models.NerfModel`; every tensor of the two
   state dicts must be equal, and the reference's `extract_model_state_dict` must return exactly our key set;
2. the other direction: the reference model's weights, stored the way Lightning stores them (`state_dict` under the
   `nerf.` prefix, train.py:48), are loaded by this package's `load_ckpt` into an arena-backed model; equal again;
3. the same for the legacy nerf_pl `NeRF` (models/nerf.py) under the prefix eval.py uses.
Import shims: the ones of make_golden.py plus import-only stubs for `torchvision.transforms`, `cv2` and `PIL`
(the reference's `utils/__init__.py` pulls in its visualisation helpers; none of them is on the loader's path).
Prints one JSON line with what was compared; tests/test_host_api.py::test_ckpt_round_trip_through_the_reference_loader
runs it whenever /root/reference is present (it never is on the GPU box)."""
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("HN_REFERENCE", "/root/reference")
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

import hypernerf_torch_amd as HN                                    # noqa: E402  (before the reference shadows `models`)
from hypernerf_torch_amd.hypernerf import models as my_models       # noqa: E402
from hypernerf_torch_amd.models import nerf as my_legacy            # noqa: E402
from hypernerf_torch_amd import utils as my_utils                   # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_stub("immutabledict", immutabledict=dict)
_stub("torchsummary")
_stub("torchsearchsorted", searchsorted=lambda a, v, side="left": torch.searchsorted(a, v, right=(side == "right")))
tv = _stub("torchvision")
tv.transforms = _stub("torchvision.transforms")
_stub("cv2", COLORMAP_JET=2)
pil = _stub("PIL")
pil.Image = _stub("PIL.Image")
torch.Tensor.cuda = lambda self, *a, **k: self
sys.path.insert(0, REF)

import utils as R_utils                                              # noqa: E402  the reference's loader
from hypernerf import models as R_models                             # noqa: E402
from models import nerf as R_nerf                                    # noqa: E402

EMB = {"warp": list(range(100)), "camera": [0], "appearance": list(range(100)), "time": list(range(100))}
KW = dict(hyper_slice_method="bendy_sheet", use_nerf_embed=True, use_alpha_cond=True, view_fourier_dim=6)


def same(a, b, what):
    assert set(a) == set(b), (what, set(a) ^ set(b))
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k].detach().cpu().float(), b[k].detach().cpu().float()), (what, k)
    return len(a)


def main():
    report = {}
    with tempfile.TemporaryDirectory() as tmp:
        # 1. ours -> reference loader
        torch.manual_seed(11)
        mine = my_models.NerfModel(EMB, **KW)
        arena = HN.ParamArena(mine.parameters())
        with torch.no_grad():
            arena.data.add_(torch.randn_like(arena.data) * 0.01)    # not the initial weights of anyone
        path = my_utils.save_ckpt(mine, os.path.join(tmp, "epoch=3.ckpt"), model_name="nerf", epoch=3, global_step=77)
        theirs = R_models.NerfModel(EMB, **KW)
        before = {k: v.clone() for k, v in theirs.state_dict().items()}
        picked = R_utils.extract_model_state_dict(path, model_name="nerf")
        assert set(picked) == set(mine.state_dict())
        R_utils.load_ckpt(theirs, path, model_name="nerf")
        report["ours_to_reference_tensors"] = same(theirs.state_dict(), mine.state_dict(), "ours -> reference")
        assert any(not torch.equal(before[k], v) for k, v in theirs.state_dict().items())
        # prefixes_to_ignore goes through the reference's loader too
        theirs2 = R_models.NerfModel(EMB, **KW)
        keep = theirs2.warp_field.mlp.linears[0].weight.clone()
        R_utils.load_ckpt(theirs2, path, model_name="nerf", prefixes_to_ignore=["warp_field"])
        assert torch.equal(theirs2.warp_field.mlp.linears[0].weight, keep)
        assert torch.equal(theirs2.hyper_sheet_mlp.mlp.linears[0].weight.float(),
                           mine.hyper_sheet_mlp.mlp.linears[0].weight.detach().float())

        # 2. reference -> ours (Lightning layout)
        torch.manual_seed(12)
        src = R_models.NerfModel(EMB, **KW)
        lightning = {"epoch": 5, "global_step": 9, "state_dict": {"nerf." + k: v.clone() for k, v in src.state_dict().items()}}
        lightning["state_dict"]["loss.dummy"] = torch.zeros(1)
        p2 = os.path.join(tmp, "epoch=5.ckpt")
        torch.save(lightning, p2)
        dst = my_models.NerfModel(EMB, **KW)
        arena2 = HN.ParamArena(dst.parameters())
        my_utils.load_ckpt(dst, p2, "nerf")
        report["reference_to_ours_tensors"] = same(dst.state_dict(), src.state_dict(), "reference -> ours")
        assert arena2.attached(dst.warp_field.mlp.linears[0].weight) is not None

        # 3. legacy nerf_pl NeRF both ways (eval.py loads 'nerf_coarse' / 'nerf_fine' prefixes in nerf_pl; any prefix works)
        torch.manual_seed(13)
        lm = my_legacy.NeRF()
        p3 = my_utils.save_ckpt(lm, os.path.join(tmp, "legacy.ckpt"), model_name="nerf_coarse")
        lr = R_nerf.NeRF()
        R_utils.load_ckpt(lr, p3, model_name="nerf_coarse")
        report["legacy_ours_to_reference_tensors"] = same(lr.state_dict(), lm.state_dict(), "legacy ours -> reference")
        torch.save({"state_dict": {"nerf_coarse." + k: v + 0.5 for k, v in lr.state_dict().items()}}, p3)
        my_utils.load_ckpt(lm, p3, "nerf_coarse")
        report["legacy_reference_to_ours_tensors"] = same(lm.state_dict(), {k: v + 0.5 for k, v in lr.state_dict().items()},
                                                          "legacy reference -> ours")
    report["reference_loader"] = os.path.join(REF, "utils/__init__.py") + ":66-88"
    print(json.dumps(report))


if __name__ == "__main__":
    main()
