"""Helpers shared by the GPU parity tests."""
import atexit
import json
import os

import numpy as np
import torch

import hashprng as H

DEV = "cuda:0"
EMB = {"warp": list(range(100)), "camera": [0], "appearance": list(range(100)), "time": list(range(100))}


def rays_for(seed, b, n_img=100):
    o = H.uniform(seed, "rays_o", (b, 3), -1.0, 1.0)
    d = H.uniform(seed, "rays_d", (b, 3), -1.0, 1.0)
    d = d / d.norm(dim=-1, keepdim=True) * H.uniform(seed, "rays_dn", (b, 1), 0.7, 1.6)
    idx = (H.uniform01(seed, "rays_idx", b) * n_img).astype(np.int64)
    return o, d, torch.from_numpy(idx)


def oracle_threads(cap=64):
    """Threads for a CPU-oracle run inside a GPU test: the affinity mask capped by the cgroup CPU quota (the pool's GPU
    boxes show 256 CPUs under a 16-CPU quota; a thread pool sized by the mask runs several times slower)."""
    from oracle_train import usable_cores
    return max(1, min(int(cap), usable_cores()))


def load_hash(module, seed):
    sd = module.state_dict()
    new = H.fill_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed)
    module.load_state_dict(new)
    return new


# Every comparison made through the helpers below is logged (what, measured error, bound) and written at exit to
# gpurun_out/parity_errors.json when HN_PARITY_REPORT is set: the numbers behind the tolerances quoted in DESIGN.md.
_REPORT = []


def _record(what, kind, err, bound):
    _REPORT.append({"what": what, "kind": kind, "err": float(err), "bound": float(bound)})


def _write_report():
    path = os.environ.get("HN_PARITY_REPORT")
    if path and _REPORT:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "w") as f:
                json.dump(_REPORT, f, indent=0)
        except OSError:
            pass


atexit.register(_write_report)


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


FLOOR_FRAC = 0.05      # elements below 5 % of the tensor's scale are compared absolutely (at tol x that floor)


def assert_close(a, b, tol, what, elementwise=True):
    """The north-star's '<= 1e-4 rel', ELEMENT-wise (elementwise=False: error over the tensor's scale, the bf16
    throughput mode's coarse structural guard): |a-b| <= tol * max(|b|, FLOOR_FRAC * scale) for every element,
    scale = max(1, max|b|) — relative to the element itself wherever it is not near zero, and an absolute
    tol * 0.05 * scale (5e-6 for tol 1e-4 on an O(1) tensor) for the elements that are.  The measured worst ratios
    of a GPU run are written to HN_PARITY_REPORT (profiles/r02_parity_errors.json)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.isfinite(a).all(), what
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    if not elementwise:
        err = float((a - b).abs().max()) / scale if a.numel() else 0.0
        _record(what, "tensor-scale", err, tol)
        assert err <= tol, f"{what}: max abs err / {scale:.3g} = {err:.3e} > {tol:.1e}"
        return
    denom = torch.clamp(b.abs(), min=FLOOR_FRAC * scale)
    ratio = float(((a - b).abs() / denom).max()) if a.numel() else 0.0
    _record(what, f"element-wise rel (floor {FLOOR_FRAC:g} x scale)", ratio, tol)
    assert ratio <= tol, f"{what}: max |a-b| / max(|ref|, {FLOOR_FRAC:g} x {scale:.3g}) = {ratio:.3e} > {tol:.1e}"


def assert_rel_close(a, b, tol, floor, what):
    """ELEMENT-wise relative bound with an absolute floor: |a-b| <= tol * max(|b|, floor) for every element — the
    north-star's '<= 1e-4 rel' taken literally wherever the reference value is not near zero (|b| >= floor)."""
    a = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.isfinite(a).all(), what
    ratio = float(((a - b).abs() / torch.clamp(b.abs(), min=floor)).max()) if a.numel() else 0.0
    _record(what, f"element-wise rel (floor {floor:g})", ratio, tol)
    assert ratio <= tol, f"{what}: max |a-b| / max(|ref|, {floor:g}) = {ratio:.3e} > {tol:.1e}"


def assert_grad_close(g, ref, tol, what, frobenius=False):
    """gradients: max error measured against the largest reference entry of the tensor; with
    frobenius=True the relative L2 error of the whole tensor (bf16 mode: a single ReLU that flips under
    bf16 rounding moves one summand of a small-batch gradient by O(1), so max-norm is meaningless there)."""
    if ref is None:
        assert g is None or float(g.abs().max()) == 0.0, what
        return
    assert g is not None, what + " missing"
    g = g.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert torch.isfinite(g).all(), what
    if frobenius:
        err = float((g - ref).norm() / (ref.norm() + 1e-30))
        _record(what, "gradient rel L2", err, tol)
        assert err <= tol, f"{what}: relative L2 err {err:.3e} > {tol:.1e}"
        return
    err = float((g - ref).abs().max())
    scale = float(ref.abs().max()) + 1e-12
    _record(what, "gradient max / max|ref|", err / scale, tol)
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} > {tol:.1e} * {scale:.3g}"
