"""MI355X drop-in for the reference's `hypernerf/rigid_body.py`: skew, rp_to_se3, exp_so3, exp_se3, to_homogenous,
from_homogenous (reference: hypernerf/rigid_body.py:21-93).

Same names and argument meaning; every function also takes a BATCH of inputs (leading dimensions), which upstream's
cannot (`w.view(3)`, rigid_body.py:35).  The exponential maps run on the GPU through the SE(3) kernel of the render path
(`hn_se3_apply_forward`, csrc/hn_render.hip — the kernel SE3Field's warp uses, pinned to the reference's one valid
`exp_se3` result by tests/golden G13): the rigid transform is applied to the origin and the three basis vectors, which
yields t and the columns of R.  For the unit screw axis of Modern Robotics eq. 3.88 that the reference's docstrings cite
and that SE3Field passes (w / theta, warping.py:226-232) the result is upstream's.  Every other input upstream accepts is
accepted too and gets the exponential map of the twist [S] theta itself: a non-unit `w` rotates about w / |w| by |w| theta
(upstream's formula returns I + sin(theta)[w] + (1 - cos(theta))[w]^2 there, which is no rotation), and w = 0 is the pure
translation R = I, p = theta v (the same as upstream's).  `STRICT_UNIT_AXIS = True` turns the old refusal of non-unit axes
back on (it costs a device-to-host sync per call; off, nothing here synchronises, so the functions can be captured).
Tensors must live on the GPU: there is no CPU path (the reference hard-codes `.cuda()` here too, rigid_body.py:38, 52, 57).

Two upstream defects are not reproduced: `to_homogenous` RESHAPES (N, 4) into (4, N) instead of transposing it
(rigid_body.py:85-89: correct for one point only, scrambled beyond), and `exp_se3` reshapes `v` likewise (:76).  Here both
are the transposes their comments ask for, which coincide with upstream for the single-point inputs it can take.
"""
from __future__ import annotations

import torch

from .. import _lib as L
from .. import functional as F


def matmul(a, b):
    return torch.matmul(a, b)


def skew(w: torch.Tensor) -> torch.Tensor:
    """(..., 3) -> (..., 3, 3) with skew(w) @ v == w x v (Modern Robotics eq. 3.30; reference rigid_body.py:21-38)."""
    L.require_gpu(w)
    w = w.reshape(*w.shape[:-1], 3) if w.shape[-1] == 3 else w.reshape(3)
    z = torch.zeros_like(w[..., 0])
    rows = [torch.stack([z, -w[..., 2], w[..., 1]], -1), torch.stack([w[..., 2], z, -w[..., 0]], -1),
            torch.stack([-w[..., 1], w[..., 0], z], -1)]
    return torch.stack(rows, -2).float()


def rp_to_se3(r: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    """(..., 3, 3), (..., 3) -> (..., 4, 4) homogeneous transform (reference rigid_body.py:40-53)."""
    L.require_gpu(r, p)
    p = p.reshape(*r.shape[:-2], 3, 1)
    up = torch.cat([r, p], dim=-1)
    low = torch.zeros(*r.shape[:-2], 1, 4, dtype=r.dtype, device=r.device)
    low[..., 0, 3] = 1.0
    return torch.cat([up, low], dim=-2)


STRICT_UNIT_AXIS = False


def _unit_axis(w: torch.Tensor, what: str):
    if not STRICT_UNIT_AXIS:
        return
    n = w.norm(dim=-1)
    if not bool(((n - 1.0).abs() <= 1e-4).all()):
        raise ValueError(f"{what}: the rotation axis must be a unit vector (Modern Robotics eq. 3.51 / 3.88: "
                         "w_hat = w / theta, as SE3Field passes it); normalise it and put the magnitude into theta")


def _magnitudes(theta, n: int, device) -> torch.Tensor:
    tt = torch.as_tensor(theta, dtype=torch.float32, device=device).reshape(-1)
    if tt.numel() == 1:
        return tt.expand(n)
    if tt.numel() != n:
        raise ValueError(f"{tt.numel()} magnitudes for {n} axes")
    return tt


def _rigid(w: torch.Tensor, v: torch.Tensor, theta: torch.Tensor):
    """(R (N,3,3), t (N,3)) of exp([S] theta) for N screw axes through hn_se3_apply_forward: the kernel takes the
    exponential coordinates (w theta, v theta) and a point.  The origin with the twist's v gives t; the three basis vectors
    with v = 0 give the columns of R directly (no (R e_i + t) - t cancellation when |t| is large).  Rows with w theta = 0
    (the kernel divides by |w theta|, as SE3Field does: warping.py:226-232) are the pure translation R = I, t = v theta,
    selected by a device-side mask — no host sync."""
    n = w.shape[0]
    th = theta.reshape(n, 1).float()
    wt, vt = (w.float() * th).contiguous(), (v.float() * th).contiguous()
    still = (wt == 0).all(dim=-1, keepdim=True)                                            # (n, 1)
    e1 = torch.zeros(1, 3, device=w.device)
    e1[0, 0] = 1.0
    wk = torch.where(still, e1.expand(n, 3), wt)                                           # any finite axis: result replaced below
    pts = torch.cat([torch.zeros(1, 3), torch.eye(3)], 0).to(w.device)                     # origin, e1, e2, e3
    vk = torch.cat([vt[:, None, :], torch.zeros(n, 3, 3, device=w.device)], 1)             # v for the origin only
    y = F.se3_apply(wk[:, None, :].expand(n, 4, 3).reshape(-1, 3), vk.reshape(-1, 3),
                    pts[None].expand(n, 4, 3).reshape(-1, 3)).view(n, 4, 3)
    t = torch.where(still, vt, y[:, 0])
    r = y[:, 1:].transpose(1, 2)                                                            # column i = R e_i
    r = torch.where(still[:, :, None], torch.eye(3, device=w.device).expand(n, 3, 3), r)
    return r, t


def exp_so3(w: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """Rodrigues: I + sin(theta) [w] + (1 - cos(theta)) [w]^2 for unit axes w (..., 3), theta (...) or scalar
    (reference rigid_body.py:55-57) -> (..., 3, 3); a non-unit w rotates about w / |w| by |w| theta, w = 0 gives I."""
    L.require_gpu(w)
    lead = w.shape[:-1] if w.dim() > 1 and w.shape[-1] == 3 else ()
    wf = w.reshape(-1, 3)
    _unit_axis(wf, "exp_so3")
    th = _magnitudes(theta, wf.shape[0], w.device)
    r, _ = _rigid(wf, torch.zeros_like(wf), th)
    return r.view(*lead, 3, 3)


def exp_se3(S: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """Exponential map of the screw axis S = (w_hat, v) (..., 6) with magnitude theta -> (..., 4, 4) (Modern Robotics
    eq. 3.88; reference rigid_body.py:59-83, whose own call shape is S (1, 1, 6), theta (1, 1) -> (4, 4))."""
    L.require_gpu(S)
    sf = S.reshape(-1, 6)
    _unit_axis(sf[:, :3], "exp_se3")
    th = _magnitudes(theta, sf.shape[0], S.device)
    r, t = _rigid(sf[:, :3], sf[:, 3:], th)
    out = rp_to_se3(r, t)
    if sf.shape[0] == 1:            # the reference's single-screw call returns a plain (4, 4)
        return out[0]
    return out.view(*S.shape[:-1], 4, 4)


def to_homogenous(v: torch.Tensor) -> torch.Tensor:
    """(N, 1, 3) or (N, 3) points -> (4, N) homogeneous columns (reference rigid_body.py:85-89, as its comment states the
    layout; upstream's reshape equals this transpose for N == 1 only)."""
    L.require_gpu(v)
    ones = torch.ones_like(v[..., :1])
    res = torch.cat([v, ones], dim=-1).reshape(-1, 4)
    return res.transpose(0, 1).contiguous()


def from_homogenous(v: torch.Tensor) -> torch.Tensor:
    """(..., 4) -> (..., 3): v[..., :3] / v[..., -1:] (reference rigid_body.py:91-93)."""
    L.require_gpu(v)
    return v[..., :3] / v[..., -1:]
