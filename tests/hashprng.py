"""Counter-based hash PRNG (splitmix64) shared by the golden generator and the tests.

Fixtures carry SEEDS, not megabytes of weights: the generator (which imports the
reference) and the tests (which never do) both rebuild identical tensors from
(seed, name) with this module.  Pure numpy, platform independent.
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, name: str, n: int) -> np.ndarray:
    """n float32 values in [0,1), deterministic in (seed, name)."""
    key = np.uint64((zlib.crc32(name.encode()) << 32) ^ (seed & 0xFFFFFFFF))
    with np.errstate(over="ignore"):
        ctr = _splitmix64(np.arange(n, dtype=np.uint64) + _splitmix64(np.array([key], dtype=np.uint64))[0])
    return ((ctr >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))).astype(np.float32)


def uniform(seed, name, shape, lo=-1.0, hi=1.0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    v = uniform01(seed, name, n) * np.float32(hi - lo) + np.float32(lo)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def normal(seed, name, shape) -> torch.Tensor:
    """Box-Muller on two hash streams (fp64 math, rounded to fp32)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(seed, name + "#a", n).astype(np.float64)
    u2 = uniform01(seed, name + "#b", n).astype(np.float64)
    v = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def fill_state_dict(shapes: dict, seed: int, gain: float = 1.0) -> dict:
    """Deterministic weights for a {name: shape} map.

    2-D `*.weight` (out,in): U(-a,a), a = gain*sqrt(6/(in+out)) (xavier range);
    `*.bias`: U(-0.1,0.1); `*.embed.weight`: U(-0.5,0.5).  `logit_layer`s use the same
    xavier range (NOT the reference's 1e-4/1e-5 output inits) so every branch carries
    signal large enough to show errors — except the output layers of the warp field and
    the hyper sheet inside a full model (`warp_field.` / `hyper_sheet_mlp.` prefixes), which
    are scaled by 0.02: their outputs are re-encoded with sin(2^9 x) by the template, and
    O(1) displacements would turn 1-ulp differences into percent-level chaos.
    """
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        if name.endswith("embed.weight"):
            out[name] = uniform(seed, name, shp, -0.5, 0.5)
        elif name.endswith("weight") and len(shp) == 2:
            a = gain * float(np.sqrt(6.0 / (shp[0] + shp[1])))
            if name.startswith(("warp_field.mlp.logit_layer", "hyper_sheet_mlp.mlp.logit_layer")):
                a *= 0.02
            out[name] = uniform(seed, name, shp, -a, a)
        elif name.startswith(("warp_field.mlp.logit_layer", "hyper_sheet_mlp.mlp.logit_layer")):
            out[name] = uniform(seed, name, shp, -0.002, 0.002)
        else:
            out[name] = uniform(seed, name, shp, -0.1, 0.1)
    return out
