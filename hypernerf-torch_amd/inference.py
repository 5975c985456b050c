"""Evaluation-side drivers of the render path (reference: eval.py:77-101 `batched_inference`, the image assembly of
eval.py:139-166 and train.py:166-182).

`batched_inference` keeps the reference's signature and result (every per-level tensor concatenated over chunks).
`render_image` is what an MI355X wants instead: it keeps only the per-ray outputs an image needs (the (B,S,7) point
tensors the reference also concatenates are pure HBM traffic), uses chunks that fill the GPU, and shards the rays
of the image over the ranks of a data-parallel job (one all-gather of pixels, dist.all_gather_pixels).
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, Sequence

import torch

from . import dist as hdist
from .hypernerf import model_utils

_EXTRA = {'nerf_alpha': None, 'warp_alpha': None, 'hyper_alpha': None, 'hyper_sheet_alpha': None}


@torch.no_grad()
def batched_inference(model, rays_dict, N_samples=None, N_importance=None, use_disp=False, chunk=1024 * 32,
                      white_back=False) -> Dict:
    """Render all rays of `rays_dict` in chunks of `chunk` rays and concatenate every output.  N_samples,
    N_importance, use_disp and white_back are accepted and ignored exactly as the reference ignores them (the model
    carries those settings); unlike the reference, `chunk` is honoured (eval.py:84 overwrites it with 1024)."""
    n = rays_dict["origins"].shape[0]
    pieces = defaultdict(list)
    for start in range(0, n, chunk):
        out = model(model_utils.extract_rays_batch(rays_dict, start, start + chunk), dict(_EXTRA))
        for level, tensors in out.items():
            pieces[level].append(tensors)
    return {level: model_utils.concat_ray_batch(chunks) for level, chunks in pieces.items()}


@torch.no_grad()
def render_image(model, rays: torch.Tensor, chunk: int = 16384, level: str = 'fine',
                 keys: Sequence[str] = ('rgb', 'depth', 'acc'), group=None) -> Dict[str, torch.Tensor]:
    """rays (N, 8|9) of one image -> {key: (N, ...)} for the requested per-ray outputs of `level`.
    With torch.distributed initialised every rank renders a contiguous 1/world slice and the pixels are gathered
    with one all-gather per key (slices are padded to equal length)."""
    n = rays.shape[0]
    world = torch.distributed.get_world_size(group) if hdist.dist.is_available() and hdist.dist.is_initialized() else 1
    rank = torch.distributed.get_rank(group) if world > 1 else 0
    lo, hi = hdist.shard_range(n, rank, world)
    per = -(-n // world)                                  # padded slice length, equal on all ranks
    mine = rays[lo:hi]
    outs = {k: [] for k in keys}
    for start in range(0, mine.shape[0], chunk):
        rd = model_utils.prepare_ray_dict(mine[start:start + chunk])
        res = model(rd, dict(_EXTRA))[level]
        for k in keys:
            outs[k].append(res[k])
    result = {}
    for k in keys:
        if outs[k]:
            x = torch.cat(outs[k], dim=0)
        else:       # a rank without rays (more ranks than rays): shape from a zero-length template
            x = torch.zeros((0, 3) if k == 'rgb' else (0,), dtype=torch.float32, device=rays.device)
        if world > 1:
            pad = per - x.shape[0]
            if pad:
                x = torch.cat([x, x.new_zeros((pad,) + tuple(x.shape[1:]))], dim=0)
            full = hdist.all_gather_pixels(x, group)
            parts = []
            for r in range(world):
                a, b = hdist.shard_range(n, r, world)
                parts.append(full[r * per:r * per + (b - a)])
            x = torch.cat(parts, dim=0)
        result[k] = x
    return result


def write_png(path: str, img8) -> None:
    """8-bit RGB (H, W, 3) uint8 array / tensor -> a PNG file, with the standard library only (zlib + struct): what
    eval.py:166 does through imageio (`imageio.imwrite(f'{i:03d}.png', img_pred_)`), without an image library in the
    package's dependencies.  Truecolour, 8 bits per channel, no interlace, filter type 0 on every scanline."""
    import struct
    import zlib
    import numpy as np
    a = np.ascontiguousarray(img8.cpu().numpy() if isinstance(img8, torch.Tensor) else img8, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("write_png: (H, W, 3) uint8")
    h, w = a.shape[:2]
    raw = np.concatenate([np.zeros((h, 1), dtype=np.uint8), a.reshape(h, w * 3)], axis=1).tobytes()      # filter byte 0 per row

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def read_png(path: str):
    """The inverse of write_png for its own files (tests): (H, W, 3) uint8 numpy array."""
    import struct
    import zlib
    import numpy as np
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] != (zlib.crc32(tag + body) & 0xffffffff):
            raise ValueError("PNG chunk CRC mismatch")
        if tag == b"IHDR":
            w, h, depth, colour = struct.unpack(">IIBB", body[:10])
            if (depth, colour) != (8, 2):
                raise ValueError("read_png reads 8-bit truecolour only")
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w)
    if rows[:, 0].any():
        raise ValueError("read_png reads filter type 0 only")
    return rows[:, 1:].reshape(h, w, 3).copy()


@torch.no_grad()
def evaluate_images(model, images, chunk: int = 16384, white_back: bool = False, save_dir=None, group=None,
                    image_format: str = "png") -> Dict:
    """The per-image loop of the reference's eval.py:145-178 on the GPU: for every sample {'rays': (H*W, 8|9),
    'rgbs': (H*W, 3) optional, 'hw': (H, W) optional} render the fine level in chunks, form the (H, W, 3) image,
    its 8-bit version (eval.py:165 `(img*255).astype(uint8)`) and, when ground truth is present, the PSNR
    (metrics.psnr, eval.py:169-172).  Returns {'images': [uint8 (H,W,3) CPU tensors], 'depths': [...],
    'psnrs': [float], 'mean_psnr': float | None}.  With `save_dir` the 8-bit frames are also written as `{i:03d}.png`
    (eval.py:166; `write_png`: standard library only — round 6; `image_format="ppm"` keeps the binary PPM of rounds 2-5).
    The reference's GIF of all frames (eval.py:172, imageio.mimsave) is not written: the frames are in the result.
    `white_back` is accepted and ignored as in the reference's batched_inference (eval.py:77-85)."""
    if image_format not in ("png", "ppm"):
        raise ValueError("image_format: 'png' or 'ppm'")
    from .losses import psnr as _psnr
    imgs, depths, psnrs = [], [], []
    for i, sample in enumerate(images):
        rays = sample['rays']
        res = render_image(model, rays, chunk=chunk, keys=('rgb', 'depth'), group=group)
        n = rays.shape[0]
        h, w = sample.get('hw', (1, n))
        img = res['rgb'].view(h, w, 3)
        img8 = (img * 255).to(torch.uint8).cpu()          # truncation, like numpy's astype(uint8) on [0, 1] data
        imgs.append(img8)
        depths.append(torch.nan_to_num(res['depth'].view(h, w)).cpu())
        if sample.get('rgbs') is not None:
            gt = sample['rgbs'].to(img.device).view(h, w, 3)
            psnrs.append(float(_psnr(gt, img)))
        if save_dir is not None:
            import os
            os.makedirs(save_dir, exist_ok=True)
            if image_format == "png":
                write_png(os.path.join(save_dir, f"{i:03d}.png"), img8)
            else:
                with open(os.path.join(save_dir, f"{i:03d}.ppm"), "wb") as f:
                    f.write(f"P6 {w} {h} 255\n".encode() + img8.numpy().tobytes())
    return {'images': imgs, 'depths': depths, 'psnrs': psnrs,
            'mean_psnr': (sum(psnrs) / len(psnrs)) if psnrs else None}
