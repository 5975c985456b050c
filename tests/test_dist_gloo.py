"""Data-parallel host logic on CPU: 2 processes, gloo backend (the GPU run uses the same code over RCCL).
Covers ray sharding, the single flat gradient all-reduce (with a parameter that received no gradient on one
rank) and the pixel all-gather."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import hypernerf_torch_amd  # noqa: F401
    from hypernerf_torch_amd.dist import GradBucket, all_gather_pixels, shard_range, shard_rays
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rays = torch.arange(11 * 9, dtype=torch.float32).view(11, 9)
        mine = shard_rays(rays)
        lo, hi = shard_range(11, rank, world)
        assert torch.equal(mine, rays[lo:hi])
        a = torch.nn.Parameter(torch.zeros(5, 3))
        b = torch.nn.Parameter(torch.zeros(7))          # no gradient on rank 1 ("unused parameter")
        c = torch.nn.Parameter(torch.zeros(2, 2))
        a.grad = torch.full((5, 3), float(rank + 1))
        if rank == 0:
            b.grad = torch.arange(7, dtype=torch.float32)
        c.grad = torch.full((2, 2), 10.0 * (rank + 1))
        GradBucket([a, b, c]).all_reduce_mean()
        ok = (torch.allclose(a.grad, torch.full((5, 3), 1.5)) and
              torch.allclose(b.grad, torch.arange(7, dtype=torch.float32) / 2) and
              torch.allclose(c.grad, torch.full((2, 2), 15.0)))
        # the same through a ParamArena: gradients already are one buffer, reduced in place (GradBucket fast path)
        from hypernerf_torch_amd import ParamArena
        d = torch.nn.Parameter(torch.ones(3, 3))
        e = torch.nn.Parameter(torch.ones(5))
        arena = ParamArena([d, e])
        d.grad.fill_(float(rank + 1))
        e.grad.copy_(torch.arange(5, dtype=torch.float32) * (rank + 1))
        GradBucket([d, e]).all_reduce_mean()
        ok = ok and torch.allclose(d.grad, torch.full((3, 3), 1.5)) and \
            torch.allclose(e.grad, torch.arange(5, dtype=torch.float32) * 1.5) and \
            arena.attached(d) == 0 and arena.attached(e) == 12 and \
            torch.allclose(arena.grad[:9], torch.full((9,), 1.5))
        px = torch.full((4, 3), float(rank))
        allpx = all_gather_pixels(px)
        ok = ok and allpx.shape == (8, 3) and torch.equal(allpx[:4], torch.zeros(4, 3)) and \
            torch.equal(allpx[4:], torch.ones(4, 3))
        q.put((rank, bool(ok), (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == (0, 6) and res[1][2] == (6, 11)     # contiguous, remainder to the first ranks


def test_shard_range_covers_everything():
    sys.path.insert(0, ROOT)
    import hypernerf_torch_amd  # noqa: F401
    from hypernerf_torch_amd.dist import shard_range
    for n in (1, 7, 8, 1024, 16385):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
