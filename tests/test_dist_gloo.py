"""Data-parallel host logic on CPU: 2 processes, gloo backend (the GPU run uses the same code over RCCL).
Covers ray sharding, the single flat gradient all-reduce (with a parameter that received no gradient on one
rank), the two-bucket overlapped reduction (GradSync) and the pixel all-gather."""
import datetime
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
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
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
        # gloo stages through host memory: the collective cannot be recorded into a HIP graph, so TrainStep / bench.py
        # keep it out of the step graph on this backend (round 4: with nccl = RCCL it is captured with the step)
        from hypernerf_torch_amd.dist import collective_capturable
        ok = ok and collective_capturable() is False
        # force=True issues the collective whatever the group size (what lets ONE RCCL rank exercise the captured
        # all-reduce on a one-GPU box); a plain call in a 2-rank group reduces as before
        arena.grad.fill_(float(rank + 1))
        arena.all_reduce_sum(force=True)
        ok = ok and torch.allclose(arena.grad, torch.full_like(arena.grad, 3.0))
        px = torch.full((4, 3), float(rank))
        allpx = all_gather_pixels(px)
        ok = ok and allpx.shape == (8, 3) and torch.equal(allpx[:4], torch.zeros(4, 3)) and \
            torch.equal(allpx[4:], torch.ones(4, 3))
        # GradSync (dist.py): the all-reduce in two buckets around the held weight-gradient launch.  A module whose
        # parameter names follow NerfModel's (the template networks are the tail of the arena) gives the split;
        # `run_held` stands for the held launch: it writes into the HEAD of the buffer after bucket 0's all-reduce
        # has been issued, and must be part of what bucket 1 reduces.
        from hypernerf_torch_amd import machine
        from hypernerf_torch_amd.dist import GradSync

        class Toy(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.warp_field = torch.nn.Linear(3, 4)
                self.nerf_mlps_coarse = torch.nn.Linear(4, 2)
                self.nerf_mlps_fine = torch.nn.Linear(4, 2)
        toy = Toy()
        ta = ParamArena(toy.parameters())
        sync = GradSync(ta, toy)
        ok = ok and sync.split == 16 and ta.numel == 16 + 2 * (8 + 4)
        ta.grad.fill_(float(rank + 1))
        with sync.splitting():
            inside = machine.WGRAD_SPLIT_OFFSET
        ok = ok and inside == 16 and machine.WGRAD_SPLIT_OFFSET is None
        sync.reduce(lambda: ta.grad[:16].add_(10.0 * (rank + 1)))
        ok = ok and torch.allclose(ta.grad[16:], torch.full((24,), 3.0)) and \
            torch.allclose(ta.grad[:16], torch.full((16,), 33.0))
        # no usable split (template parameters not the tail): one all-reduce, the held launch still runs first
        toy2 = torch.nn.Sequential(torch.nn.Linear(2, 2))
        tb = ParamArena(toy2.parameters())
        s2 = GradSync(tb, toy2)
        tb.grad.fill_(1.0)
        s2.reduce(lambda: tb.grad.add_(float(rank)))
        ok = ok and s2.split is None and torch.allclose(tb.grad, torch.full((tb.numel,), 3.0))
        # a graph capture next to a live process group must not bind other threads (the group's watchdog polls events
        # while the capture is open: graphs._capture_mode, GPU test test_graph_capture_survives_the_process_group_watchdog)
        from hypernerf_torch_amd import graphs
        ok = ok and graphs._capture_mode() == "thread_local"
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


def test_capture_mode_without_a_process_group_is_global():
    sys.path.insert(0, ROOT)
    import hypernerf_torch_amd  # noqa: F401
    from hypernerf_torch_amd import graphs
    assert not dist.is_initialized()
    assert graphs._capture_mode() == "global"


def test_shard_range_covers_everything():
    sys.path.insert(0, ROOT)
    import hypernerf_torch_amd  # noqa: F401
    from hypernerf_torch_amd.dist import shard_range
    for n in (1, 7, 8, 1024, 16385):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
