"""autograd glue between PyTorch tensors and the C-ABI kernels.

PyTorch is used for device memory, streams and the autograd graph only; all arithmetic of the
hot path happens in the HIP kernels.  Every function here raises if handed a CPU tensor.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from .arena import ParamArena
from .machine import MlpRunner, PendingWgrad, Program, ResolvedWgrad, launch_resolved_wgrads, resolve_pending

_PRECISION = os.environ.get("HN_PRECISION", "bf16")


def set_precision(p: str):
    """'bf16' (bf16 MFMA operands, fp32 accumulate; throughput mode), 'fp32' (fp32 MFMA; parity mode) or — opt-in —
    'bf16s8': 'bf16' with the training stash (what only the weight gradient reads) kept in 8 bits, include/hn_kernels.h
    HN_MODE_BF16_S8: same forward, same input gradients, rounding noise in the weight gradients, half the stash bytes."""
    global _PRECISION
    if p not in ("bf16", "fp32", "bf16s8"):
        raise ValueError(p)
    _PRECISION = p


def get_precision() -> str:
    return _PRECISION


def mode_of(precision: Optional[str] = None) -> int:
    p = precision or _PRECISION
    return L.HN_MODE_BF16 if p == "bf16" else L.HN_MODE_BF16_S8 if p == "bf16s8" else L.HN_MODE_F32


# --------------------------------------------------------------------------------------------
# fused MLP programs
# --------------------------------------------------------------------------------------------
class ProgramCall:
    """A compiled program + how its sources / outputs map to tensors."""

    def __init__(self, program: Program, src_per_ray: Sequence[bool], dst_widths: Sequence[int],
                 grad_srcs: Sequence[Tuple[str, int]], gather_src: Optional[int] = None,
                 bwd_src_from_out: Optional[Dict[int, int]] = None,
                 fill_from_gather: Optional[Tuple[int, int]] = None):
        # grad_srcs[i] describes backward source 4+i: ('g', k) = gradient of output k (zeros if autograd has none),
        # ('go', k) = the same but simply absent when autograd has none, ('y', k) = output k
        # gather_src: index of a per-ray source given as (table, ray indices): the kernels read table[idx[ray]]
        # bwd_src_from_out: {source index: output index} — sources the forward publishes itself (no pointer) and
        # the backward reads back from the forward's output tensor (fused level programs)
        # fill_from_gather: (output index, first column) — after the launch the gathered rows table[idx[ray]] are
        # copied into those columns of every point of the ray (axis-aligned hyper coordinates as part of
        # `warped_points`, models.py:533-534; a pure copy, no gradient flows back through these columns)
        self.program = program
        self.runner = MlpRunner(program)
        self.src_per_ray = list(src_per_ray)
        self.dst_widths = list(dst_widths)
        self.grad_srcs = list(grad_srcs)
        self.gather_src = gather_src
        self.bwd_src_from_out = dict(bwd_src_from_out or {})
        self.fill_from_gather = fill_from_gather
        self.cache = {}


class _ProgramFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, call: ProgramCall, mode: int, samples_per_ray: int, n_src: int, training: bool, gather_idx,
                *tensors):
        srcs = list(tensors[:n_src])
        L.require_gpu(*[s for s in srcs if s is not None])
        first = next(s for s in srcs if s is not None)
        device = first.device
        n_points = None
        flat_srcs = []
        for i, (s, per_ray) in enumerate(zip(srcs, call.src_per_ray)):
            if s is None:
                flat_srcs.append(None)
                continue
            if i == call.gather_src:
                if gather_idx is None or s.dim() != 2:
                    raise L.HnError("a gathered source needs a (rows, dim) table and int64 ray indices")
                flat_srcs.append((s.detach().contiguous(), True, gather_idx.reshape(-1).to(torch.int64).contiguous()))
                continue
            s2 = s.detach()
            if s2.dim() != 2 or s2.stride(-1) != 1:          # row-strided 2-D views are read in place (ld = stride)
                s2 = s2.reshape(-1, s.shape[-1]).contiguous()
            if not per_ray:
                n_points = s2.shape[0] if n_points is None else n_points
                if s2.shape[0] != n_points:
                    raise L.HnError("per-point sources disagree on the number of points")
            flat_srcs.append((s2, per_ray))
        if n_points is None:
            raise L.HnError("a program needs at least one per-point source")
        if samples_per_ray <= 0 or n_points % samples_per_ray:
            raise L.HnError(f"{n_points} points are not a whole number of rays of {samples_per_ray} samples")
        n_rays = n_points // samples_per_ray
        for i, fs in enumerate(flat_srcs):      # the kernels index per-ray sources and gather indices by ray, unchecked
            if fs is None:
                continue
            if len(fs) > 2 and fs[2].numel() != n_rays:
                raise L.HnError(f"gather index holds {fs[2].numel()} rows for {n_rays} rays")
            if len(fs) == 2 and fs[1] and fs[0].shape[0] != n_rays:
                raise L.HnError(f"per-ray source {i} holds {fs[0].shape[0]} rows for {n_rays} rays")
        outs = [torch.empty(n_points, w, dtype=torch.float32, device=device) for w in call.dst_widths]
        stash, masks = call.runner.forward(mode, n_points, samples_per_ray, flat_srcs, outs, training)
        if call.fill_from_gather is not None:
            k, c0 = call.fill_from_gather
            table, _, gidx = flat_srcs[call.gather_src]
            safe = gidx.clamp(0, table.shape[0] - 1)
            rows = table.index_select(0, safe)
            rows.masked_fill_((safe != gidx)[:, None], float("nan"))    # the kernels stage NaN for such a ray: so does the copy
            outs[k].view(-1, samples_per_ray, outs[k].shape[1])[:, :, c0:c0 + table.shape[1]] = rows[:, None, :]
        ctx.call, ctx.mode, ctx.spr, ctx.n_src, ctx.n_points = call, mode, samples_per_ray, n_src, n_points
        ctx.flat_srcs = flat_srcs
        ctx.tables = {call.gather_src: srcs[call.gather_src]} if call.gather_src is not None else {}
        ctx.src_shapes = [None if s is None else s.shape for s in srcs]
        ctx.src_tags = [None if s is None else getattr(s, "_hn_embed", None) for s in srcs]
        ctx.stash, ctx.masks = stash, masks
        # detached aliases: the backward only needs the values.  Keeping the returned tensors themselves on ctx would
        # close a reference cycle output -> grad_fn (this node) -> ctx -> output that neither Python's collector nor
        # autograd can break: every eager training step would leak its outputs and sources (~10 MB at config 2)
        ctx.outs = [o.detach() for o in outs]
        # outputs nobody differentiates (warped_points of a level program: P x 7 floats) arrive as None in backward,
        # not as a zero fill launched by autograd; the ('go', k) sources are simply absent then
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        call: ProgramCall = ctx.call
        if ctx.stash is None:
            raise L.HnError("backward through a program that ran without training=True")
        bsrcs = list(ctx.flat_srcs) + [None] * (4 - len(ctx.flat_srcs))
        for i, k in call.bwd_src_from_out.items():
            bsrcs[i] = (ctx.outs[k], False)
        for kind, k in call.grad_srcs:
            if kind in ("g", "go"):
                g = gouts[k]
                if g is None:
                    if kind == "go":
                        bsrcs.append(None)
                        continue
                    g = torch.zeros_like(ctx.outs[k])
                bsrcs.append((g if (g.dim() == 2 and g.stride(-1) == 1) else g.contiguous(), False))
            else:
                bsrcs.append((ctx.outs[k], False))
        # parameters attached to a ParamArena: accumulate into its gradient buffer, return nothing through autograd
        prog = call.program
        base = 6 + ctx.n_src
        wants = [ctx.needs_input_grad[base + j] for j in range(len(prog.params))]
        target = ParamArena.lookup(prog.params) if all(wants) else None
        # the gathered per-ray source (GLO table): its gradient is reduced and scattered inside the backward machine
        embed, d_table, table_ret = None, None, None
        gs = call.gather_src
        gather_cols = {c: sl for (si, c), sl in prog.dsrc_map.items() if si == gs} if gs is not None else {}
        if gs is not None and ctx.needs_input_grad[6 + gs] and gather_cols:
            table = ctx.tables[gs]
            tt = ParamArena.lookup([table]) if isinstance(table, torch.nn.Parameter) else None
            if tt is not None:
                d_table = table.grad                      # the arena's view: accumulate in place
            else:
                d_table = table_ret = torch.zeros(table.shape, dtype=torch.float32, device=table.device)
            if ctx.spr % 32 == 0:
                embed = (d_table, ctx.flat_srcs[gs][2], gs)
                # batched weight gradient with partial slabs: the table's gradient goes the same way — per-block rows
                # stored by the backward machine, summed per table row in a fixed order by the reduce launch
                # (machine.WGRAD_PARTIALS): with it the whole gradient of a step is bit-reproducible
                from . import machine as _M
                if (BATCH_WGRADS and target is not None and table_ret is None and _M.WGRAD_PARTIALS and table.shape[1] <= 32
                        and call.runner.effective_mode(ctx.mode) != L.HN_MODE_BF16_S8):
                    nblk = (ctx.n_points + 31) // 32
                    part = torch.empty(nblk * table.shape[1], dtype=torch.float32, device=table.device)
                    embed = embed + (part,)
        others = [i for i, shp in enumerate(ctx.src_shapes)
                  if shp is not None and i != gs and ctx.needs_input_grad[6 + i]
                  and any(si == i for (si, _c) in prog.dsrc_map)]
        want_dsrc = bool(others) or (d_table is not None and embed is None)
        dsrc, flat = call.runner.backward(ctx.mode, ctx.n_points, ctx.spr, bsrcs, ctx.stash, ctx.masks,
                                          grad_target=(target[0].grad, target[1]) if target else None,
                                          defer=BATCH_WGRADS, embed=embed, want_dsrc=want_dsrc)
        if isinstance(flat, list):          # deferred: this program's share(s) of the batched weight-gradient launch
            if embed is not None and len(embed) > 3:
                mask = 0
                for c_ in gather_cols:
                    mask |= 1 << c_
                flat[0].embed = {"partial": embed[3], "idx": embed[1], "grad": d_table, "n_blocks": (ctx.n_points + 31) // 32,
                                 "spr": ctx.spr, "col_mask": mask}
            for pend in flat:
                _defer_wgrad(pend)
            flat = None
        ctx.stash = ctx.masks = None
        src_grads: List[Optional[torch.Tensor]] = []
        for i, shp in enumerate(ctx.src_shapes):
            if shp is None or not ctx.needs_input_grad[6 + i]:
                src_grads.append(None)
                continue
            if i == gs:
                if d_table is not None and embed is None:     # ragged rays: per-ray reduction + scatter as two launches
                    width = shp[-1]
                    n_rays = ctx.n_points // ctx.spr
                    rows = sum_samples(dsrc, gather_cols, n_rays, ctx.spr, width)
                    L.launch("hn_embed_backward", L.ptr(rows), C.c_int(width), C.c_int(0), L.ptr(ctx.flat_srcs[gs][2]),
                             C.c_int(n_rays), C.c_int(1), C.c_int(width), C.c_int(shp[0]), L.ptr(d_table),
                             L.stream_handle())
                src_grads.append(table_ret)
                continue
            cols = {c: s for (si, c), s in prog.dsrc_map.items() if si == i}
            if not cols:
                src_grads.append(None)
                continue
            width = shp[-1]
            if call.src_per_ray[i]:
                n_rays = ctx.n_points // ctx.spr
                if _scatter_embed_grad(ctx.src_tags[i], dsrc, cols, n_rays, ctx.spr, width):
                    src_grads.append(None)       # went straight into the embedding table's arena gradient
                    continue
                g = sum_samples(dsrc, cols, n_rays, ctx.spr, width)
            else:
                key = ("colidx", i)
                idx = call.cache.get(key)
                if idx is None:
                    idx = torch.tensor([cols.get(c, 0) for c in range(width)], dtype=torch.int64, device=dsrc.device)
                    mask = torch.tensor([1.0 if c in cols else 0.0 for c in range(width)], device=dsrc.device)
                    idx = (idx, None if all(c in cols for c in range(width)) else mask)
                    call.cache[key] = idx
                slots = [cols.get(c, -1) for c in range(width)]
                if all(sl == slots[0] + k for k, sl in enumerate(slots)) and slots[0] >= 0:
                    g = dsrc.narrow(1, slots[0], width)   # consecutive rows of the accumulator tile: a view, no launch
                else:
                    g = dsrc.index_select(1, idx[0])      # one gather instead of a copy kernel per column
                    if idx[1] is not None:
                        g = g * idx[1]
            # always the source's own shape (autograd checks it): splitting dim 0 of the row-strided view `narrow`
            # returns is a legal view; anything view() cannot express is copied
            try:
                src_grads.append(g.view(shp))
            except RuntimeError:
                src_grads.append(g.reshape(shp))
        # heads that ADD a source to their result (OutSpec.residual: TranslationField's p + delta, warping.py:90-96):
        # the output gradient reaches that source directly as well
        for ly in prog.layers:
            o = ly.out
            if o is None or o.residual is None or o.wide:
                continue
            rs, rc = o.residual
            if rs >= len(src_grads) or ctx.src_shapes[rs] is None or not ctx.needs_input_grad[6 + rs] or gouts[o.dst] is None:
                continue
            n = prog._rows(ly)
            g_res = gouts[o.dst].reshape(ctx.n_points, -1)[:, o.col:o.col + n]
            width = ctx.src_shapes[rs][-1]
            cur = src_grads[rs]
            cur = torch.zeros(ctx.n_points, width, dtype=torch.float32, device=g_res.device) if cur is None \
                else cur.reshape(ctx.n_points, width).clone()
            cur[:, rc:rc + n] += g_res
            src_grads[rs] = cur.view(ctx.src_shapes[rs])
        if flat is None:
            return (None, None, None, None, None, None, *src_grads, *([None] * len(prog.params)))
        pgrads = call.runner.split_grads(flat)
        out_p = [pgrads[j] if wants[j] else None for j in range(len(prog.params))]
        return (None, None, None, None, None, None, *src_grads, *out_p)


# Arena mode: the weight-gradient kernels only feed arena.grad, which nothing reads before backward() returns, so
# they need not sit between the backward-data kernels.  Two schedules:
#   serial (default): ONE batched launch over all programs of the backward pass, queued as the end-of-backward
#     callback (one global heaviest-first job order, one ramp and one tail).
#   forked (set_wgrad_overlap(True), env HN_WGRAD_OVERLAP=1, bench.py --fork-wgrad): each program's weight-gradient
#     launch goes onto a SIDE stream right behind its own backward-data kernel (event on the main stream -> wait on
#     the side stream) and is joined by the end-of-backward callback; inside a stream capture the fork/join becomes
#     two parallel branches of the HIP graph, which the runtime does execute concurrently on two hardware queues.
#     MEASURED SLOWER on one MI355X (config 2: 2.00 ms against 1.87 ms per step; config 3: 38.29 against 38.39 ms;
#     kernel timeline in profiles/r03_wgrad_fork_trace.txt): forward / backward / weight-gradient workgroups each fill
#     a CU (512 threads x 256 registers, 75-128 KiB of LDS), so the two launches can only share the chip CU by CU —
#     and the weight-gradient stream is limited PER CU (~9-10 B/clk: LDS-DMA in flight over HBM latency), not chip-wide:
#     every CU handed to the backward machine takes its share of the stream rate away, the 8-us compositing kernel
#     between the two waits 230 us for a free CU behind 0.2-ms jobs, and the split launch pays a second ramp and tail.
#     Kept as an option for chips / shapes where that balance differs; per-kernel timing (bench.py's roofline leg,
#     rocprofv3 --stats) is always taken in the serial schedule.
BATCH_WGRADS = True
WGRAD_OVERLAP = os.environ.get("HN_WGRAD_OVERLAP", "0") == "1"
_PENDING: List[PendingWgrad] = []
_PENDING_TASK = [-1]      # autograd graph-task id the pending entries belong to
_FORKED: List[ResolvedWgrad] = []    # launched on the side stream, not yet joined (keeps their stashes alive)
_SIDE_STREAMS: Dict[str, "torch.cuda.Stream"] = {}


_HELD: List[ResolvedWgrad] = []      # bucket-1 shares (machine.WGRAD_SPLIT_OFFSET), launched by flush_held_wgrads()


def set_wgrad_overlap(on: bool):
    """Switch between the forked (side-stream) and the serial (one batched launch) weight-gradient schedule.
    A captured graph keeps the schedule it was captured with."""
    global WGRAD_OVERLAP
    WGRAD_OVERLAP = bool(on)


def _side_stream(device) -> "torch.cuda.Stream":
    key = str(device)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def _fork_wgrad(p: PendingWgrad):
    """Launch one program's weight-gradient jobs on the side stream, ordered behind everything the main stream has
    been given so far (its backward-data kernel wrote the dZ stash the jobs read).  A launch of its own: the jobs are
    sized by this program's bytes; a held bucket (data-parallel split) still waits for flush_held_wgrads()."""
    dev = p.stash.device
    shares = resolve_pending([p])
    _HELD.extend(r for r in shares if r.bucket != 0)
    now = [r for r in shares if r.bucket == 0]
    if not now:
        return
    main, side = torch.cuda.current_stream(dev), _side_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        launch_resolved_wgrads(now)
    _FORKED.extend(now)


def _join_wgrads():
    """Main stream waits for the side stream; only then may the stashes return to the allocator (they were
    allocated on the main stream: freed after the join, every later use is ordered behind the side stream's reads)."""
    if not _FORKED:
        return
    devs = {str(p.stash.device): p.stash.device for p in _FORKED}
    for dev in devs.values():
        torch.cuda.current_stream(dev).wait_stream(_side_stream(dev))
    _FORKED.clear()


def _flush_wgrads():
    pending = list(_PENDING)
    _PENDING.clear()
    shares = resolve_pending(pending) if pending else []
    now = [p for p in shares if p.bucket == 0]
    _HELD.extend(p for p in shares if p.bucket != 0)
    if now:
        launch_resolved_wgrads(now)
    _join_wgrads()


def drop_pending_wgrads():
    """Forget every queued / held weight-gradient share WITHOUT launching it (their stashes may belong to a stream
    capture that was abandoned part-way: training.TrainStep falls back to another schedule after a failed capture)."""
    _PENDING.clear()
    _HELD.clear()
    _FORKED.clear()
    from . import machine as _M
    _M.drop_pending_reduce()


def held_wgrads() -> int:
    return len(_HELD)


def flush_held_wgrads():
    """Launch the weight-gradient jobs the last backward pass held back (data-parallel overlap: the all-reduce of the
    first bucket is already in flight).  A no-op without machine.WGRAD_SPLIT_OFFSET."""
    held = list(_HELD)
    _HELD.clear()
    if held:
        launch_resolved_wgrads(held)


def _defer_wgrad(p: PendingWgrad):
    task = torch._C._current_graph_task_id()
    if (_PENDING or _FORKED) and _PENDING_TASK[0] != task:
        _flush_wgrads()     # left over from a backward pass that did not reach its callback (an exception)
    if not _PENDING and not _FORKED:
        _PENDING_TASK[0] = task
        torch.autograd.Variable._execution_engine.queue_callback(_flush_wgrads)
    if WGRAD_OVERLAP:
        _fork_wgrad(p)
    else:
        _PENDING.append(p)


def run_program(call: ProgramCall, srcs: Sequence[Optional[torch.Tensor]], samples_per_ray: int,
                precision: Optional[str] = None, gather_idx: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, ...]:
    """srcs[call.gather_src], if any, is the (rows, dim) table itself and `gather_idx` the (B,) int64 row per ray."""
    L.load()
    params = call.program.params
    training = torch.is_grad_enabled() and (any(s is not None and s.requires_grad for s in srcs) or
                                            any(p.requires_grad for p in params))
    return _ProgramFn.apply(call, mode_of(precision), int(samples_per_ray), len(srcs), training, gather_idx, *srcs,
                            *params)


_ARANGE: Dict[tuple, torch.Tensor] = {}


def sum_samples(d_points: torch.Tensor, cols: Dict[int, int], n_rays: int, n_samples: int, width: int):
    """(n_rays, width) with out[b, c] = sum_s d_points[b*S + s, slot(c)] — HIP reduction kernel, one launch per
    run of consecutive (column, slot) pairs (the 8 GLO components are one run)."""
    L.load()
    idx = _ARANGE.get((n_rays, d_points.device))
    if idx is None:
        idx = _ARANGE[(n_rays, d_points.device)] = torch.arange(n_rays, dtype=torch.int64, device=d_points.device)
    items = sorted(cols.items())
    out = None
    i = 0
    while i < len(items):
        j = i
        while j + 1 < len(items) and items[j + 1][0] == items[j][0] + 1 and items[j + 1][1] == items[j][1] + 1:
            j += 1
        c0, s0, run = items[i][0], items[i][1], j - i + 1
        tmp = torch.zeros(n_rays, run, dtype=torch.float32, device=d_points.device)
        L.launch("hn_embed_backward", L.ptr(d_points), C.c_int(d_points.shape[1]), C.c_int(s0), L.ptr(idx),
                 C.c_int(n_rays), C.c_int(n_samples), C.c_int(run), C.c_int(n_rays), L.ptr(tmp), L.stream_handle())
        if run == width:
            return tmp
        if out is None:
            out = torch.zeros(n_rays, width, dtype=torch.float32, device=d_points.device)
        out[:, c0:c0 + run] = tmp
        i = j + 1
    return out if out is not None else torch.zeros(n_rays, width, dtype=torch.float32, device=d_points.device)


# --------------------------------------------------------------------------------------------
# GLO embedding
# --------------------------------------------------------------------------------------------
class _EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table: torch.Tensor, idx: torch.Tensor):
        L.require_gpu(table, idx)
        L.load()
        idx = idx.reshape(-1).to(torch.int64).contiguous()
        n, dim = idx.numel(), table.shape[1]
        out = torch.empty(n, dim, dtype=torch.float32, device=table.device)
        L.launch("hn_embed_gather", L.ptr(table.detach().contiguous()), L.ptr(idx), C.c_int(n), C.c_int(dim),
                                    C.c_int(table.shape[0]), L.ptr(out), L.stream_handle())
        ctx.idx, ctx.shape, ctx.table = idx, table.shape, table
        return out

    @staticmethod
    def backward(ctx, g):
        L.load()
        g = g.contiguous()
        target = ParamArena.lookup([ctx.table]) if isinstance(ctx.table, torch.nn.Parameter) else None
        if target is not None:      # scatter-add straight into the arena's view of table.grad
            d_table, ret = ctx.table.grad, None
        else:
            d_table = ret = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        n, dim = g.shape
        L.launch("hn_embed_backward", L.ptr(g), C.c_int(dim), C.c_int(0), L.ptr(ctx.idx), C.c_int(n), C.c_int(1),
                                      C.c_int(dim), C.c_int(ctx.shape[0]), L.ptr(d_table), L.stream_handle())
        return ret, None


def embed_lookup(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    out = _EmbedFn.apply(table, idx)
    # lets a program that consumes this per-ray embedding send its gradient straight to the table (see
    # _scatter_embed_grad); ordinary autograd through _EmbedFn.backward stays valid for every other consumer
    out._hn_embed = (table, idx.reshape(-1).to(torch.int64))
    return out


def _scatter_embed_grad(tag, d_points, cols, n_rays, n_samples, width) -> bool:
    """d_table[idx[b], c] += sum_s d_points[b*S + s, slot(c)] in ONE kernel, when the per-ray source is the output
    of embed_lookup on a table that lives in a ParamArena and all `width` components form one run of slots.
    Replaces a per-ray reduction (+ zero fill) followed by the scatter-add of _EmbedFn.backward."""
    if tag is None:
        return False
    table, idx = tag
    if not isinstance(table, torch.nn.Parameter) or ParamArena.lookup([table]) is None:
        return False
    items = sorted(cols.items())
    if len(items) != width or width != table.shape[1] or idx.numel() != n_rays:
        return False
    if any(c != k or s != items[0][1] + k for k, (c, s) in enumerate(items)):
        return False
    L.launch("hn_embed_backward", L.ptr(d_points), C.c_int(d_points.shape[1]), C.c_int(items[0][1]), L.ptr(idx),
             C.c_int(n_rays), C.c_int(n_samples), C.c_int(width), C.c_int(table.shape[0]), L.ptr(table.grad),
             L.stream_handle())
    return True


# --------------------------------------------------------------------------------------------
# sampling (no gradients flow through sample positions: the reference detaches them too)
# --------------------------------------------------------------------------------------------
def sample_along_rays(origins, directions, lower, upper, t_rand, scale: float = 1.0, want_points=True):
    """z = lower + (upper-lower)*(scale*t_rand), pts = o + z*d.  lower/upper: (n,) or (B,n)."""
    L.require_gpu(origins, directions, lower)
    L.load()
    b = origins.shape[0]
    n = lower.shape[-1]
    if origins.stride(-1) != 1 or directions.stride(-1) != 1 or origins.stride(0) != directions.stride(0):
        origins, directions = origins.contiguous(), directions.contiguous()
    z = torch.empty(b, n, dtype=torch.float32, device=origins.device)
    pts = torch.empty(b, n, 3, dtype=torch.float32, device=origins.device) if want_points else None
    lower = lower.contiguous()
    upper = upper.contiguous() if upper is not None else None
    t_rand = t_rand.contiguous() if t_rand is not None else None
    L.launch("hn_sample_along_rays", L.ptr(origins), L.ptr(directions), C.c_int(origins.stride(0)), L.ptr(lower),
                                     L.ptr(upper), C.c_int(1 if lower.dim() == 2 else 0), L.ptr(t_rand),
                                     C.c_float(scale), C.c_int(b), C.c_int(n), L.ptr(z), L.ptr(pts),
                                     L.stream_handle())
    return z, pts


def sample_pdf(weights, z, u, origins=None, directions=None, want_points=True, bins=None, merge=True, split=False):
    """Inverse-CDF sampling at draws u (B,Nf).

    Fused form (bins=None): `weights` = coarse weights (B,S); the kernel uses columns 1..S-2 and the
    midpoints of z (B,S) as bin edges.  General form: `bins` (B,n+1) and `weights` (B,n) given.
    Returns (z_all | None, pts | None, inds (B,Nf) int64, z_samples (B,Nf)); with `split` (needs the merge and the
    rays) two more: perm (B,S+Nf) int32 — sorted position -> entry of cat(z, z_samples) — and pts_new (B,Nf,3), the
    points of the new samples in draw order (a fine level that evaluates only those through the warp field)."""
    L.require_gpu(weights, u)
    L.load()
    nf = u.shape[1]
    dev = u.device
    weights = weights.detach()
    if weights.stride(-1) != 1:
        weights = weights.contiguous()
    if bins is None:
        b, nc = z.shape
        nb = nc - 2
        w_ptr = C.c_void_p(weights.data_ptr() + 4)
    else:
        b, nb = weights.shape
        bins = bins.detach().contiguous()
        nc = z.shape[1] if z is not None else 0
        w_ptr = L.ptr(weights)
    if z is not None:
        z = z.contiguous()
    do_merge = merge and z is not None
    want_points = want_points and do_merge and origins is not None
    if want_points and (origins.stride(-1) != 1 or directions.stride(-1) != 1 or origins.stride(0) != directions.stride(0)):
        origins, directions = origins.contiguous(), directions.contiguous()
    z_all = torch.empty(b, nc + nf, dtype=torch.float32, device=dev) if do_merge else None
    pts = torch.empty(b, nc + nf, 3, dtype=torch.float32, device=dev) if want_points else None
    inds = torch.empty(b, nf, dtype=torch.int64, device=dev)
    zs = torch.empty(b, nf, dtype=torch.float32, device=dev)
    if split:
        if not (do_merge and want_points):
            raise L.HnError("sample_pdf(split=True) needs the merge (z) and the rays (origins, directions)")
        perm = torch.empty(b, nc + nf, dtype=torch.int32, device=dev)
        pts_new = torch.empty(b, nf, 3, dtype=torch.float32, device=dev)
        L.launch("hn_sample_pdf_split", w_ptr, C.c_int(weights.stride(0)), L.ptr(bins), C.c_int(nb), L.ptr(z), C.c_int(nc),
                 L.ptr(u.contiguous()), L.ptr(origins), L.ptr(directions), C.c_int(origins.stride(0)), C.c_int(b),
                 C.c_int(nf), L.ptr(z_all), L.ptr(pts), L.ptr(inds), L.ptr(zs), L.ptr(perm), L.ptr(pts_new),
                 L.stream_handle())
        return z_all, pts, inds, zs, perm, pts_new
    L.launch("hn_sample_pdf", w_ptr, C.c_int(weights.stride(0)), L.ptr(bins), C.c_int(nb),
                              L.ptr(z if (do_merge or bins is None) else None), C.c_int(nc), L.ptr(u.contiguous()),
                              L.ptr(origins if want_points else None), L.ptr(directions if want_points else None),
                              C.c_int(origins.stride(0) if want_points else 0), C.c_int(b), C.c_int(nf),
                              L.ptr(z_all), L.ptr(pts), L.ptr(inds), L.ptr(zs), L.stream_handle())
    return z_all, pts, inds, zs


# --------------------------------------------------------------------------------------------
# random draws
# --------------------------------------------------------------------------------------------
_DRAW_STATE: Dict[str, torch.Tensor] = {}
FAST_DRAWS = os.environ.get("HN_FAST_DRAWS", "1") != "0"


def seed_draws(seed: Optional[int] = None, device=None):
    """(Re)seed the on-device generator of `random_draws` (default: torch.initial_seed(), i.e. what torch.manual_seed
    set).  The state {seed, offset, ticket} lives in device memory and is advanced by the kernel itself."""
    L.load()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if seed is None:
        # what torch.manual_seed set; data-parallel ranks seeded alike must still draw different numbers
        seed = torch.initial_seed()
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            seed += 0x9E3779B97F4A7C15 * torch.distributed.get_rank()
    seed = int(seed)
    st = torch.tensor([seed & 0x7fffffffffffffff, 0, 0], dtype=torch.int64, device=device)
    _DRAW_STATE[str(device)] = st
    return st


def _draw_state(device) -> torch.Tensor:
    st = _DRAW_STATE.get(str(device))
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            raise L.HnError("random_draws: first use inside a stream capture (run a warm-up step first)")
        st = seed_draws(None, device)
    return st


def _draw_table(specs: Sequence[Tuple[Tuple[int, ...], str]], device):
    if len(specs) > L.HN_MAX_DRAWS:
        raise L.HnError(f"at most {L.HN_MAX_DRAWS} buffers per random_draws call")
    outs = [torch.empty(shape, dtype=torch.float32, device=device) for shape, _ in specs]
    arr = (L.HnDraw * max(1, len(specs)))()
    for i, ((shape, kind), t) in enumerate(zip(specs, outs)):
        if kind not in ("uniform", "normal"):
            raise ValueError(kind)
        arr[i].ptr, arr[i].n, arr[i].kind = t.data_ptr(), t.numel(), 0 if kind == "uniform" else 1
    return outs, arr


def random_draws(specs: Sequence[Tuple[Tuple[int, ...], str]], device) -> List[torch.Tensor]:
    """One launch (hn_random_fill) for all the random tensors of a render step.  specs: [(shape, 'uniform' | 'normal')]
    -> fp32 tensors on `device`: U[0,1) (24 bits, as torch.rand) / N(0,1).  Philox4x32-10 with a device-resident
    counter: graph replays draw fresh numbers; `torch.manual_seed` before the first call (or `seed_draws`) makes the
    sequence reproducible.  Not the same stream as torch's own generator (nor is torch's GPU stream its CPU one)."""
    L.load()
    device = torch.device(device)
    st = _draw_state(device)
    outs, arr = _draw_table(specs, device)
    L.launch("hn_random_fill", arr, C.c_int(len(specs)), L.ptr(st), L.stream_handle())
    return outs


PROLOGUE = os.environ.get("HN_PROLOGUE", "1") != "0"      # A/B switch of the fused step head (NerfModel.forward)


def render_prologue(device, mode: int, pack_groups, specs: Sequence[Tuple[Tuple[int, ...], str]], sample=None, ids=None):
    """The head of a render step as ONE launch (hn_render_prologue): the weight streams of `pack_groups`
    (machine.collect_pack_jobs: at most one group), the random tensors `specs` (as random_draws), the coarse samples
    placed from draw `sample['draw']` — sample = dict(draw, origins, directions, lower, upper, scale) -> (z, pts) as
    sample_along_rays — and ids = (float column view (B,), out int64 (B,)).  Returns (draw tensors, z | None, pts | None)."""
    from . import machine as _M
    L.load()
    device = torch.device(device)
    outs, arr = _draw_table(specs, device)
    st = _draw_state(device) if specs else None
    if len(pack_groups) > 1:
        raise L.HnError("render_prologue: one pack group (one numeric mode, <= HN_MAX_PACK_JOBS programs)")
    p = L.HnPrologue()
    p.t_rand_draw = -1
    z = pts = None
    keep = []
    if sample is not None:
        o, d = sample["origins"], sample["directions"]
        if o.stride(-1) != 1 or d.stride(-1) != 1 or o.stride(0) != d.stride(0):
            o, d = o.contiguous(), d.contiguous()
        lower, upper = sample["lower"].contiguous(), sample["upper"].contiguous()
        b, n = o.shape[0], lower.shape[-1]
        z = torch.empty(b, n, dtype=torch.float32, device=device)
        pts = torch.empty(b, n, 3, dtype=torch.float32, device=device)
        p.t_rand_draw, p.n_rays, p.n, p.ray_ld = int(sample["draw"]), b, n, o.stride(0)
        p.per_ray_bounds, p.scale = int(lower.dim() == 2), float(sample.get("scale", 1.0))
        p.origins, p.dirs, p.lower, p.upper = o.data_ptr(), d.data_ptr(), lower.data_ptr(), upper.data_ptr()
        p.z_out, p.pts_out = z.data_ptr(), pts.data_ptr()
        keep += [o, d, lower, upper]
    if ids is not None:
        src, dst = ids
        if src.dtype != torch.float32 or src.dim() != 1 or dst.dtype != torch.int64 or not dst.is_contiguous() or \
                dst.numel() != src.numel():
            raise L.HnError("render_prologue: ids = (fp32 column view (B,), contiguous int64 (B,))")
        p.ids_src, p.ids_dst, p.ids_ld, p.n_ids = src.data_ptr(), dst.data_ptr(), max(1, src.stride(0)), src.numel()
    if pack_groups:
        m, jobs, grp = pack_groups[0]
        tag = "+".join(r.prog.name for r, *_ in grp)
    else:
        m, jobs, grp, tag = mode, None, [], ""
    L.launch("hn_render_prologue", C.c_int(m), jobs, C.c_int(len(grp)), arr if specs else None, C.c_int(len(specs)),
             L.ptr(st), C.byref(p), L.stream_handle(), tag=tag)
    if grp:
        _M.mark_packed(grp)
    return outs, z, pts


# --------------------------------------------------------------------------------------------
# compositing
# --------------------------------------------------------------------------------------------
class _CompositeFn(torch.autograd.Function):
    """One level's compositing.  With `perm` the level arrives in TWO parts (include/hn_kernels.h, HnCompositeArgs.perm):
    (rgb, raw, warped) = the coarse level's samples re-evaluated by the fine template, (rgb1, raw1, warped1) = the new
    samples; z, noise, keep and the returned weights are in sorted order, and one more output — the parts' warped
    rows in sorted order, the level's `warped_points` — is appended."""

    @staticmethod
    def forward(ctx, rgb, raw, noise, z, dirs, warped, variant, white_bg, sample_at_infinity, want_median,
                dust_threshold=None, keep=None, noise_scale=1.0, rgb1=None, raw1=None, warped1=None, perm=None,
                then_pdf=None):
        L.require_gpu(rgb, raw, z, dirs)
        L.load()
        b, s = z.shape
        dev = z.device
        s0 = s
        if perm is not None:
            L.require_gpu(rgb1, raw1, perm)
            if perm.dtype != torch.int32 or tuple(perm.shape) != (b, s):
                raise L.HnError("composite: perm must be a (B, S) int32 tensor")
            s0 = raw.numel() // b
            if raw.numel() != b * s0 or raw1.numel() != b * (s - s0) or rgb.numel() != 3 * b * s0 or \
                    rgb1.numel() != 3 * b * (s - s0):
                raise L.HnError("composite: the two parts do not add up to the level's samples")
            if (warped is None) != (warped1 is None) or (warped is not None and warped.shape[-1] != warped1.shape[-1]):
                raise L.HnError("composite: both parts need warped points of the same width (or neither)")
        rgb_c, raw_c, z_c = rgb.detach().contiguous(), raw.detach().reshape(b, s0).contiguous(), z.contiguous()
        noise_c = noise.detach().reshape(b, s).contiguous() if noise is not None else None
        dirs_c = dirs if dirs.stride(-1) == 1 else dirs.contiguous()
        warped_c = warped.detach().contiguous() if warped is not None else None
        a = L.HnCompositeArgs()
        a.variant, a.n_rays, a.n_samples = variant, b, s
        a.white_bg, a.sample_at_infinity = int(white_bg), int(sample_at_infinity)
        a.warped_ld = warped_c.shape[-1] if warped_c is not None else 0
        a.rgb, a.raw, a.noise, a.z = rgb_c.data_ptr(), raw_c.data_ptr(), (noise_c.data_ptr() if noise_c is not None else 0), z_c.data_ptr()
        a.dirs, a.ray_ld = dirs_c.data_ptr(), dirs_c.stride(0)
        a.warped = warped_c.data_ptr() if warped_c is not None else 0
        keep_c = keep.detach().reshape(b, s).contiguous().float() if keep is not None else None
        a.has_dust, a.dust_threshold = int(dust_threshold is not None), float(dust_threshold or 0.0)
        a.keep = keep_c.data_ptr() if keep_c is not None else 0
        a.noise_scale = float(noise_scale)
        rgb1_c = raw1_c = warped1_c = perm_c = o_warped = None
        if perm is not None:
            perm_c = perm.contiguous()
            rgb1_c, raw1_c = rgb1.detach().contiguous(), raw1.detach().reshape(b, s - s0).contiguous()
            warped1_c = warped1.detach().contiguous() if warped1 is not None else None
            a.perm, a.split, a.rgb1, a.raw1 = perm_c.data_ptr(), s0, rgb1_c.data_ptr(), raw1_c.data_ptr()
            if warped1_c is not None:
                a.warped1 = warped1_c.data_ptr()
                o_warped = torch.empty(b, s, a.warped_ld, dtype=torch.float32, device=dev)
                a.out_warped = o_warped.data_ptr()
        o_rgb = torch.empty(b, 3, dtype=torch.float32, device=dev)
        o_depth = torch.empty(b, dtype=torch.float32, device=dev)
        o_acc = torch.empty(b, dtype=torch.float32, device=dev)
        o_w = torch.empty(b, s, dtype=torch.float32, device=dev)
        o_md = torch.empty(b, dtype=torch.float32, device=dev) if want_median else None
        o_mp = torch.empty(b, dtype=torch.float32, device=dev) if (want_median and warped is not None) else None
        a.out_rgb, a.out_depth, a.out_acc, a.out_weights = o_rgb.data_ptr(), o_depth.data_ptr(), o_acc.data_ptr(), o_w.data_ptr()
        a.out_med_depth = o_md.data_ptr() if o_md is not None else 0
        a.out_med_points = o_mp.data_ptr() if o_mp is not None else 0
        pdf_out = None
        if then_pdf is not None:
            # the inverse-CDF sampling of the next level from this level's weights, in the same launch (sample_pdf's fused
            # form + split): then_pdf = dict(u (B, Nf), origins, directions, split)
            if perm is not None:
                raise L.HnError("composite(then_pdf=...) needs a level composited in one part")
            u_c = then_pdf["u"].contiguous()
            o_, d_ = then_pdf["origins"], then_pdf["directions"]
            if o_.stride(-1) != 1 or d_.stride(-1) != 1 or o_.stride(0) != d_.stride(0):
                o_, d_ = o_.contiguous(), d_.contiguous()
            nf = u_c.shape[1]
            z_all = torch.empty(b, s + nf, dtype=torch.float32, device=dev)
            pts_all = torch.empty(b, s + nf, 3, dtype=torch.float32, device=dev)
            inds = torch.empty(b, nf, dtype=torch.int64, device=dev)
            zs = torch.empty(b, nf, dtype=torch.float32, device=dev)
            perm_o = pts_new = None
            if then_pdf.get("split"):
                perm_o = torch.empty(b, s + nf, dtype=torch.int32, device=dev)
                pts_new = torch.empty(b, nf, 3, dtype=torch.float32, device=dev)
            L.launch("hn_composite_sample_pdf", C.byref(a), L.ptr(u_c), L.ptr(o_), L.ptr(d_), C.c_int(o_.stride(0)),
                     C.c_int(nf), L.ptr(z_all), L.ptr(pts_all), L.ptr(inds), L.ptr(zs), L.ptr(perm_o), L.ptr(pts_new),
                     L.stream_handle())
            pdf_out = [z_all, pts_all, inds, zs] + ([perm_o, pts_new] if perm_o is not None else [])
        else:
            L.launch("hn_composite_forward", C.byref(a), L.stream_handle())
        ctx.saved = (rgb_c, raw_c, noise_c, z_c, dirs_c)
        ctx.parts = (perm_c, rgb1_c, raw1_c, s0)
        ctx.filt = (dust_threshold, keep_c)
        ctx.noise_scale = float(noise_scale)
        ctx.cfg = (variant, int(white_bg), int(sample_at_infinity), b, s)
        ctx.raw_shape = raw.shape
        ctx.raw1_shape = raw1.shape if raw1 is not None else None
        ctx.n_med = 0
        outs = [o_rgb, o_depth, o_acc, o_w]
        nd = []
        if o_md is not None:
            outs.append(o_md); nd.append(o_md); ctx.n_med += 1
        if o_mp is not None:
            outs.append(o_mp); nd.append(o_mp); ctx.n_med += 1
        if o_warped is not None:
            # the sorted `warped_points` stays differentiable w.r.t. the parts' warped rows (a loss the caller puts on
            # it: rare, un-permuted with torch ops in backward)
            outs.append(o_warped)
        if pdf_out is not None:     # appended LAST (backward ignores them): (z_all, pts, inds, z_samples[, perm, pts_new])
            outs += pdf_out
            nd += pdf_out
        ctx.mark_non_differentiable(*nd)
        ctx.set_materialize_grads(False)      # unused outputs (depth, acc, weights) arrive as None, not as zero fills
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_acc, g_w, *rest):
        L.load()
        rgb_c, raw_c, noise_c, z_c, dirs_c = ctx.saved
        perm_c, rgb1_c, raw1_c, s0 = ctx.parts
        variant, white_bg, sai, b, s = ctx.cfg
        a = L.HnCompositeArgs()
        a.variant, a.n_rays, a.n_samples, a.white_bg, a.sample_at_infinity = variant, b, s, white_bg, sai
        a.rgb, a.raw, a.z = rgb_c.data_ptr(), raw_c.data_ptr(), z_c.data_ptr()
        a.noise = noise_c.data_ptr() if noise_c is not None else 0
        a.noise_scale = ctx.noise_scale
        a.dirs, a.ray_ld = dirs_c.data_ptr(), dirs_c.stride(0)
        dust, keep_c = ctx.filt
        a.has_dust, a.dust_threshold = int(dust is not None), float(dust or 0.0)
        a.keep = keep_c.data_ptr() if keep_c is not None else 0
        keep = []
        for name, g in (("g_rgb", g_rgb), ("g_depth", g_depth), ("g_acc", g_acc), ("g_weights", g_w)):
            if g is not None:
                g = g.contiguous(); keep.append(g)
                setattr(a, name, g.data_ptr())
        d_rgb = torch.empty_like(rgb_c)
        d_raw = torch.empty_like(raw_c)
        a.d_rgb, a.d_raw = d_rgb.data_ptr(), d_raw.data_ptr()
        d_rgb1 = d_raw1 = d_warped = d_warped1 = None
        if perm_c is not None:
            d_rgb1, d_raw1 = torch.empty_like(rgb1_c), torch.empty_like(raw1_c)
            a.perm, a.split, a.rgb1, a.raw1 = perm_c.data_ptr(), s0, rgb1_c.data_ptr(), raw1_c.data_ptr()
            a.d_rgb1, a.d_raw1 = d_rgb1.data_ptr(), d_raw1.data_ptr()
            g_ws = rest[ctx.n_med] if len(rest) > ctx.n_med else None
            if g_ws is not None and g_ws.numel() > 0:      # a loss on the sorted warped points: scatter it back to the parts
                h = g_ws.shape[-1]
                cat = torch.zeros(b, s, h, dtype=g_ws.dtype, device=g_ws.device)
                cat.scatter_(1, perm_c.long().unsqueeze(-1).expand(b, s, h), g_ws.reshape(b, s, h))
                d_warped, d_warped1 = cat[:, :s0].contiguous(), cat[:, s0:].contiguous()
        L.launch("hn_composite_backward", C.byref(a), L.stream_handle())
        return (d_rgb, d_raw.view(ctx.raw_shape), None, None, None, d_warped, None, None, None, None, None, None, None,
                d_rgb1, d_raw1.view(ctx.raw1_shape) if d_raw1 is not None else None, d_warped1, None, None)


COMPOSITE_PDF = os.environ.get("HN_COMPOSITE_PDF", "1") != "0"      # A/B switch: coarse compositing + inverse-CDF sampling as one launch


def composite(rgb, raw, noise, z, dirs, warped=None, variant=0, white_bg=False, sample_at_infinity=True,
              want_median=True, dust_threshold=None, keep=None, noise_scale: float = 1.0, rgb1=None, raw1=None,
              warped1=None, perm=None, then_pdf=None):
    """Returns (rgb (B,3), depth (B), acc (B), weights (B,S)[, med_depth (B)[, med_points (B)]]).
    `noise` (B,S): standard-normal draws, scaled by `noise_scale` inside the kernel (noise_std of noise_regularize).
    dust_threshold / keep (B,S 0/1): the reference's filter_sigma (models.py:35-63) applied to the activated density.
    With `perm` (B,S) int32 the level comes in two parts — (rgb, raw, warped) and (rgb1, raw1, warped1), see
    _CompositeFn — and the sorted warped points (B,S,H) are appended to the result.
    `then_pdf` = dict(u (B,Nf), origins, directions, split): the next level's inverse-CDF sampling from this level's
    weights in the SAME launch (hn_composite_sample_pdf); sample_pdf's results (z_all, pts, inds, z_samples[, perm,
    pts_new]) are appended to the result."""
    return _CompositeFn.apply(rgb, raw, noise, z, dirs, warped, variant, white_bg, sample_at_infinity, want_median,
                              dust_threshold, keep, noise_scale, rgb1, raw1, warped1, perm, then_pdf)


# --------------------------------------------------------------------------------------------
# loss head
# --------------------------------------------------------------------------------------------
class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coarse, fine, target):
        L.require_gpu(coarse, target)
        L.load()
        c = coarse.detach().contiguous()
        f = fine.detach().contiguous() if fine is not None else None
        t = target.detach().contiguous()
        if c.shape != t.shape or (f is not None and f.shape != t.shape):
            raise L.HnError("mse_loss: prediction and target shapes differ")
        loss = torch.empty((), dtype=torch.float32, device=c.device)
        ctx.unit = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            # the gradients for a root gradient of exactly 1 come out of the same launch (what `functional.backward(loss)`
            # passes: the loss of a training step IS the root); any other incoming gradient takes the backward kernel
            dc = torch.empty_like(c)
            df = torch.empty_like(f) if f is not None else None
            L.launch("hn_mse_loss_forward_grad", L.ptr(c), L.ptr(f), L.ptr(t), C.c_int64(c.numel()), L.ptr(loss), L.ptr(dc),
                     L.ptr(df), L.stream_handle())
            ctx.unit = (dc, df)
        else:
            L.launch("hn_mse_loss_forward", L.ptr(c), L.ptr(f), L.ptr(t), C.c_int64(c.numel()), L.ptr(loss),
                     L.stream_handle())
        ctx.saved = (c, f, t)
        return loss

    @staticmethod
    def backward(ctx, g):
        L.load()
        c, f, t = ctx.saved
        if ctx.unit is not None and g.data_ptr() in _UNIT_ROOTS:      # the cached device scalar 1.0 of functional.backward
            return ctx.unit[0], ctx.unit[1], None
        g = g.contiguous().float()
        dc = torch.empty_like(c)
        df = torch.empty_like(f) if f is not None else None
        L.launch("hn_mse_loss_backward", L.ptr(c), L.ptr(f), L.ptr(t), C.c_int64(c.numel()), L.ptr(g), L.ptr(dc),
                 L.ptr(df), L.stream_handle())
        return dc, df, None


_UNIT_ROOTS = set()     # data_ptr() of the cached root gradients that hold exactly 1.0 (never freed: _ROOT_GRADS keeps them)
_ROOT_GRADS: Dict[tuple, torch.Tensor] = {}


def backward(loss: torch.Tensor, weight: float = 1.0):
    """`(weight * loss).backward()` without the two launches autograd adds around it: the root gradient is a cached
    device scalar holding `weight` (loss.backward() alone fills a fresh ones_like every call, and `loss * w` is one
    more elementwise kernel) — what training.TrainStep and bench.py call once per step / chunk."""
    key = (str(loss.device), float(weight), loss.dtype)
    g = _ROOT_GRADS.get(key)
    if g is None:
        if torch.cuda.is_current_stream_capturing():
            raise L.HnError("functional.backward: first use of a new weight inside a stream capture (run a warm-up step)")
        g = _ROOT_GRADS[key] = torch.full((), float(weight), dtype=loss.dtype, device=loss.device)
        if float(weight) == 1.0 and loss.dtype == torch.float32:
            _UNIT_ROOTS.add(g.data_ptr())
    loss.backward(gradient=g)


def mse_loss(coarse: torch.Tensor, fine: Optional[torch.Tensor], target: torch.Tensor) -> torch.Tensor:
    """mean((coarse-target)^2) (+ mean((fine-target)^2)): losses.MSELoss of the reference as one kernel per direction."""
    return _MseFn.apply(coarse, fine, target)


# --------------------------------------------------------------------------------------------
# on-device ray generation
# --------------------------------------------------------------------------------------------
def generate_rays(h: int, w: int, focal: float, c2w: torch.Tensor, near: float, far: float, ndc: bool = False,
                  ndc_near: float = 1.0, image_id: Optional[int] = None) -> torch.Tensor:
    """(h*w, 8|9) ray rows [o, d, near, far(, image id)] of one image, generated on the GPU
    (reference: datasets/ray_utils.py get_ray_directions / get_rays / get_ndc_rays, datasets/llff.py:244-264)."""
    L.require_gpu(c2w)
    L.load()
    if tuple(c2w.shape) != (3, 4):
        raise L.HnError("c2w must be (3, 4)")
    c = c2w.detach().contiguous().float()
    cols = 9 if image_id is not None else 8
    rays = torch.empty(h * w, cols, dtype=torch.float32, device=c2w.device)
    L.launch("hn_generate_rays", C.c_int(h), C.c_int(w), C.c_float(focal), L.ptr(c), C.c_int(int(ndc)),
             C.c_float(ndc_near), C.c_float(near), C.c_float(far), C.c_float(float(image_id or 0)), C.c_int(cols),
             L.ptr(rays), L.stream_handle())
    return rays


# --------------------------------------------------------------------------------------------
# SE(3) exponential-map warp
# --------------------------------------------------------------------------------------------
class _Se3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, v, points):
        L.require_gpu(w, v, points)
        L.load()
        w_c, v_c, p_c = (t.detach().reshape(-1, 3).contiguous().float() for t in (w, v, points))
        n = p_c.shape[0]
        if w_c.shape[0] != n or v_c.shape[0] != n:
            raise L.HnError("se3_apply: w, v and points must have the same number of rows")
        out = torch.empty_like(p_c)
        L.launch("hn_se3_apply_forward", L.ptr(w_c), L.ptr(v_c), L.ptr(p_c), C.c_int(n), L.ptr(out),
                 L.stream_handle())
        ctx.saved = (w_c, v_c, p_c)
        ctx.shapes = (w.shape, v.shape, points.shape)
        return out.view(points.shape)

    @staticmethod
    def backward(ctx, g):
        L.load()
        w_c, v_c, p_c = ctx.saved
        g = g.reshape(-1, 3).contiguous()
        need = ctx.needs_input_grad
        d_w = torch.empty_like(w_c) if need[0] else None
        d_v = torch.empty_like(v_c) if need[1] else None
        d_p = torch.empty_like(p_c) if need[2] else None
        L.launch("hn_se3_apply_backward", L.ptr(w_c), L.ptr(v_c), L.ptr(p_c), L.ptr(g), C.c_int(p_c.shape[0]),
                 L.ptr(d_w) if d_w is not None else None, L.ptr(d_v) if d_v is not None else None,
                 L.ptr(d_p) if d_p is not None else None, L.stream_handle())
        sw, sv, sp = ctx.shapes
        return (d_w.view(sw) if d_w is not None else None, d_v.view(sv) if d_v is not None else None,
                d_p.view(sp) if d_p is not None else None)


def se3_apply(w: torch.Tensor, v: torch.Tensor, points: torch.Tensor) -> torch.Tensor:
    """y = exp([S] theta) . p with theta = |w|, S = (w, v) / theta, per point (reference: warping.py:226-238)."""
    return _Se3Fn.apply(w, v, points)


class _Se3WarpFn(torch.autograd.Function):
    """SE3Field's tail in one launch each way: wv (P, 6) = [w | v] straight from the field program (read with row
    stride 6, no slicing copies), points (P, 3) -> xyz (P, 3) and, optionally, `warped` (P, 3 + H) = [xyz | table row
    of the ray] (axis-aligned-plane levels).  The template reads the hyper coordinates from the gathered table itself,
    so in a training step no gradient arrives through `warped`; a loss the CALLER puts on results['warped_points']
    (differentiable w.r.t. both parts in the reference: models.py:578-581) does — its xyz columns join the gradient
    of `xyz`, its hyper columns are summed over the ray's samples into the table rows (rare path, torch ops).  The
    gradient of `xyz` arrives as columns of the template's source-gradient tensor (any row stride) and leaves as
    one (P, 6) tensor."""

    @staticmethod
    def forward(ctx, wv, points, table, idx, samples_per_ray):
        L.require_gpu(wv, points)
        L.load()
        wv_c = wv.detach()
        if wv_c.dim() != 2 or wv_c.shape[1] != 6 or wv_c.stride(1) != 1:
            wv_c = wv_c.reshape(-1, 6).contiguous()
        p_c = points.detach().reshape(-1, 3)
        if p_c.stride(1) != 1:
            p_c = p_c.contiguous()
        n = p_c.shape[0]
        if wv_c.shape[0] != n:
            raise L.HnError("se3_warp: wv and points must have the same number of rows")
        xyz = torch.empty(n, 3, dtype=torch.float32, device=p_c.device)
        warped, tab, gidx, h = None, None, None, 0
        if table is not None:
            tab = table.detach().contiguous()
            gidx = idx.reshape(-1).to(torch.int64).contiguous()
            h = tab.shape[1]
            if gidx.numel() * int(samples_per_ray) != n:
                raise L.HnError(f"se3_warp: {gidx.numel()} ray indices x {samples_per_ray} samples != {n} points")
            warped = torch.empty(n, 3 + h, dtype=torch.float32, device=p_c.device)
        L.launch("hn_se3_warp_forward", C.c_void_p(wv_c.data_ptr()), C.c_int(wv_c.stride(0)),
                 C.c_void_p(wv_c.data_ptr() + 12), C.c_int(wv_c.stride(0)), L.ptr(p_c), C.c_int(p_c.stride(0)), C.c_int(n),
                 L.ptr(xyz), L.ptr(warped), C.c_int(3 + h), L.ptr(tab), L.ptr(gidx), C.c_int(h),
                 C.c_int(tab.shape[0] if tab is not None else 0), C.c_int(int(samples_per_ray)), L.stream_handle())
        ctx.saved = (wv_c, p_c)
        ctx.pshape = points.shape
        ctx.rows = None if table is None else (gidx, tuple(table.shape), int(samples_per_ray))
        ctx.set_materialize_grads(False)
        if warped is None:
            return xyz
        return xyz, warped

    @staticmethod
    def backward(ctx, g, g_warped=None):
        L.load()
        wv_c, p_c = ctx.saved
        d_table = None
        if g_warped is not None:        # a loss on warped_points itself
            gw = g_warped.reshape(p_c.shape[0], -1)
            g = gw[:, :3] if g is None else g.reshape(-1, 3) + gw[:, :3]
            if ctx.needs_input_grad[2]:
                gidx, tshape, spr = ctx.rows
                d_table = torch.zeros(tshape, dtype=torch.float32, device=p_c.device)
                ok = (gidx >= 0) & (gidx < tshape[0])
                d_table.index_add_(0, gidx.clamp(0, tshape[0] - 1),
                                   gw[:, 3:].reshape(gidx.numel(), spr, -1).sum(1) * ok.unsqueeze(1))
        if g is None:
            return None, None, d_table, None, None
        if g.dim() != 2 or g.stride(1) != 1:
            g = g.reshape(-1, 3).contiguous()
        n = p_c.shape[0]
        need_p = ctx.needs_input_grad[1]
        d_wv = torch.empty(n, 6, dtype=torch.float32, device=p_c.device)
        d_p = torch.empty(n, 3, dtype=torch.float32, device=p_c.device) if need_p else None
        L.launch("hn_se3_warp_backward", C.c_void_p(wv_c.data_ptr()), C.c_int(wv_c.stride(0)),
                 C.c_void_p(wv_c.data_ptr() + 12), C.c_int(wv_c.stride(0)), L.ptr(p_c), C.c_int(p_c.stride(0)),
                 L.ptr(g), C.c_int(g.stride(0)), C.c_int(n), C.c_void_p(d_wv.data_ptr()), C.c_int(6),
                 C.c_void_p(d_wv.data_ptr() + 12), C.c_int(6), L.ptr(d_p), L.stream_handle())
        return d_wv, (d_p.view(ctx.pshape) if d_p is not None else None), d_table, None, None


def se3_warp(wv: torch.Tensor, points: torch.Tensor, table: Optional[torch.Tensor] = None,
             idx: Optional[torch.Tensor] = None, samples_per_ray: int = 1):
    """xyz = exp([S] theta) . p from the field's (P, 6) head output [w | v]; with `table` / `idx` also the (P, 3 + H)
    `warped_points` rows [xyz | table[idx[ray]]] (reference: warping.py:226-238, models.py:533-534, 578-581)."""
    return _Se3WarpFn.apply(wv, points, table, idx, samples_per_ray)


# --------------------------------------------------------------------------------------------
# stand-alone positional encoders
# --------------------------------------------------------------------------------------------
class _PosencFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, freqs, identity, jax_cos):
        L.require_gpu(x, freqs)
        L.load()
        c = x.shape[-1]
        xf = x.detach().reshape(-1, c).contiguous().float()
        n, nf = xf.shape[0], freqs.numel()
        width = c * (2 * nf + (1 if identity else 0))
        out = torch.empty(n, width, dtype=torch.float32, device=x.device)
        L.launch("hn_posenc", L.ptr(xf), C.c_int64(n), C.c_int(c), L.ptr(freqs), C.c_int(nf), C.c_int(int(identity)),
                              C.c_int(int(jax_cos)), L.ptr(out), None, None, L.stream_handle())
        ctx.save_for_backward(xf, freqs)
        ctx.cfg = (identity, jax_cos, x.shape)
        return out.view(*x.shape[:-1], width)

    @staticmethod
    def backward(ctx, g):
        L.load()
        xf, freqs = ctx.saved_tensors
        identity, jax_cos, shp = ctx.cfg
        n, c = xf.shape
        g = g.reshape(n, -1).contiguous()
        gx = torch.empty_like(xf)
        L.launch("hn_posenc", L.ptr(xf), C.c_int64(n), C.c_int(c), L.ptr(freqs), C.c_int(freqs.numel()),
                              C.c_int(int(identity)), C.c_int(int(jax_cos)), None, L.ptr(g), L.ptr(gx),
                              L.stream_handle())
        return gx.view(shp), None, None, None


def posenc(x, freqs: torch.Tensor, identity: bool, jax_cos: bool = False):
    return _PosencFn.apply(x, freqs, identity, jax_cos)


def sample_legacy(rays, t_vals, one_minus_t, use_disp, t_rand, scale):
    """nerf_pl coarse sampling with per-ray near/far (columns 6,7 of the ray rows)."""
    L.require_gpu(rays)
    L.load()
    if rays.stride(-1) != 1:
        rays = rays.contiguous()
    b, n = rays.shape[0], t_vals.numel()
    z = torch.empty(b, n, dtype=torch.float32, device=rays.device)
    pts = torch.empty(b, n, 3, dtype=torch.float32, device=rays.device)
    t_rand = t_rand.contiguous() if t_rand is not None else None
    L.launch("hn_sample_legacy", L.ptr(rays), C.c_int(rays.stride(0)), L.ptr(t_vals), L.ptr(one_minus_t),
                                 C.c_int(int(use_disp)), L.ptr(t_rand), C.c_float(scale), C.c_int(b), C.c_int(n),
                                 L.ptr(z), L.ptr(pts), L.stream_handle())
    return z, pts
