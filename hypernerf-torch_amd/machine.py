"""Host side of the MLP machine: turns a stack of nn.Linear layers into the op programs, weight
packing tables and weight-gradient job lists that the HIP kernels consume (include/hn_kernels.h).

Everything here is pure Python/numpy "compilation" — it runs on a machine without a GPU (and is
unit-tested there, including a lane-level emulation of the device data layouts).  The launch
methods require a ROCm device and the built extension; there is no CPU execution path.

Vocabulary
  layer    one nn.Linear (+activation) of a network.  Its input is [main | aux]:
           main = the running activation (`cur` on the device), aux = GENERATED features
           (positional encodings / per-ray conditions), described by an `AuxSpec`.
  head     a layer with <= 4 outputs whose result leaves the machine (rgb, alpha, warp offset...).
  program  the forward op list, the backward op list, the packing tables of both weight streams,
           the stash/mask slot table and the dW job list of one fused network.
"""
from __future__ import annotations

import os
import weakref

import ctypes as C
import itertools
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L

CHUNK = L.HN_CHUNK_UNITS
_FORCE_WIDE = os.environ.get("HN_FORCE_WIDE", "0") == "1"


def pow2ceil(n: int) -> int:
    p = 1
    while p < n:
        p *= 2
    return p


def bf16_like(mode: int) -> bool:
    """bf16 weight streams and matrix products (HN_MODE_BF16, and HN_MODE_BF16_S8 which differs in the stash only)."""
    return mode in (L.HN_MODE_BF16, L.HN_MODE_BF16_S8)


def mode_consts(mode: int):
    """(units per 32x32 weight block, bytes per stashed 32x32 tile)."""
    if mode == L.HN_MODE_BF16_S8:
        return 2, 1024
    return (2, 2048) if mode == L.HN_MODE_BF16 else (4, 4096)


# HN_MODE_BF16_S8: the backward machine carries 2^DZ_SCALE_LOG2 * dZ so that its e5m2 stash (normal range 6.1e-5 ..
# 57344) keeps gradients between 2^-30 and 0.87 (mean-reduced losses put them around 1e-7 .. 1e-3); the weight-gradient
# kernel divides the sums by it again.  A power of two: exact everywhere but in the 8-bit rounding itself.
DZ_SCALE_LOG2 = int(os.environ.get("HN_DZ_SCALE_LOG2", 16))


# --------------------------------------------------------------------------------------------
# network description
# --------------------------------------------------------------------------------------------
@dataclass
class Feature:
    src: int            # source array index (0..3)
    comp: int           # column of the source row
    kind: int           # HN_FEAT_*
    freq: float = 1.0
    need_grad: bool = False


def posenc_features(src: int, comps: Sequence[int], n_freqs: int, need_grad: bool = False) -> List[Feature]:
    """Feature list of posenc_orig (hypernerf/model_utils.py:234-246) over `comps` of source `src`."""
    out = [Feature(src, c, L.HN_FEAT_ID, 1.0, need_grad) for c in comps]
    for k in range(n_freqs):
        f = float(2.0 ** k)
        out += [Feature(src, c, L.HN_FEAT_SIN, f, need_grad) for c in comps]
        out += [Feature(src, c, L.HN_FEAT_COS, f, need_grad) for c in comps]
    return out


def posenc_jax_features(src: int, comps: Sequence[int], min_deg: int, max_deg: int, use_identity: bool,
                        need_grad: bool = False) -> List[Feature]:
    """Feature list of model_utils.posenc (hypernerf/model_utils.py:255-274), quirks included."""
    steps = max_deg - min_deg
    scales = (2.0 ** torch.linspace(float(min_deg), float(max_deg), steps=steps)).tolist()
    out = [Feature(src, c, L.HN_FEAT_ID, 1.0, need_grad) for c in comps] if use_identity else []
    for s in scales:
        out += [Feature(src, c, L.HN_FEAT_SIN, float(np.float32(s)), need_grad) for c in comps]
        out += [Feature(src, c, L.HN_FEAT_SINP, float(np.float32(s)), need_grad) for c in comps]
    return out


def copy_features(src: int, comps: Sequence[int], need_grad: bool = False) -> List[Feature]:
    return [Feature(src, c, L.HN_FEAT_ID, 1.0, need_grad) for c in comps]


@dataclass
class AuxSpec:
    feats: List[Feature]
    feat_off: int = -1      # set by the program
    slot: int = -1          # stash slot of the transposed features (set by the program)

    @property
    def n(self):
        return len(self.feats)

    @property
    def groups(self):
        return (len(self.feats) + 63) // 64


@dataclass
class OutSpec:
    dst: int
    col: int = 0
    act: str = "none"                       # 'none' | 'sigmoid'
    residual: Optional[Tuple[int, int]] = None   # (src index, col)
    wide: bool = False
    # (src, col0): the n results are ALSO published as the staged components (src, col0 + i) of the block, so that
    # later layers of the same program can encode them (fused level: warp field / hyper sheet -> template).  The
    # forward launch gets no pointer for that source; the backward launch reads it from the forward's output tensor.
    publish: Optional[Tuple[int, int]] = None
    publish_ci: int = -1                    # set by the program


@dataclass
class GradIn:
    src: int                # backward source index (4..7) holding d(out) per point; -1 = no external gradient
    col: int = 0
    sigmoid_y: Optional[Tuple[int, int]] = None   # (src index, col) of the forward output y
    # (src, col0): add the source gradient the rest of the program accumulated for the published components
    from_dsrc: Optional[Tuple[int, int]] = None
    dacc_q: int = 0                         # set by the program: accumulator rows 8q .. 8q+3


@dataclass
class Layer:
    name: str
    weight: torch.nn.Parameter                  # or a LIST of (rows_i, in) parameters stacked by rows (rows_i % 32 == 0):
    bias: Optional[torch.nn.Parameter]          # several Linears reading the same input run as one layer (SE3Field heads)
    main: Optional[Tuple[int, int]] = None      # (first column, n columns) of W applied to `cur`; a NEGATIVE first
                                                # column -k0 makes the layer read cur[k0:] only (block-sparse window)
    aux: Optional[AuxSpec] = None
    aux_c0: int = 0
    act: str = "none"                           # 'none' | 'relu'
    commit: bool = True
    out: Optional[OutSpec] = None
    grad_in: Optional[GradIn] = None
    # filled by Program
    prev: Optional["Layer"] = None
    n_out: int = 0
    nt: int = 0
    w_id: int = -1
    b_id: int = -1
    bias_off: int = 0
    mask_slot: int = -1
    out_slot: int = -1      # stash of the output activation
    dz_slot: int = -1       # stash of dZ
    parts: List[Tuple[int, int, int, int]] = field(default_factory=list)   # (w_id, b_id, first row, rows) per stacked matrix
    in_features: int = 0

    def part_of_row(self, row: int) -> Tuple[int, int, int, int]:
        for prt in self.parts:
            if prt[2] <= row < prt[2] + prt[3]:
                return prt
        return self.parts[-1]


@dataclass
class SlotInfo:
    kind: str       # 'stash' | 'mask'
    nt: int         # tiles (stash) or dwords per lane (mask) per block


class Program:
    """Compiles a list of `Layer`s (forward order) into device tables for both numeric modes."""

    def __init__(self, layers: List[Layer], n_src: int = 4, name: str = "mlp", no_direct: Sequence[int] = ()):
        """no_direct: source indices whose components must all be staged in LDS (a GATHERED per-ray source: the direct
        global read of surplus identity features, hn_direct_source, does not gather)."""
        self.name = name
        self.layers = layers
        self.n_src = n_src
        self.no_direct = frozenset(no_direct)
        self.params: List[torch.nn.Parameter] = []
        self.slots: List[SlotInfo] = []
        self.feat_table: List[Feature] = []
        self.dsrc_map: Dict[Tuple[int, int], int] = {}
        self.comp_map: Dict[Tuple[int, int], int] = {}     # (src, column) -> staged component index
        self.bias_len = 0
        self._link()
        self.fwd_ops = self._build_fwd_ops()
        self.bwd_ops = self._build_bwd_ops()
        self.tables: Dict[int, dict] = {}

    # ---- structure ----------------------------------------------------------------------------
    def _new_slot(self, kind: str, nt: int) -> int:
        self.slots.append(SlotInfo(kind, nt))
        if len(self.slots) > L.HN_MAX_SLOTS:
            raise ValueError(f"{self.name}: more than {L.HN_MAX_SLOTS} stash/mask slots")
        return len(self.slots) - 1

    def _link(self):
        cur: Optional[Layer] = None
        pid = {}
        # heads that publish their results as staged components get consecutive component indices; heads that take
        # their gradient from the source-gradient accumulators get rows 8q .. 8q+3 (registers 4q .. 4q+3 of the
        # lanes with h = 0, where the backward LOAD op needs them)
        self.reserved_slots = set()
        q = 0
        for ly in self.layers:
            o, gi = ly.out, ly.grad_in
            if o is not None and o.publish is not None:
                n = self._rows(ly)
                if o.wide or n > 4:
                    raise ValueError(f"{ly.name}: only narrow heads can publish components")
                o.publish_ci = len(self.comp_map)
                for i in range(n):
                    key = (o.publish[0], o.publish[1] + i)
                    if key in self.comp_map:
                        raise ValueError(f"{ly.name}: component {key} published twice")
                    self.comp_map[key] = len(self.comp_map)
            if gi is not None and gi.from_dsrc is not None:
                if q >= 4:
                    raise NotImplementedError("more than 4 heads fed from the source-gradient accumulators")
                gi.dacc_q = q
                for i in range(self._rows(ly)):
                    self.dsrc_map[(gi.from_dsrc[0], gi.from_dsrc[1] + i)] = 8 * q + i
                self.reserved_slots.update(range(8 * q, 8 * q + 4))
                q += 1
        # staging slots go to the components of TRIGONOMETRIC features first, whichever layer uses them (they can only
        # be evaluated from LDS); identity copies take what is left and are read directly from global memory beyond that
        # — otherwise a wide per-ray condition early in the program (GLO_dim >= ~24) starves a later encoder
        for ly in self.layers:
            if ly.aux is not None:
                for ft in ly.aux.feats:
                    if ft.kind not in (L.HN_FEAT_ZERO, L.HN_FEAT_ID) and (ft.src, ft.comp) not in self.comp_map:
                        if len(self.comp_map) >= L.HN_MAX_COMPS:
                            raise NotImplementedError(
                                f"{ly.name}: more than {L.HN_MAX_COMPS} distinct encoded source components")
                        self.comp_map[(ft.src, ft.comp)] = len(self.comp_map)
        for ly in self.layers:
            ws = list(ly.weight) if isinstance(ly.weight, (list, tuple)) else [ly.weight]
            bs = list(ly.bias) if isinstance(ly.bias, (list, tuple)) else [ly.bias] * len(ws)
            if len(ws) > 1 and any(w.shape[0] % 32 or w.shape[1] != ws[0].shape[1] for w in ws):
                raise ValueError(f"{ly.name}: stacked matrices need equal in_features and rows in multiples of 32")
            ly.in_features = ws[0].shape[1]
            ly.n_out = sum(w.shape[0] for w in ws)
            nt_valid = (ly.n_out + 31) // 32
            ly.nt = pow2ceil(nt_valid)
            if ly.nt > 8:
                raise NotImplementedError(f"{ly.name}: width {ly.n_out} > 256 is not supported by the MLP machine")
            if ly.out is not None and not ly.out.wide and ly.n_out > 4:
                raise ValueError(f"{ly.name}: narrow OUT needs <= 4 outputs")
            for prm in ws + bs:
                if prm is not None and id(prm) not in pid:
                    pid[id(prm)] = len(self.params)
                    self.params.append(prm)
            ly.parts, row0 = [], 0
            for w_, b_ in zip(ws, bs):
                ly.parts.append((pid[id(w_)], pid[id(b_)] if b_ is not None else -1, row0, w_.shape[0]))
                row0 += w_.shape[0]
            ly.w_id, ly.b_id = ly.parts[0][0], ly.parts[0][1]
            ly.bias_off = self.bias_len
            self.bias_len += 32 * ly.nt
            if ly.main is not None:
                if cur is None:
                    raise ValueError(f"{ly.name}: main input without a running activation")
                if ly.main[1] != cur.n_out:
                    raise ValueError(f"{ly.name}: main width {ly.main[1]} != previous output {cur.n_out}")
                ly.prev = cur
            if ly.aux is not None:
                if ly.aux.groups > L.HN_AUXG_MAX:
                    raise NotImplementedError(f"{ly.name}: more than {64 * L.HN_AUXG_MAX} generated input features")
                if ly.aux.feat_off < 0:
                    ly.aux.feat_off = len(self.feat_table)
                    pad = ly.aux.groups * 64 - ly.aux.n
                    self.feat_table += list(ly.aux.feats) + [Feature(0, 0, L.HN_FEAT_ZERO)] * pad
                    ly.aux.slot = self._new_slot("stash", 2 * ly.aux.groups)
                    for ft in ly.aux.feats:
                        if ft.kind != L.HN_FEAT_ZERO and (ft.src, ft.comp) not in self.comp_map:
                            # staged in LDS while there is room; trigonometric features always are; surplus
                            # identity features (wide raw inputs of stand-alone modules) are read directly
                            if len(self.comp_map) < L.HN_MAX_COMPS:
                                self.comp_map[(ft.src, ft.comp)] = len(self.comp_map)
                            elif ft.kind != L.HN_FEAT_ID or ft.comp > 255 or ft.src in self.no_direct:
                                raise NotImplementedError(
                                    f"{ly.name}: more than {L.HN_MAX_COMPS} distinct encoded source components"
                                    + (" (identity features of a gathered source cannot be read directly)"
                                       if ft.src in self.no_direct else ""))
                        if ft.need_grad and (ft.src, ft.comp) not in self.dsrc_map:
                            used = set(self.dsrc_map.values()) | self.reserved_slots
                            free = [k for k in range(L.HN_DSRC_COMPS) if k not in used]
                            if not free:
                                raise NotImplementedError(
                                    f"{self.name}: gradients w.r.t. more than {L.HN_DSRC_COMPS} source components (one "
                                    "accumulator tile per block); stand-alone modules with wider differentiable inputs "
                                    "are outside the render path")
                            self.dsrc_map[(ft.src, ft.comp)] = free[0]
            if ly.main is None and ly.aux is None:
                raise ValueError(f"{ly.name}: layer without input")
            main_in = 0
            if ly.main is not None:     # a negative first column: only cur[-main[0]:] is read
                main_in = ly.main[1] + min(0, ly.main[0])
                if ly.main[0] < 0 or ly.main[0] + ly.main[1] > ly.in_features:
                    main_in = min(ly.in_features - (ly.aux.n if ly.aux else 0), main_in)
            total_in = main_in + (ly.aux.n if ly.aux else 0)
            if total_in != ly.in_features:
                raise ValueError(f"{ly.name}: inputs {total_in} != in_features {ly.in_features}")
            if ly.commit:
                cur = ly
        if self.n_dsrc > L.HN_DSRC_COMPS:
            raise NotImplementedError(f"{self.name}: more than {L.HN_DSRC_COMPS} source-gradient components")
        if len(self.comp_map) > L.HN_MAX_COMPS:
            raise NotImplementedError(f"{self.name}: more than {L.HN_MAX_COMPS} staged source components")

        consumers = {id(l.prev) for l in self.layers if l.prev is not None}
        for ly in self.layers:
            ly.dz_slot = self._new_slot("stash", ly.nt)
            if id(ly) in consumers:
                ly.out_slot = self._new_slot("stash", ly.nt)
            if ly.act == "relu" and (id(ly) in consumers or (ly.out is not None and ly.out.wide)):
                ly.mask_slot = self._new_slot("mask", (ly.nt + 1) // 2)
        last = self.layers[-1]
        if last.out is None:
            raise ValueError("the last layer of a program must produce an output")

    @staticmethod
    def _rows(ly: Layer) -> int:
        w = ly.weight
        return sum(x.shape[0] for x in w) if isinstance(w, (list, tuple)) else w.shape[0]

    @staticmethod
    def _main_end(ly: Layer) -> int:
        """One past the last valid source column of the main block (a window layer's matrix is narrower than cur)."""
        end = ly.main[0] + ly.main[1]
        if end > ly.in_features:            # window layer
            if ly.aux is not None:
                raise NotImplementedError(f"{ly.name}: a window on the running activation cannot be combined with "
                                          "generated input features")
            end = ly.in_features
        return end

    @property
    def n_trig_comps(self) -> int:
        """Staged components 0 .. n-1 cover every component a trigonometric feature reads (they are assigned first,
        `_link`): the bf16 forward stages x / 2pi as hi + lo for exactly those."""
        idx = [self.comp_map[(f.src, f.comp)] for f in self.feat_table
               if f.kind not in (L.HN_FEAT_ZERO, L.HN_FEAT_ID) and (f.src, f.comp) in self.comp_map]
        return max(idx) + 1 if idx else 0

    @property
    def n_dsrc(self):
        """Rows of the source-gradient accumulator in use (slot numbers may have gaps: reserved head rows)."""
        return max(self.dsrc_map.values()) + 1 if self.dsrc_map else 0

    def chains(self) -> List[List[Layer]]:
        """The layer list cut into chains: a layer without a main input (fed by generated features only) starts a new
        network whose activations do not depend on `cur` — the warp field, the hyper sheet and the template of a
        fused level program are three chains."""
        out: List[List[Layer]] = []
        for ly in self.layers:
            if ly.main is None or not out:
                out.append([])
            out[-1].append(ly)
        return out

    def embed_fold(self, src: int):
        """(register mask, slot -> table column) of the source-gradient rows that belong to per-ray source `src`:
        what the backward machine needs to reduce them over a block and scatter them into the embedding gradient."""
        col = [-1] * L.HN_DSRC_COMPS
        mask = 0
        for (s_, c), slot in self.dsrc_map.items():
            if s_ == src:
                col[slot] = c
                mask |= 1 << ((slot & 3) + 4 * (slot >> 3))
        return mask, col

    # ---- forward ops --------------------------------------------------------------------------
    def _build_fwd_ops(self) -> np.ndarray:
        ops = []
        stashed_aux = set()
        for ly in self.layers:
            k32 = ly.prev.nt if ly.main is not None else 0
            ng = ly.aux.groups if ly.aux is not None else 0
            act = L.HN_ACT_RELU if ly.act == "relu" else L.HN_ACT_NONE
            flags = 0 if ly.commit else L.HN_LAYER_NO_COMMIT
            if ly.aux is not None and any(f.kind != L.HN_FEAT_ZERO and (f.src, f.comp) not in self.comp_map
                                          for f in ly.aux.feats):
                flags |= L.HN_LAYER_DIRECT
            aux_slot = -1
            if ly.aux is not None and id(ly.aux) not in stashed_aux:
                aux_slot = ly.aux.slot
                stashed_aux.add(id(ly.aux))
            ops.append([L.HN_OP_LAYER, k32 | ng << 8 | ly.nt << 16 | act << 24 | flags << 28, ly.bias_off,
                        ly.aux.feat_off if ly.aux is not None else 0, ly.mask_slot, ly.out_slot, aux_slot, 0])
            if ly.out is not None:
                o = ly.out
                if o.wide:
                    if not ly.commit:
                        raise ValueError("wide outputs are read from the committed activation")
                    ops.append([L.HN_OP_OUT_WIDE, o.dst, o.col, ly.n_out, ly.nt, 0, 0, 0])
                else:
                    ops.append([L.HN_OP_OUT, o.dst, o.col, ly.n_out, 1 if o.act == "sigmoid" else 0,
                                o.residual[0] if o.residual else -1, o.residual[1] if o.residual else 0,
                                o.publish_ci + 1 if o.publish is not None else 0])
        return np.asarray(ops, dtype=np.int32)

    # ---- backward ops -------------------------------------------------------------------------
    def _aux_grad_groups(self, ly: Layer) -> List[int]:
        if ly.aux is None:
            return []
        return [g for g in range(ly.aux.groups) if any(f.need_grad for f in ly.aux.feats[64 * g:64 * g + 64])]

    @staticmethod
    def _aux_tile_word(ly: Layer, g: int) -> int:
        """w2 of an HN_BOP_AUX op: bit tt = 32-feature tile tt of the group holds a feature with a gradient (only those
        tiles are in the backward weight stream and computed: the encoder of a plain input in front of a GLO row and the
        padding behind it cost a W^T dZ product and 16 sines per lane each — 7 of the 22 tiles of a config-2 level),
        bit 8 + tt = one of them is trigonometric (identity-only tiles skip the chain-rule factor, which is 1)."""
        word = 0
        for tt in range(2):
            fts = [f for f in ly.aux.feats[64 * g + 32 * tt:64 * g + 32 * tt + 32] if f.need_grad]
            if fts or not AUX_TILE_SKIP:
                word |= 1 << tt
            if any(f.kind not in (L.HN_FEAT_ID, L.HN_FEAT_ZERO) for f in fts) or AUX_TILE_SKIP < 2:
                word |= 256 << tt
        return word

    def _build_bwd_ops(self) -> np.ndarray:
        """Reverse walk, chain by chain (last chain first).  Records, next to the ops, the weight blocks each op
        streams (self.bwd_plan)."""
        ops, plan = [], []

        def emit_load(ly: Layer, to2: bool):
            gi = ly.grad_in
            if gi is None:
                raise ValueError(f"{ly.name}: output layer without a gradient source")
            if ly.out.wide:
                ops.append([L.HN_BOP_LOAD_WIDE, gi.src, gi.col, ly.n_out, ly.nt, ly.mask_slot, 0, ly.dz_slot])
            else:
                sy = gi.sigmoid_y
                w3 = ly.n_out | (256 if to2 else 0)
                if gi.from_dsrc is not None:
                    w3 |= 512 | gi.dacc_q << 10
                ops.append([L.HN_BOP_LOAD, gi.src, gi.col, w3, 1 if sy else 0,
                            sy[0] if sy else 0, sy[1] if sy else 0, ly.dz_slot])
            plan.append(("load",))

        def emit_aux(ly: Layer, from2: bool):
            for g in self._aux_grad_groups(ly):
                k32 = 0 if from2 else ly.nt
                ops.append([L.HN_BOP_AUX, k32 | (1 if from2 else 0) << 8 | 1 << 16, self._aux_tile_word(ly, g),
                            ly.aux.feat_off + 64 * g, 0, 0, 0, 0])
                plan.append(("aux", ly, g, from2))

        for items in reversed(self.chains()):
            final = items[-1]
            if final.out is None:
                raise ValueError(f"{final.name}: the last layer of a chain must produce an output")
            emit_load(final, False)
            emit_aux(final, False)
            consumer = final
            while consumer.prev is not None:
                P = consumer.prev
                heads = [h for h in items if (not h.commit) and h.prev is P and h is not consumer and h is not final]
                if len(heads) > 1:
                    raise NotImplementedError("more than one side head on one activation")
                head = heads[0] if heads else None
                if head is not None:
                    emit_load(head, True)
                    emit_aux(head, True)
                mask = P.mask_slot if P.act == "relu" else -1
                ops.append([L.HN_BOP_LAYER, consumer.nt | (1 if head else 0) << 8 | P.nt << 16, 0, 0, mask, P.dz_slot,
                            0, 0])
                plan.append(("layer", consumer, head, P))
                consumer = P
                emit_aux(P, False)
            # every head must have been visited
            seen = {id(p[1]) for p in plan if p[0] == "layer"} | {id(p[2]) for p in plan if p[0] == "layer" and p[2]}
            for ly in items:
                if not ly.commit and ly is not final and id(ly) not in seen:
                    raise NotImplementedError(f"{ly.name}: head without a later consumer of its input")
        self.bwd_plan = plan
        return np.asarray(ops, dtype=np.int32)

    # ---- weight streams -----------------------------------------------------------------------
    @staticmethod
    def _take(ctr: int, n: int) -> Tuple[int, int]:
        if (ctr % CHUNK) + n > CHUNK:
            ctr = (ctr + CHUNK - 1) // CHUNK * CHUNK
        return ctr, ctr + n

    def _stream_fwd(self, mode: int):
        u32, _ = mode_consts(mode)
        units: Dict[int, tuple] = {}
        ctr = 0

        def put(pos, w_id, ld, r0, c0, r_end, c_end, transposed):
            for u in range(u32):
                k0 = 16 * u if bf16_like(mode) else 4 * u
                units[pos + u] = (w_id, ld, r0, c0, r_end, c_end, k0, transposed)

        for ly in self.layers:
            ld = ly.in_features
            for t in range(ly.nt):
                w_id, _b, row0, rows = ly.part_of_row(32 * t)
                if ly.main is not None:
                    k32 = ly.prev.nt
                    pos, ctr = self._take(ctr, k32 * u32)
                    for k in range(k32):
                        put(pos + k * u32, w_id, ld, 32 * t - row0, ly.main[0] + 32 * k, rows, self._main_end(ly), 0)
                if ly.aux is not None:
                    for g in range(ly.aux.groups):
                        pos, ctr = self._take(ctr, 2 * u32)
                        for kk in range(2):
                            put(pos + kk * u32, w_id, ld, 32 * t - row0, ly.aux_c0 + 64 * g + 32 * kk, rows,
                                ly.aux_c0 + ly.aux.n, 0)
        return units, ctr

    def _stream_bwd(self, mode: int):
        u32, _ = mode_consts(mode)
        units: Dict[int, tuple] = {}
        ctr = 0

        def put(pos, ly: Layer, r0, c0, c_end):
            w_id, _b, row0, rows = ly.part_of_row(r0)
            for u in range(u32):
                k0 = 16 * u if bf16_like(mode) else 4 * u
                units[pos + u] = (w_id, ly.in_features, r0 - row0, c0, rows, c_end, k0, 1)

        for step in self.bwd_plan:
            if step[0] == "layer":
                _, consumer, head, P = step
                for t in range(P.nt):
                    pos, ctr = self._take(ctr, consumer.nt * u32)
                    for k in range(consumer.nt):
                        put(pos + k * u32, consumer, 32 * k, consumer.main[0] + 32 * t, self._main_end(consumer))
                    if head is not None:
                        pos, ctr = self._take(ctr, u32)
                        put(pos, head, 0, head.main[0] + 32 * t, self._main_end(head))
            elif step[0] == "aux":
                _, ly, g, from2 = step
                for tt in range(2):
                    if not (self._aux_tile_word(ly, g) >> tt) & 1:
                        continue
                    c0 = ly.aux_c0 + 64 * g + 32 * tt
                    if not from2:
                        pos, ctr = self._take(ctr, ly.nt * u32)
                        for k in range(ly.nt):
                            put(pos + k * u32, ly, 32 * k, c0, ly.aux_c0 + ly.aux.n)
                    else:
                        pos, ctr = self._take(ctr, u32)
                        put(pos, ly, 0, c0, ly.aux_c0 + ly.aux.n)
                    # the 0/1 selection block that sums this tile's feature gradients into their source components
                    pos, ctr = self._take(ctr, u32)
                    f0 = ly.aux.feat_off + 64 * g + 32 * tt
                    for u in range(u32):
                        k0 = 16 * u if bf16_like(mode) else 4 * u
                        units[pos + u] = (self.sel_w_id, len(self.feat_table), 0, f0, self.n_dsrc,
                                          len(self.feat_table), k0, 0)
        return units, ctr

    @property
    def sel_w_id(self) -> int:
        """Pseudo weight id of the selection matrix (one past the parameters) in the packer's pointer list."""
        return len(self.params)

    def selection_matrix(self) -> np.ndarray:
        """S[d, f] = 1 where generated feature f differentiates into source-gradient column d.  The backward machine
        reduces (W_aux^T dZ * dfeature/dx) over features with one more matrix product against S."""
        S = np.zeros((max(1, self.n_dsrc), max(1, len(self.feat_table))), dtype=np.float32)
        for i, f in enumerate(self.feat_table):
            if f.need_grad and (f.src, f.comp) in self.dsrc_map:
                S[self.dsrc_map[(f.src, f.comp)], i] = 1.0
        return S

    def _units_array(self, units: Dict[int, tuple], ctr: int) -> Tuple[np.ndarray, int]:
        n_chunks = max(1, (ctr + CHUNK - 1) // CHUNK)
        arr = np.zeros(n_chunks * CHUNK, dtype=L.PACK_UNIT_DT)
        arr["w_id"] = -1
        for pos, v in units.items():
            arr[pos] = v
        return arr, n_chunks

    def host_tables(self, mode: int) -> dict:
        """All position-independent tables of one numeric mode (numpy, cached)."""
        if mode in self.tables:
            return self.tables[mode]
        fu, fc = self._stream_fwd(mode)
        bu, bc = self._stream_bwd(mode)
        fwd_units, fwd_chunks = self._units_array(fu, fc)
        bwd_units, bwd_chunks = self._units_array(bu, bc)
        rows = []
        for ly in self.layers:
            for j, (_w, b_id, row0, nrows) in enumerate(ly.parts):
                last = j == len(ly.parts) - 1
                rows.append((b_id, nrows if b_id >= 0 else 0, ly.bias_off + row0, (32 * ly.nt - row0) if last else nrows))
        bias = np.zeros(len(rows), dtype=L.PACK_BIAS_DT)
        for i, r_ in enumerate(rows):
            bias[i] = r_
        feat = np.zeros(max(1, len(self.feat_table)), dtype=L.FEAT_DT)
        for i, f in enumerate(self.feat_table):
            slot = self.dsrc_map.get((f.src, f.comp), -1) + 1 if f.need_grad else 0
            if f.kind != L.HN_FEAT_ZERO and (f.src, f.comp) not in self.comp_map:
                word = f.src << 8 | L.HN_FEAT_ID_DIRECT << 12 | slot << 16 | f.comp << 24       # comp >= 128 sets bit 31
                feat[i] = (int(np.uint32(word).view(np.int32)), np.float32(f.freq))
                continue
            ci = self.comp_map.get((f.src, f.comp), 0)
            feat[i] = (ci | f.kind << 12 | slot << 16, np.float32(f.freq))
        comps = np.zeros(max(1, len(self.comp_map)), dtype=np.int32)
        for (src, col), ci in self.comp_map.items():
            comps[ci] = src << 16 | col
        t = dict(fwd_units=fwd_units, fwd_chunks=fwd_chunks, bwd_units=bwd_units, bwd_chunks=bwd_chunks,
                 bias=bias, feat=feat, comps=comps)
        self.tables[mode] = t
        return t

    # ---- per-size layout ------------------------------------------------------------------------
    def layout(self, mode: int, n_points: int):
        """Byte offsets of every slot for `n_points` points: (slot structs, stash bytes, mask bytes)."""
        _, tile_bytes = mode_consts(mode)
        nblk = (n_points + 31) // 32
        offs, stash_b, mask_b = [], 0, 0
        for s in self.slots:
            if s.kind == "stash":
                offs.append((stash_b, s.nt))
                stash_b += nblk * s.nt * tile_bytes
            else:
                offs.append((mask_b, s.nt))
                mask_b += nblk * s.nt * 64 * 4
        return offs, stash_b, mask_b

    def resolved_ops(self, mode: int, n_points: int) -> Tuple[np.ndarray, np.ndarray]:
        """(forward ops, backward ops) with every slot field replaced by the slot's offset for `n_points` points:
        stash regions in KiB, mask regions in units of 256 B (include/hn_kernels.h).  The kernels then need no slot
        table (a kernel-argument lookup costs two dependent scalar loads per slot and layer)."""
        offs, _, _ = self.layout(mode, n_points)

        def res(slot: int) -> int:
            if slot < 0:
                return -1
            off = offs[slot][0]
            unit = 1024 if self.slots[slot].kind == "stash" else 256
            assert off % unit == 0 and off // unit < 2 ** 31
            return off // unit
        fwd = self.fwd_ops.copy()
        for w in fwd:
            if w[0] == L.HN_OP_LAYER:
                w[4], w[5], w[6] = res(int(w[4])), res(int(w[5])), res(int(w[6]))
        bwd = self.bwd_ops.copy()
        for w in bwd:
            if w[0] == L.HN_BOP_LOAD:
                w[7] = res(int(w[7]))
            elif w[0] == L.HN_BOP_LOAD_WIDE:
                w[5], w[7] = res(int(w[5])), res(int(w[7]))
            elif w[0] == L.HN_BOP_LAYER:
                w[4], w[5] = res(int(w[4])), res(int(w[5]))
        return fwd, bwd

    def grad_offsets(self) -> Tuple[List[int], int]:
        offs, tot = [], 0
        for p in self.params:
            offs.append(tot)
            tot += (p.numel() + 3) // 4 * 4
        return offs, tot

    @staticmethod
    def _wave_grid(n_nt: int, n_kt: int, one_row: bool = False) -> Tuple[int, int]:
        """(gn, gk), gn*gk <= 8 waves, every wave's rectangle <= 4x2 tiles.  `one_row`: the grids of rounds 1-3."""
        if WGRAD_GRID == 0 or one_row:   # one row of waves for <= 4 dZ tiles (a 4x4 job runs on 4 of its 8 waves)
            if n_nt <= 4:
                return 1, min(8, max(1, n_kt)) if n_kt <= 8 else 8
            return 2, 4
        # round 4: as many ACTIVE waves as the rectangle admits (a wave without a tile only issues DMA; with one
        # active wave per SIMD nothing hides its LDS read latencies), then the smallest rectangle per wave, then the
        # fewest operand tiles read per block over all waves
        best = None
        for gn in range(1, 9):
            for gk in range(1, 8 // gn + 1):
                if gn > n_nt or gk > n_kt:
                    continue
                tn, tk = -(-n_nt // gn), -(-n_kt // gk)
                if tn > 4 or tk > 2:
                    continue
                key = (-(gn * gk), tn * tk, gn * gk * (tn + tk))
                if best is None or key < best[0]:
                    best = (key, gn, gk)
        if best is None:
            raise NotImplementedError(f"no wave grid for a {n_nt} x {n_kt} tile rectangle")
        return best[1], best[2]

    def _wgrad_rects(self, mode: int, n_points: int) -> list:
        """The dW rectangles of every layer: (layer, dZ offset, X offset, X tiles per block, first column, columns,
        first dZ tile, first k-tile, dZ tiles, k-tiles, with bias, stacked part, second X slot | None)."""
        offs, _, _ = self.layout(mode, n_points)
        tmax = 8 if bf16_like(mode) else 4
        rects = []
        for ly in self.layers:
            segs = []
            if ly.main is not None:
                segs.append((offs[ly.prev.out_slot][0], ly.prev.nt, ly.main[0], ly.main[1]))
            if ly.aux is not None:
                segs.append((offs[ly.aux.slot][0], 2 * ly.aux.groups, ly.aux_c0, ly.aux.n))
            z_off = offs[ly.dz_slot][0]
            # a skip layer's two input segments (running activation | re-appended encoder input) as ONE rectangle
            # with two X slots when it fits a job (<= 8 k-tiles): its dZ tiles are then read once, not once per
            # segment (round 4: the launch is HBM-bound, every byte counts — 3.6 % of its reads at config 2)
            x2 = None
            if (WGRAD_FUSE_SEGS and bf16_like(mode) and len(segs) == 2 and segs[0][2] == 0 and segs[0][3] % 32 == 0
                    and segs[1][2] == segs[0][3] and segs[0][3] // 32 + (segs[1][3] + 31) // 32 <= tmax):
                x2 = (segs[1][0], segs[1][1], segs[0][3] // 32)          # (offset, tiles per block, k-tiles of slot 1)
                segs = [(segs[0][0], segs[0][1], 0, segs[0][3] + segs[1][3])]
            first = True
            for (x_off, x_nt, c0, ncols) in segs:
                k_tiles = (ncols + 31) // 32
                for prt in ly.parts:                    # one rectangle set per stacked matrix
                    p_t0, p_tiles = prt[2] // 32, (prt[3] + 31) // 32
                    for nt0 in range(p_t0, p_t0 + p_tiles, tmax):
                        n_nt = min(tmax, p_t0 + p_tiles - nt0)
                        for kt0 in range(0, k_tiles, tmax):
                            n_kt = min(tmax, k_tiles - kt0)
                            with_bias = first and kt0 == 0 and prt[1] >= 0
                            rects.append((ly, z_off, x_off, x_nt, c0, ncols, nt0, kt0, n_nt, n_kt, with_bias, prt, x2))
                first = False
        return rects

    def wgrad_stream_bytes(self, mode: int, n_points: int, grad_offsets: Optional[Sequence[int]] = None,
                           min_w_off: Optional[int] = None) -> float:
        """Stash bytes the weight-gradient jobs of this program stream for `n_points` points (every rectangle reads its
        dZ and X tiles of every block once) — what `wgrad_jobs` sizes a batched launch's jobs by.  `min_w_off`: only the
        rectangles whose matrix lies at or behind that offset of the gradient buffer (the first bucket of a split
        launch, WGRAD_SPLIT_OFFSET)."""
        goffs = list(grad_offsets) if grad_offsets is not None else self.grad_offsets()[0]
        nblk = (n_points + 31) // 32
        tiles = sum(r[8] + r[9] for r in self._wgrad_rects(mode, n_points)
                    if min_w_off is None or goffs[r[11][0]] >= min_w_off)
        return float(tiles) * nblk * mode_consts(mode)[1]

    def wgrad_jobs(self, mode: int, n_points: int, target_jobs: int = 512,
                   grad_offsets: Optional[Sequence[int]] = None, job_bytes: Optional[int] = None,
                   launch_bytes: Optional[float] = None) -> np.ndarray:
        """One job per (layer input segment, tile rectangle, block chunk).  The chunks are sized so that the
        launch is about `target_jobs` workgroups of equal stash bytes — or, with `job_bytes`, so that every job
        streams about that many bytes (the batched launch mixes the jobs of several programs; `launch_bytes` = the
        stash bytes of that WHOLE launch, all its programs: jobs of big rectangles are sized to one CU's share of it)."""
        goffs = list(grad_offsets) if grad_offsets is not None else self.grad_offsets()[0]
        nblk = (n_points + 31) // 32
        # tiles per LDS stage of hn_wgrad_kernel: 64 KiB (2-stage ring; 32 KiB x 4 in rounds 2-5a), 8-bit stash 48 KiB (3-stage ring)
        stage_tiles = 48 if mode == L.HN_MODE_BF16_S8 else WGRAD_STAGE_KB * 1024 // mode_consts(mode)[1]
        rects = self._wgrad_rects(mode, n_points)
        total_tiles = sum(r[8] + r[9] for r in rects)
        jobs = []
        for (ly, z_off, x_off, x_nt, c0, ncols, nt0, kt0, n_nt, n_kt, with_bias, prt, x2) in rects:
            # fp32 (parity) mode at full batch sizes is bound by the 16x slower fp32 matrix pipe, not by latencies: there
            # the extra operand reads of more, smaller wave rectangles cost 6 % on the launch (config 2, same box:
            # 4.74 -> 5.04 ms), while small batches gain like the bf16 mode (config 1: 0.966 -> 0.750 ms)
            gn, gk = self._wave_grid(n_nt, n_kt, one_row=(not bf16_like(mode)) and n_points >= 65536)
            bps = max(1, stage_tiles // (n_nt + n_kt))
            nstage = -(-nblk // bps)
            if job_bytes is not None:
                tile_bytes = mode_consts(mode)[1]
                # a job ends with the flush of its dW rectangle by float atomics, which a CU retires at ~5 GB/s (one
                # 256-B wave-instruction per ~50 ns: MI355X_MICROARCH.md) while it streams at ~21 GB/s: an 8 x 8
                # rectangle (256 KiB) costs 51 us per job — 22 % on top of a 5-MiB job's stream —, a 4 x 4 one 13 us.
                # Jobs of big rectangles therefore stream as much as one CU's share of the whole launch allows (ONE
                # flush per CU and rectangle; measured at config 2, same box: 12.5-15 MiB jobs 0.661-0.668 ms against
                # 0.717-0.725 ms with 5 MiB, 20 MiB 0.835 ms: one job longer than the launch), jobs of small ones
                # fewer (they fill the tail of the launch): WGRAD_JOB_SCALE / `launch_bytes`
                big, small = n_nt * n_kt >= 48, n_nt * n_kt < 12
                share_cu = (launch_bytes if launch_bytes else total_tiles * nblk * tile_bytes) / N_CUS
                # round 5: the base size grows with the launch — a CU never gets more than ~WGRAD_JOBS_PER_CU (16) mid-sized
                # jobs (config 3 streams 266 MB per CU: 5-MiB jobs were 50 flushes per CU, and 0.5 ms of slab reduction)
                base = max(job_bytes, share_cu / WGRAD_JOBS_PER_CU) if WGRAD_JOBS_PER_CU > 0 else job_bytes
                if WGRAD_JOB_SCALE is not None:
                    jb = job_bytes * WGRAD_JOB_SCALE[0 if big else (2 if small else 1)]
                elif big:
                    jb = min(max(0.85 * share_cu, base), WGRAD_BIG_CAP * base)
                else:
                    jb = base * (0.5 if small else 1.0)
                share = max(1, min(nstage, round((n_nt + n_kt) * nblk * tile_bytes / jb)))
            else:
                share = max(1, min(nstage, round(target_jobs * (n_nt + n_kt) / total_tiles)))
            # split the stages as evenly as possible over `share` jobs
            bounds = [min(nblk, (i * nstage // share) * bps) for i in range(share + 1)]
            bounds[-1] = nblk
            for b0, b1 in zip(bounds[:-1], bounds[1:]):
                if b1 > b0:
                    c_end = min(c0 + ncols, ly.in_features)      # window layers: the matrix is narrower than cur
                    jobs.append((z_off, x_off, ly.nt, x_nt, nt0, kt0, n_nt, n_kt, b0, b1,
                                 goffs[prt[0]], ly.in_features, 32 * nt0 - prt[2], c0 + 32 * kt0, prt[3], c_end,
                                 goffs[prt[1]] if with_bias else -1, gn | gk << 8 | bps << 16)
                                + ((x2[0], x2[1], 0, x2[2], 0) if x2 is not None else (0, 0, 0, n_kt, 0)))
        # heaviest jobs first: the tail of the launch is then made of short jobs
        jobs.sort(key=lambda j: -(j[6] + j[7]) * (j[9] - j[8]))
        if job_bytes is not None and WGRAD_TAIL_FRAC > 0.0:
            # guided self-scheduling: the lightest jobs (they run last) are cut again, so that the CUs finish together
            n_tail = int(len(jobs) * WGRAD_TAIL_FRAC)
            head, tail = jobs[:len(jobs) - n_tail], jobs[len(jobs) - n_tail:]
            cut = []
            for j in tail:
                bps = (j[17] >> 16) & 255
                nst = -(-(j[9] - j[8]) // bps)
                parts = min(WGRAD_TAIL_PARTS, nst)
                bounds = [j[8] + (i * nst // parts) * bps for i in range(parts)] + [j[9]]
                for b0, b1 in zip(bounds[:-1], bounds[1:]):
                    if b1 > b0:
                        cut.append(j[:8] + (b0, b1) + j[10:])
            jobs = head + cut
            jobs.sort(key=lambda j: -(j[6] + j[7]) * (j[9] - j[8]))
        arr = np.zeros(len(jobs), dtype=L.DWJOB_DT)
        for i, j in enumerate(jobs):
            arr[i] = j
        # slab of every job in the launch's partials workspace (HnDwBatch.partials): its n_nt x n_kt accumulator tiles
        tiles = slab_tiles(arr)
        arr["p_tile"] = np.cumsum(tiles) - tiles
        return arr


# --------------------------------------------------------------------------------------------
# runtime
# --------------------------------------------------------------------------------------------
class _DevTables:
    pass


def wgrad_mode_word(mode: int) -> int:
    """The `mode` argument of hn_mlp_wgrad*: HN_MODE_BF16_S8 carries the dZ scale in bits 8.., the other modes the LDS
    stage (KiB) the job tables were cut for — the library refuses one that does not fit its ring (include/hn_kernels.h)."""
    return mode | (DZ_SCALE_LOG2 << 8) if mode == L.HN_MODE_BF16_S8 else mode | (WGRAD_STAGE_KB << 8)


# job size relative to WGRAD_JOB_BYTES for rectangles of >= 48 / >= 12 / fewer tiles (see wgrad_jobs)
# HN_WGRAD_JOB_SCALE="a,b,c" pins them (A/B runs: "1,1,1" = rounds 1-3); default: sized from the launch's bytes
WGRAD_JOB_SCALE = (tuple(float(x) for x in os.environ["HN_WGRAD_JOB_SCALE"].split(","))
                   if os.environ.get("HN_WGRAD_JOB_SCALE") else None)
N_CUS = 256                      # MI355X
# backward feature-gradient ops: 2 = only tiles with a differentiable feature, no chain-rule factor on identity-only tiles;
# 1 = only the tile skip; 0 = every tile of a group with a gradient (rounds 1-3)
AUX_TILE_SKIP = int(os.environ.get("HN_AUX_TILE_SKIP", 2))
# LDS stage of hn_wgrad_kernel's ring (bf16 / fp32 builds), KiB: as large as the build's ring allows (2 x 64 KiB; A/B knob)
WGRAD_STAGE_KB = min(int(os.environ.get("HN_WGRAD_STAGE_KB") or L.WGRAD_MAX_STAGE_KB), L.WGRAD_MAX_STAGE_KB)
WGRAD_JOBS_PER_CU = float(os.environ.get("HN_WGRAD_JOBS_PER_CU", 16.0))   # 0: the base job size never grows with the launch (round 4)
WGRAD_BIG_CAP = float(os.environ.get("HN_WGRAD_BIG_CAP", 8.0))     # largest job of a big rectangle, in units of WGRAD_JOB_BYTES
WGRAD_FUSE_SEGS = int(os.environ.get("HN_WGRAD_FUSE_SEGS", 1))     # 0: one job per input segment of a skip layer (rounds 1-3)
WGRAD_GRID = int(os.environ.get("HN_WGRAD_GRID", 1))       # 0: the wave grids of rounds 1-3 (Program._wave_grid)
WGRAD_TAIL_FRAC = float(os.environ.get("HN_WGRAD_TAIL_FRAC", 0.4))   # lightest 40 % of the jobs are halved: -1.8 % step time at config 2
WGRAD_TAIL_PARTS = int(os.environ.get("HN_WGRAD_TAIL_PARTS", 2))
WGRAD_JOB_BYTES = int(float(os.environ.get("HN_WGRAD_JOB_MB", 5)) * (1 << 20))     # stash bytes one job of a batched weight-gradient launch streams (~3 jobs per CU
                              # and step at config 2; keeps the atomic flushes at a few % of the traffic)


class PendingWgrad:
    """One program's share of a batched weight-gradient launch, not yet cut into jobs: the job sizes follow the bytes of
    the WHOLE launch (Program.wgrad_jobs), which are only known once every program of the backward pass has queued
    its share — `resolve_pending`.  Holds the stash alive until the launch ran."""

    def __init__(self, runner: "MlpRunner", mode: int, n_points: int, stash, grads, goffs, split):
        self.runner, self.mode, self.n_points, self.stash, self.grads = runner, mode, n_points, stash, grads
        self.goffs, self.split = goffs, split
        # per-block partial rows of a gathered GLO table's gradient that the backward machine STORED instead of adding
        # them by atomics (HnMlpArgs.embed_partial): {partial, idx, grad, n_blocks, spr, col_mask}; reduced with the slabs
        self.embed: Optional[dict] = None

    def stream_bytes(self, first_bucket_only: bool = True) -> float:
        return self.runner.prog.wgrad_stream_bytes(self.mode, self.n_points, self.goffs,
                                                   self.split if first_bucket_only else None)


def slab_tiles(jobs: np.ndarray) -> np.ndarray:
    """4-KiB slab tiles of every job in a partials workspace: its n_nt x n_kt dW tiles + one for its bias sums."""
    return jobs["n_nt"].astype(np.int64) * jobs["n_kt"] + (jobs["b_off"] >= 0)


class ResolvedWgrad:
    """A PendingWgrad cut into jobs (device table cached by the program's runner)."""

    def __init__(self, mode, jobs_dev, n_jobs, stash, grads, weights, bucket=0, jobs_host=None, owner=None):
        self.mode, self.jobs_dev, self.n_jobs, self.stash, self.grads = mode, jobs_dev, n_jobs, stash, grads
        self.owner = owner              # the MlpRunner whose `_jobs` entry holds jobs_dev: tables derived from a GROUP of
                                        # job tables (global order, reduce tables) live in the first share's owner
        self.weights = weights          # per job: stash tiles it streams (host numpy, for the global order)
        self.bucket = bucket            # 0 = launched at the end of backward, 1 = held (see WGRAD_SPLIT_OFFSET)
        self.jobs_host = jobs_host      # the job table (numpy): what the reduce tables of a partials launch are built from
        self.embed: Optional[dict] = None      # see PendingWgrad.embed (rides on the program's first share)


# A job ends with the flush of its dW rectangle.  Float atomics leave a CU at ~5 GB/s (one 256-B wave-instruction per
# ~50 ns): 100 us of a 650-us launch at config 2 (measured with a build that skips the flush: 0.648 -> 0.548 ms), during
# which the CU streams nothing.  With WGRAD_PARTIALS the jobs store their raw accumulator tiles into a workspace
# (HnDwBatch.partials) and ONE more launch (hn_mlp_wgrad_reduce, a workgroup per destination tile) sums the slabs and adds
# every gradient element once: 1/12 of the atomics, and sums whose order no longer depends on which job finished first.
WGRAD_PARTIALS = int(os.environ.get("HN_WGRAD_PARTIALS", 1))
BIAS_MFMA_BUILD = L.WGRAD_BIAS_MFMA      # A/B build: bias by all-ones MFMAs + atomics (asked of the library: hn_build_config)
_UID = itertools.count(1)
_ORPHAN_GROUP_CACHE: Dict[tuple, tuple] = {}      # shares built without an owner (tests that assemble ResolvedWgrad by hand)


def _uid(t: torch.Tensor) -> int:
    """Identity of a device table / gradient buffer that is NOT its address: the caching allocator hands the address of
    a freed model's job table to the next model's, and a table derived from the old one (destination offsets of the
    reduce launch) would then be used on the new one without any error (advisor, round 5)."""
    u = getattr(t, "_hn_uid", None)
    if u is None:
        u = t._hn_uid = next(_UID)
    return u


def _group_cache(grp: Sequence["ResolvedWgrad"]) -> dict:
    """Where the tables derived from this GROUP of job tables are kept: on the runner that owns the first share's job
    table, so that they are freed with it (the key names every table of the group by uid, so an entry can never be
    taken for another group's)."""
    owner = grp[0].owner
    if owner is None:
        return _ORPHAN_GROUP_CACHE
    return owner._group_tables


def _embed_struct(embeds: Sequence[dict]) -> "L.HnEmbedReduce":
    """HnEmbedReduce over the programs of one launch that gathered the SAME table."""
    e0 = embeds[0]
    em = L.HnEmbedReduce()
    em.grad, em.rows, em.dim = e0["grad"].data_ptr(), e0["grad"].shape[0], e0["grad"].shape[1]
    mask = 0
    for e in embeds:
        mask |= e["col_mask"]
    em.col_mask, em.n_src = mask, len(embeds)
    for i, e in enumerate(embeds):
        em.partial[i], em.idx[i] = e["partial"].data_ptr(), e["idx"].data_ptr()
        em.n_blocks[i], em.samples_per_ray[i] = e["n_blocks"], e["spr"]
    return em


def _launch_embed_reduce(mode: int, embeds: Sequence[dict]):
    em = _embed_struct(embeds)
    L.launch("hn_mlp_wgrad_reduce", C.c_int(wgrad_mode_word(mode)), None, C.c_int(0), None, None, C.c_int(0), C.byref(em),
             L.stream_handle())


def _reduce_tables(grp: Sequence["ResolvedWgrad"], device):
    """(tiles table, slab list, tiles) on the device for one batched launch: every destination tile (32 x 32 elements
    of one gradient matrix, across ALL programs of the launch that write it) with the slabs that hold a partial of it,
    in (program, job) order."""
    dest: Dict[tuple, list] = {}
    for k, p in enumerate(grp):
        jb = p.jobs_host
        gptr = p.grads.data_ptr()
        for ji in range(len(jb)):
            j = jb[ji]
            n_nt, n_kt = int(j["n_nt"]), int(j["n_kt"])
            if j["b_off"] >= 0 and p.mode != L.HN_MODE_BF16_S8 and not BIAS_MFMA_BUILD:     # bias record (ld = 0): the job's extra slab tile
                key = (gptr, int(j["b_off"]), 0, int(j["r0"]), n_nt, int(j["r_end"]), 0)
                dest.setdefault(key, [k]).append((k << 28) | (int(j["p_tile"]) + n_nt * n_kt))
            if j["w_off"] < 0:
                continue
            for i in range(n_nt):
                row = int(j["r0"]) + 32 * i
                if row >= j["r_end"] or row + 32 <= 0:
                    continue
                for jj in range(n_kt):
                    col = int(j["c0"]) + 32 * jj
                    if col >= j["c_end"] or col + 32 <= 0:
                        continue
                    key = (gptr, int(j["w_off"]), int(j["ld"]), row, col, int(j["r_end"]), int(j["c_end"]))
                    dest.setdefault(key, [k]).append((k << 28) | (int(j["p_tile"]) + i * n_kt + jj))
    tiles = np.zeros(len(dest), dtype=L.DWREDUCE_DT)
    lst: List[int] = []
    # heaviest first (most slabs): the tail of the launch is then made of short tiles
    for t, (key, v) in enumerate(sorted(dest.items(), key=lambda kv: -len(kv[1]))):
        tiles[t] = (v[0], key[1], key[2], key[3], key[4], key[5], key[6], len(lst), len(v) - 1)
        lst.extend(v[1:])
    # [3]: host-side extras of the table — the destination records themselves (what adam_rest_ranges proves the partition
    # of the arena with) and a cache of what is derived from them
    extras = {"tiles_host": tiles, "one_buffer": len({key[0] for key in dest}) <= 1}
    return (L.to_device_bytes(tiles, device), L.to_device_bytes(np.asarray(lst, dtype=np.uint32), device), len(tiles), extras)


ADAM_REST_CHUNK = 2048      # elements per `rest` workgroup of hn_mlp_wgrad_reduce_adam


def adam_rest_ranges(tiles_host: np.ndarray, numel: int, embed: Optional[Tuple[int, int, int, int]] = None):
    """The partition hn_mlp_wgrad_reduce_adam needs (include/hn_kernels.h HnAdamFuse.rest): how often every element of a
    gradient arena of `numel` floats is a destination of the reduce launch described by `tiles_host` (HnDwReduceTile
    records: 32 x 32 tiles of weight matrices, bias records with ld == 0) and `embed` = (offset, rows, dim, col_mask) of the
    gathered table's gradient.  Returns (ranges, coverage): ranges = int64 (n, 2) array of (start, len <= ADAM_REST_CHUNK)
    covering exactly the elements NOTHING writes, or None when some element is a destination twice (two records of one
    launch that overlap: the unfused reduce adds both contributions, a fused optimizer would step that element twice —
    the caller then keeps the two-launch form).  Pure host arithmetic (tests/test_host_api.py)."""
    cov = np.zeros(numel, dtype=np.uint8)
    for t in tiles_host:
        w_off, ld, row0, col0, r_end, c_end = (int(t[k]) for k in ("w_off", "ld", "row0", "col0", "r_end", "c_end"))
        if ld == 0:       # bias record: col0 = dZ tiles of the rectangle, rows row0 .. row0 + 32 col0 clipped to [0, r_end)
            lo, hi = max(row0, 0), min(row0 + 32 * col0, r_end)
            if hi > lo:
                cov[w_off + lo:w_off + hi] += 1
            continue
        r_lo, r_hi = max(row0, 0), min(row0 + 32, r_end)
        c_lo, c_hi = max(col0, 0), min(col0 + 32, c_end)
        if r_hi <= r_lo or c_hi <= c_lo:
            continue
        idx = w_off + np.arange(r_lo, r_hi, dtype=np.int64)[:, None] * ld + np.arange(c_lo, c_hi, dtype=np.int64)[None, :]
        cov[idx.reshape(-1)] += 1
    if embed is not None:
        off, rows, dim, mask = embed
        cols = np.array([c for c in range(dim) if (mask >> c) & 1], dtype=np.int64)
        if rows > 0 and len(cols):
            idx = off + np.arange(rows, dtype=np.int64)[:, None] * dim + cols[None, :]
            cov[idx.reshape(-1)] += 1
    if int(cov.max(initial=0)) > 1:
        return None, cov
    free = np.flatnonzero(cov == 0)
    out = []
    if len(free):
        cut = np.flatnonzero(np.diff(free) != 1) + 1
        for run in np.split(free, cut):
            a, n = int(run[0]), len(run)
            for o in range(0, n, ADAM_REST_CHUNK):
                out.append((a + o, min(ADAM_REST_CHUNK, n - o)))
    return np.asarray(out, dtype=np.int64).reshape(-1, 2), cov


class PendingReduce:
    """The second half of a batched weight-gradient launch (hn_mlp_wgrad_reduce), held back because an optimizer
    registered to consume it (optim.ArenaAdam(fuse_reduce=True) -> REDUCE_CONSUMERS): `fused(...)` runs it as
    hn_mlp_wgrad_reduce_adam, `plain()` as the reduce it would have been.  Keeps the slab workspaces, the batch array and
    the table tensors alive until then."""

    def __init__(self, mode, red, arr, n_grp, em, keep):
        self.mode, self.red, self.arr, self.n_grp, self.em, self.keep = mode, red, arr, n_grp, em, keep

    def plain(self):
        L.launch("hn_mlp_wgrad_reduce", C.c_int(wgrad_mode_word(self.mode)), L.ptr(self.red[0]), C.c_int(self.red[2]),
                 L.ptr(self.red[1]), self.arr, C.c_int(self.n_grp), C.byref(self.em) if self.em is not None else None,
                 L.stream_handle())

    def rest_table(self, grads: torch.Tensor):
        """(device table of HnAdamRange, n) for the arena behind `grads`, or None when the launch is not a partition of
        it (cached with the reduce tables; built outside stream captures only)."""
        extras = self.red[3]
        emb = None
        if self.em is not None:
            emb = ((int(self.em.grad) - grads.data_ptr()) // 4, int(self.em.rows), int(self.em.dim), int(self.em.col_mask))
        key = ("rest", grads.numel(), emb)
        hit = extras.get(key)
        if hit is None:
            if torch.cuda.is_current_stream_capturing():
                raise L.HnError("fused reduce + Adam: rest table first needed inside a stream capture (run one warm-up "
                                "step of the same shapes first)")
            ranges = None
            if extras["one_buffer"] and (emb is None or (0 <= emb[0] and emb[0] + emb[1] * emb[2] <= grads.numel())):
                ranges, _ = adam_rest_ranges(extras["tiles_host"], grads.numel(), emb)
            if ranges is None:
                hit = (None, -1)
            else:
                tab = np.zeros(len(ranges), dtype=L.ADAM_RANGE_DT)
                if len(ranges):
                    tab["start"], tab["len"] = ranges[:, 0], ranges[:, 1]
                hit = (L.to_device_bytes(tab, grads.device), len(ranges))
            extras[key] = hit
        return None if hit[1] < 0 else hit

    def fused(self, adam_struct):
        L.launch("hn_mlp_wgrad_reduce_adam", C.c_int(wgrad_mode_word(self.mode)), L.ptr(self.red[0]), C.c_int(self.red[2]),
                 L.ptr(self.red[1]), self.arr, C.c_int(self.n_grp), C.byref(self.em) if self.em is not None else None,
                 C.byref(adam_struct), L.stream_handle())


# uid of a gradient arena -> weak reference to the optimizer that consumes its reduce launch (ArenaAdam(fuse_reduce=True))
REDUCE_CONSUMERS: Dict[int, "weakref.ref"] = {}
_PENDING_REDUCE: Dict[int, PendingReduce] = {}


def _reduce_consumer(grads: torch.Tensor):
    u = getattr(grads, "_hn_uid", None)
    ref = REDUCE_CONSUMERS.get(u) if u is not None else None
    opt = ref() if ref is not None else None
    if ref is not None and opt is None:
        del REDUCE_CONSUMERS[u]
    return opt


def take_pending_reduce(grads: torch.Tensor) -> Optional[PendingReduce]:
    u = getattr(grads, "_hn_uid", None)
    return _PENDING_REDUCE.pop(u, None) if u is not None else None


def flush_pending_reduce(grads: Optional[torch.Tensor] = None):
    """Complete the gradient buffer(s) NOW with the plain reduce launch: whoever reads a gradient arena between backward
    and the optimizer step (an all-reduce, a test, gradient clipping) calls this first — ParamArena does for its own
    collectives and zero_grad."""
    if grads is not None:
        p = take_pending_reduce(grads)
        if p is not None:
            p.plain()
        return
    for u in list(_PENDING_REDUCE):
        _PENDING_REDUCE.pop(u).plain()


def drop_pending_reduce():
    _PENDING_REDUCE.clear()


# Data-parallel overlap (training.TrainStep / bench.py with more than one rank): when set to an offset (floats) into
# the flat gradient buffer, the deferred weight-gradient jobs of a backward pass are split in two launches — jobs
# whose matrix lies at or behind the offset (bucket 0: the template networks, the bulk of the bytes) run at the end
# of backward, the others (bucket 1: warp field, hyper sheet) are HELD until functional.flush_held_wgrads(); the
# all-reduce of bucket 0's slice then runs while bucket 1 is still being computed.
WGRAD_SPLIT_OFFSET: Optional[int] = None
HELD_JOB_DIV = int(os.environ.get("HN_HELD_JOB_DIV", 6))


# EXPERIMENT builds only (-DHN_WGRAD_PERSIST=1 exports hn_mlp_wgrad_batched_p): the persistent weight-gradient launch of
# round 6 — measured, no gain, closed (profiles/r06_wgrad_persistent.md); the product library does not carry it
WGRAD_PERSISTENT = os.environ.get("HN_WGRAD_PERSISTENT", "0") != "0"
_TICKETS: Dict[tuple, torch.Tensor] = {}


def _wgrad_tickets(device) -> torch.Tensor:
    """The persistent weight-gradient launch's job ticket + exit counter: two zeroed uint32 per (device, stream) — the
    kernel re-arms them itself, launches that share a pair must be stream-ordered."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    t = _TICKETS.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            key0 = next((k for k in _TICKETS if k[0] == str(device)), None)
            if key0 is None:
                raise L.HnError("persistent weight gradient: tickets first needed inside a stream capture (run a warm-up step)")
            return _TICKETS[key0]      # a capture on a side stream: it replays in the order of its warm-up, one launch at a time
        t = _TICKETS[key] = torch.zeros(2, dtype=torch.int32, device=device)
    return t


def resolve_pending(pending: Sequence[PendingWgrad]) -> List[ResolvedWgrad]:
    """Cut the shares of ONE launch into jobs.  The launch's bytes — the sum over the programs queued for it, first
    bucket only when the pass is split — size the jobs of big rectangles (one flush of a dW rectangle per CU);
    they are a function of the programs and sizes of THIS pass, so the first pass of a shape already builds the tables
    every later pass (and a graph capture after a single warm-up) uses."""
    total = float(sum(p.stream_bytes() for p in pending))
    out: List[ResolvedWgrad] = []
    for p in pending:
        first = True
        for b, (jd, nj, w, jh) in enumerate(p.runner.wgrad_tables(p.stash.device, p.mode, p.n_points, p.goffs, p.split, total)):
            if nj > 0:
                out.append(ResolvedWgrad(p.mode, jd, nj, p.stash, p.grads, w, bucket=b, jobs_host=jh, owner=p.runner))
                if first:
                    out[-1].embed, first = p.embed, False
        if first and p.embed is not None:           # no job at all (cannot happen for a program with parameters)
            _launch_embed_reduce(p.mode, [p.embed])
    return out


def launch_resolved_wgrads(shares: Sequence[ResolvedWgrad]):
    """hn_mlp_wgrad_batched over the programs of one backward pass (groups of HN_MAX_WGRAD_BATCH per mode).
    Jobs run in one global heaviest-first order (a device table cached per set of job lists): the hardware
    hands workgroups to CUs as they free up, i.e. list scheduling, and longest-first packs it tightest."""
    by_mode: Dict[int, List[ResolvedWgrad]] = {}
    for p in shares:
        by_mode.setdefault(p.mode, []).append(p)
    for mode, lst in by_mode.items():
        lst.sort(key=lambda p: -p.n_jobs)
        for i in range(0, len(lst), L.HN_MAX_WGRAD_BATCH):
            grp = lst[i:i + L.HN_MAX_WGRAD_BATCH]
            arr = (L.HnDwBatch * len(grp))()
            use_partials = bool(WGRAD_PARTIALS) and all(p.jobs_host is not None for p in grp)
            slabs = []
            for k, p in enumerate(grp):
                arr[k].jobs, arr[k].stash, arr[k].grads = p.jobs_dev.data_ptr(), p.stash.data_ptr(), p.grads.data_ptr()
                arr[k].n_jobs = p.n_jobs
                if use_partials:
                    n_tiles = int(slab_tiles(p.jobs_host).sum())
                    if n_tiles >= 1 << 28:
                        raise L.HnError("weight-gradient partials: more than 2^28 slab tiles in one program")
                    slabs.append(torch.empty(max(1, n_tiles) * 1024, dtype=torch.float32, device=p.stash.device))
                    arr[k].partials = slabs[-1].data_ptr()
            cache = _group_cache(grp)
            key = ("order",) + tuple((_uid(p.jobs_dev), p.n_jobs) for p in grp)
            order = cache.get(key)
            if order is None:
                if torch.cuda.is_current_stream_capturing():
                    raise L.HnError("weight-gradient job order: first use of this set of programs inside a stream "
                                    "capture (run one warm-up step of the same shapes first)")
                w = np.concatenate([p.weights for p in grp])
                ids = np.concatenate([(k << 24) | np.arange(p.n_jobs, dtype=np.int64) for k, p in enumerate(grp)])
                order = torch.from_numpy(ids[np.argsort(-w, kind="stable")].astype(np.int32)).to(grp[0].stash.device)
                cache[key] = order
            red = None
            if use_partials:
                rkey = ("reduce",) + key[1:] + tuple(_uid(p.grads) for p in grp)
                red = cache.get(rkey)
                if red is None:
                    if torch.cuda.is_current_stream_capturing():
                        raise L.HnError("weight-gradient reduce tables: first use of this set of programs inside a stream "
                                        "capture (run one warm-up step of the same shapes first)")
                    red = cache[rkey] = _reduce_tables(grp, grp[0].stash.device)
            if WGRAD_PERSISTENT and mode == L.HN_MODE_BF16 and hasattr(L.load(), "hn_mlp_wgrad_batched_p"):
                # one workgroup per CU walking the job list (hn_wgrad_persist_kernel): the next job's first stage in flight
                # under the current job's last products and flush
                L.launch("hn_mlp_wgrad_batched_p", C.c_int(wgrad_mode_word(mode)), arr, C.c_int(len(grp)), L.ptr(order),
                         C.c_void_p(L.timeline_slot("hn_mlp_wgrad_batched", grp[0].stash.device)),
                         L.ptr(_wgrad_tickets(grp[0].stash.device)), L.stream_handle(), tag="batched")
            else:
                L.launch("hn_mlp_wgrad_batched_t", C.c_int(wgrad_mode_word(mode)), arr, C.c_int(len(grp)), L.ptr(order),
                         C.c_void_p(L.timeline_slot("hn_mlp_wgrad_batched", grp[0].stash.device)), L.stream_handle(),
                         tag="batched")
            embeds = [p.embed for p in grp if p.embed is not None]
            by_table: Dict[int, list] = {}
            for e in embeds:
                by_table.setdefault(e["grad"].data_ptr(), []).append(e)
            tables = list(by_table.values())
            if red is not None and red[2] > 0:
                first = tables.pop(0) if tables else None
                em = _embed_struct(first) if first else None
                pend = PendingReduce(mode, red, arr, len(grp), em, (slabs, grp, first))
                g0 = grp[0].grads
                opt = _reduce_consumer(g0) if all(p.grads is g0 for p in grp) else None
                if opt is not None and len(by_mode) == 1 and len(lst) <= L.HN_MAX_WGRAD_BATCH:
                    # an optimizer consumes this arena's reduce (ArenaAdam(fuse_reduce=True)): held until its step()
                    # — which runs it as ONE launch with the update, hn_mlp_wgrad_reduce_adam — or until anything else
                    # needs the complete gradient (flush_pending_reduce).  At most one per arena: an earlier one (the
                    # previous chunk's backward pass) is completed first.
                    flush_pending_reduce(g0)
                    _PENDING_REDUCE[_uid(g0)] = pend
                else:
                    pend.plain()
            for tb in tables:           # a second table in one launch, or no slabs at all: a reduce of its own
                _launch_embed_reduce(mode, tb)


_OPT_STEPS = [0]
BACKWARD_SERIAL = [0]      # bumped by every MlpRunner.backward


def _count_optimizer_steps(optimizer, args, kwargs):
    _OPT_STEPS[0] += 1


def note_parameters_changed():
    """Invalidate every packed weight stream.  For updates no Python-visible counter sees: a HIP-graph replay that
    contains the optimizer launch changes the parameters without running ArenaAdam.step's body, so whoever replays
    such a graph (graphs.GraphedStep, training.TrainStep) calls this afterwards."""
    _OPT_STEPS[0] += 1


try:    # any torch optimizer step invalidates packed weights (inference after training; training repacks anyway)
    from torch.optim.optimizer import register_optimizer_step_post_hook
    register_optimizer_step_post_hook(_count_optimizer_steps)
except ImportError:     # pragma: no cover
    pass


def collect_pack_jobs(runners: Sequence["MlpRunner"], device, mode: int, force: bool = False, min_jobs: int = 2):
    """[(mode, HnPackJob array, [(runner, tables, key)])] for the stale weight streams among `runners` — the programs of
    one render step —, grouped for hn_pack_units_multi / hn_render_prologue (groups of fewer than `min_jobs` programs are
    left to the program's own pack launch).  `mark_packed(group)` after the launch that packed them.  Same rule per
    program as MlpRunner.pack."""
    todo = []
    for r in runners:
        m = r.effective_mode(mode)
        d = r._tables(device, m)
        need, key = r._needs_pack(d, force)
        if need:
            todo.append((r, m, d, key))
    by_mode: Dict[int, list] = {}
    for item in todo:
        by_mode.setdefault(item[1], []).append(item)
    out = []
    for m, items in by_mode.items():
        if len(items) < min_jobs:
            continue                     # a single program: its own pack launch does it
        for i in range(0, len(items), L.HN_MAX_PACK_JOBS):
            grp = items[i:i + L.HN_MAX_PACK_JOBS]
            arr = (L.HnPackJob * len(grp))()
            for k, (r, _m, d, key) in enumerate(grp):
                r._pack_job(d, device)
                arr[k].units, arr[k].ptrs, arr[k].wstream = d.units.data_ptr(), d.ptrs.data_ptr(), d.wstream.data_ptr()
                arr[k].bias, arr[k].bias_out = d.bias_desc.data_ptr(), d.bias.data_ptr()
                arr[k].n_units, arr[k].n_bias = d.n_units, d.n_bias
            out.append((m, arr, [(r, d, key) for r, _m, d, key in grp]))
    return out


def mark_packed(group):
    for r, d, key in group:
        d.pack_backward = BACKWARD_SERIAL[0]
        d.pack_key = key


def pack_many(runners: Sequence["MlpRunner"], device, mode: int, force: bool = False):
    """Pack the weight streams of several programs that are about to run — the programs of one render step — with ONE
    launch (hn_pack_units_multi) where more than one of them needs it; each runner's own `pack` then finds its streams
    fresh."""
    for m, arr, grp in collect_pack_jobs(runners, device, mode, force):
        L.launch("hn_pack_units_multi", C.c_int(m), arr, C.c_int(len(grp)), L.stream_handle(),
                 tag="+".join(r.prog.name for r, *_ in grp))
        mark_packed(grp)


class MlpRunner:
    """Owns the device copies of a Program's tables and launches the three kernels."""

    def __init__(self, program: Program):
        self.prog = program
        self._dev: Dict[Tuple[str, int], _DevTables] = {}
        self._jobs: Dict[Tuple[str, int, int], Tuple[torch.Tensor, int]] = {}
        self._group_tables: Dict[tuple, object] = {}      # launch_resolved_wgrads: job order / reduce tables of the groups this runner leads

    def _tables(self, device, mode) -> _DevTables:
        key = (str(device), mode)
        d = self._dev.get(key)
        if d is None:
            ht = self.prog.host_tables(mode)
            d = _DevTables()
            d.units = L.to_device_bytes(np.concatenate([ht["fwd_units"], ht["bwd_units"]]), device)
            d.n_fwd_units = len(ht["fwd_units"])
            d.n_units = d.n_fwd_units + len(ht["bwd_units"])
            d.fwd_chunks, d.bwd_chunks = ht["fwd_chunks"], ht["bwd_chunks"]
            d.bias_desc = L.to_device_bytes(ht["bias"], device)
            d.n_bias = len(ht["bias"])
            d.feat = L.to_device_bytes(ht["feat"], device)
            d.comps = L.to_device_bytes(ht["comps"], device)
            d.fwd_ops = L.to_device_bytes(self.prog.fwd_ops, device)
            d.bwd_ops = L.to_device_bytes(self.prog.bwd_ops, device)
            d.wstream = torch.empty(d.n_units * 1024, dtype=torch.uint8, device=device)
            d.bias = torch.zeros(max(32, self.prog.bias_len), dtype=torch.float32, device=device)
            d.sel = torch.from_numpy(self.prog.selection_matrix()).to(device)
            d.ptr_key, d.ptrs, d.pack_key, d.pack_backward = None, None, None, -1
            self._dev[key] = d
        return d

    def _param_key(self):
        # Tensor._version does not see every update (fused optimizers write parameters without bumping it), so the
        # key also carries the arena version and a global count of optimizer steps (see _count_optimizer_steps);
        # training forwards repack unconditionally.
        key = [_OPT_STEPS[0]]
        for p in self.prog.params:
            tag = getattr(p, "_hn_arena", None)
            key.append((p.data_ptr(), p._version, tag[0].version() if tag is not None else 0))
        return tuple(key)

    def _needs_pack(self, d, force: bool):
        """(packing needed?, current parameter key) — the rule of `pack` below."""
        key = self._param_key()
        return not (d.pack_key == key and not (force and d.pack_backward != BACKWARD_SERIAL[0])), key

    def _pack_job(self, d, device):
        """The arguments of one pack launch for this program (pointer table refreshed if a parameter moved)."""
        for p in self.prog.params:
            L.require_gpu(p)
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise L.HnError("parameters must be contiguous fp32 tensors")
        pk = tuple(p.data_ptr() for p in self.prog.params) + (d.sel.data_ptr(),)
        if d.ptr_key != pk:
            d.ptrs = torch.tensor(list(pk), dtype=torch.int64).to(device)
            d.ptr_key = pk
        return d

    def pack(self, device, mode, force: bool = False):
        """(Re)pack both weight streams if a parameter is known to have changed, and — with `force`, i.e. on training
        forwards — also whenever a backward pass of ANY program has run since the last pack: parameters change
        between a backward and the next forward, and a fused optimizer step leaves no trace on the tensors.  (The
        coarse- and fine-level forwards of one step therefore share one pack.)"""
        d = self._tables(device, mode)
        need, key = self._needs_pack(d, force)
        if not need:
            return d
        d.pack_backward = BACKWARD_SERIAL[0]
        self._pack_job(d, device)
        L.launch("hn_pack_units", C.c_int(mode), L.ptr(d.units), C.c_int(d.n_units), L.ptr(d.ptrs), L.ptr(d.wstream),
                 L.ptr(d.bias_desc), C.c_int(d.n_bias), L.ptr(d.bias), L.stream_handle(), tag=self.prog.name)
        d.pack_key = key
        return d

    def _args(self, d, mode, n_points, samples_per_ray, training, ops, n_ops, wstream_ptr, n_chunks, srcs, dsts,
              stash, masks, dsrc, embed=None, kind="hn_mlp_forward"):
        a = L.HnMlpArgs()
        a.mode, a.n_points, a.samples_per_ray, a.training = mode, n_points, samples_per_ray, int(training)
        a.n_ops, a.n_chunks, a.n_dsrc = n_ops, n_chunks, self.prog.n_dsrc if dsrc is not None else 0
        a.ops, a.wstream, a.bias, a.feat = ops.data_ptr(), wstream_ptr, d.bias.data_ptr(), d.feat.data_ptr()
        a.n_bias, a.n_feat = max(32, self.prog.bias_len), max(1, len(self.prog.feat_table))
        a.max_groups = max([ly.aux.groups for ly in self.prog.layers if ly.aux is not None], default=0)
        a.prof = L.PROF_BUFFER.data_ptr() if L.PROF_BUFFER is not None else 0
        a.timeline = L.timeline_slot(f"{kind}[{self.prog.name}]", d.bias.device)
        a.comps, a.n_comps = d.comps.data_ptr(), len(self.prog.comp_map)
        # (HN_FORCE_WIDE=1: A/B knob — take the kernel build that carries the wide ops although the program has none)
        wide = any(ly.out is not None and ly.out.wide for ly in self.prog.layers) or _FORCE_WIDE
        direct = any(f.kind != L.HN_FEAT_ZERO and (f.src, f.comp) not in self.prog.comp_map for f in self.prog.feat_table)
        a.wide_ops = (1 if wide else 0) | (2 if (direct or _FORCE_WIDE) else 0)
        a.dz_scale_log2 = DZ_SCALE_LOG2 if mode == L.HN_MODE_BF16_S8 else 0
        a.n_trig_comps = min(len(self.prog.comp_map), max(1, self.prog.n_trig_comps))
        # hi + lo planes of x / 2pi for the encoded components when the forward's LDS budget allows (158 KiB: ring +
        # bias / feature tables + 8 waves x planes x 128 B), else hi alone (the one-FMA accuracy of rounds 1-2)
        n_feat, n_bias = max(1, len(self.prog.feat_table)), max(32, self.prog.bias_len)
        fixed = 2 * CHUNK * 1024 + ((n_bias + 3) & ~3) * 4 + ((n_feat + 1) & ~1) * 8 + n_feat * 16
        full = fixed + 8 * (a.n_comps + 2 * a.n_trig_comps) * 128
        a.trig_lo_planes = 1 if full <= 158 * 1024 else 0
        a.stash = stash.data_ptr() if stash is not None else 0
        a.masks = masks.data_ptr() if masks is not None else 0
        a.dsrc = dsrc.data_ptr() if dsrc is not None else 0
        for i, s in enumerate(srcs):
            if s is None:
                continue
            t, per_ray = s[0], s[1]
            L.require_gpu(t)
            if t.dtype != torch.float32 or t.stride(-1) != 1:
                raise L.HnError("sources must be fp32 with unit inner stride")
            a.src[i].ptr, a.src[i].ld, a.src[i].per_ray = t.data_ptr(), t.stride(-2) if t.dim() > 1 else 1, int(per_ray)
            if len(s) > 2 and s[2] is not None:       # gathered per-ray source: row = idx[ray] of a (rows, dim) table
                idx = s[2]
                L.require_gpu(idx)
                if idx.dtype != torch.int64 or not idx.is_contiguous():
                    raise L.HnError("gather indices must be contiguous int64")
                a.src[i].gather_idx, a.src[i].gather_rows = idx.data_ptr(), t.shape[0]
        if embed is not None:       # (gradient table, indices, source index[, partial rows]): GLOEmbed's backward inside the machine
            g_tab, idx, src_i = embed[:3]
            if len(embed) > 3 and embed[3] is not None:
                a.embed_partial = embed[3].data_ptr()
            mask, col = self.prog.embed_fold(src_i)
            a.embed_reg_mask, a.embed_grad, a.embed_idx = mask, g_tab.data_ptr(), idx.data_ptr()
            a.embed_rows, a.embed_dim = g_tab.shape[0], g_tab.shape[1]
            for k in range(L.HN_DSRC_COMPS):
                a.embed_col[k] = col[k]
        for i, t in enumerate(dsts):
            if t is None:
                continue
            L.require_gpu(t)
            a.dst[i].ptr, a.dst[i].ld = t.data_ptr(), t.stride(-2) if t.dim() > 1 else 1
        return a

    def _ops(self, device, mode, n_points):
        """Device copies of the op lists resolved for `n_points` (cached: a model sees a handful of sizes)."""
        key = ("ops", str(device), mode, n_points)
        hit = self._jobs.get(key)
        if hit is None:
            fwd, bwd = self.prog.resolved_ops(mode, n_points)
            hit = (L.to_device_bytes(fwd, device), L.to_device_bytes(bwd, device))
            self._jobs[key] = hit
        return hit

    @staticmethod
    def _job_bytes(mode: int) -> int:
        """WGRAD_JOB_BYTES is quoted for 2-KiB tiles; a job of the 8-bit stash covers the same point blocks (same
        number of jobs per launch, same products and atomics per job) in half the bytes."""
        return WGRAD_JOB_BYTES // 2 if mode == L.HN_MODE_BF16_S8 else WGRAD_JOB_BYTES

    def effective_mode(self, mode: int) -> int:
        """HN_MODE_BF16_S8 exists in the render-level kernel builds only: a program with stand-alone-module paths (wide
        outputs, directly read identity features) runs, stash included, in plain HN_MODE_BF16."""
        if mode == L.HN_MODE_BF16_S8:
            wide = any(ly.out is not None and ly.out.wide for ly in self.prog.layers) or _FORCE_WIDE
            direct = any(f.kind != L.HN_FEAT_ZERO and (f.src, f.comp) not in self.prog.comp_map
                         for f in self.prog.feat_table)
            if wide or direct:
                return L.HN_MODE_BF16
        return mode

    def forward(self, mode, n_points, samples_per_ray, srcs, dsts, training: bool):
        """Launch the forward machine.  Returns (stash, masks) (None, None when not training)."""
        mode = self.effective_mode(mode)
        device = dsts[0].device if dsts and dsts[0] is not None else srcs[0][0].device
        d = self.pack(device, mode, force=training)
        stash = masks = None
        if training:
            _, sb, mb = self.prog.layout(mode, n_points)
            stash = torch.empty(max(sb, 16), dtype=torch.uint8, device=device)
            masks = torch.empty(max(mb, 16), dtype=torch.uint8, device=device)
        a = self._args(d, mode, n_points, samples_per_ray, training, self._ops(device, mode, n_points)[0],
                       len(self.prog.fwd_ops), d.wstream.data_ptr(), d.fwd_chunks, srcs, dsts, stash, masks, None)
        L.launch("hn_mlp_forward", C.byref(a), L.stream_handle(), tag=self.prog.name)
        return stash, masks

    def backward(self, mode, n_points, samples_per_ray, srcs, stash, masks, grad_target=None, defer=False,
                 embed=None, want_dsrc=True):
        """Launch backward-data then the weight-gradient kernel.
        grad_target = (flat fp32 buffer, per-parameter offsets): accumulate the weight gradients there (a
        ParamArena's gradient buffer) instead of into a fresh zero-filled buffer.  With `defer` (and a grad_target)
        the weight-gradient kernel is NOT launched: a list holding one PendingWgrad is returned in place of the
        gradient buffer, for `resolve_pending` / `launch_resolved_wgrads`.
        embed = (gradient table (rows, dim), int64 ray indices, source index): reduce the source gradient of that
        gathered per-ray source inside the kernel and scatter-add it into the table gradient (needs
        samples_per_ray % 32 == 0).  want_dsrc=False skips the per-point source-gradient tensor altogether.
        Returns (dsrc [P, n_dsrc] or None, flat fp32 gradient buffer or None when grad_target was given)."""
        device = stash.device
        mode = self.effective_mode(mode)
        BACKWARD_SERIAL[0] += 1
        d = self._tables(device, mode)       # the streams its forward packed
        dsrc = None
        if self.prog.n_dsrc > 0 and want_dsrc:
            dsrc = torch.empty(n_points, self.prog.n_dsrc, dtype=torch.float32, device=device)
        if embed is not None and samples_per_ray % 32 != 0:
            raise L.HnError("the in-kernel embedding gradient needs samples_per_ray % 32 == 0")
        a = self._args(d, mode, n_points, samples_per_ray, True, self._ops(device, mode, n_points)[1],
                       len(self.prog.bwd_ops), d.wstream.data_ptr() + d.n_fwd_units * 1024, d.bwd_chunks, srcs, [],
                       stash, masks, dsrc, embed, kind="hn_mlp_backward")
        L.launch("hn_mlp_backward", C.byref(a), L.stream_handle(), tag=self.prog.name)
        goffs = tuple(grad_target[1]) if grad_target is not None else None
        deferred = defer and grad_target is not None
        if deferred:        # the caller launches it together with the other programs of this backward pass
            return dsrc, [PendingWgrad(self, mode, n_points, stash, grad_target[0], goffs, WGRAD_SPLIT_OFFSET)]
        jobs_dev, n_jobs, weights, _ = self.wgrad_tables(device, mode, n_points, goffs, None, None)[0]
        if grad_target is not None:
            grads, ret = grad_target[0], None
        else:
            _, gtot = self.prog.grad_offsets()
            grads = ret = torch.zeros(gtot, dtype=torch.float32, device=device)
        L.launch("hn_mlp_wgrad", C.c_int(wgrad_mode_word(mode)), L.ptr(jobs_dev), C.c_int(n_jobs), L.ptr(stash), L.ptr(grads),
                 L.stream_handle(), tag=self.prog.name)
        return dsrc, ret

    def wgrad_tables(self, device, mode: int, n_points: int, goffs, split: Optional[int],
                     launch_bytes: Optional[float]):
        """[(device job table | None, jobs, stash tiles per job)] per bucket of this program's weight-gradient jobs
        (cached).  `launch_bytes` = stash bytes of the WHOLE batched launch the jobs will run in (resolve_pending; None: a
        launch of its own, `target_jobs` equal jobs) — quantised in steps of 2^(1/4) for the cache.  `split`: two
        buckets (WGRAD_SPLIT_OFFSET)."""
        batched = launch_bytes is not None
        qh = round(4.0 * math.log2(launch_bytes)) if batched and launch_bytes > 0 else 0
        jkey = ("jobs", str(device), mode, n_points, goffs, batched, split, qh)
        entry = self._jobs.get(jkey)
        if entry is None:
            if torch.cuda.is_current_stream_capturing():
                # a table upload is a pageable host-to-device copy: never inside a capture (its buffer would also come
                # from the graph's private pool)
                raise L.HnError(f"{self.prog.name}: weight-gradient job table for {n_points} points first needed inside "
                                "a stream capture (run one warm-up step of the same shapes first)")
            jobs = self.prog.wgrad_jobs(mode, n_points, grad_offsets=goffs,
                                        job_bytes=self._job_bytes(mode) if batched else None,
                                        launch_bytes=(2.0 ** (qh / 4.0)) if qh else None)
            parts = [jobs]
            if split is not None:
                # the held bucket runs as a launch of its own, next to the all-reduce of the first: it holds ~1/6 of
                # the stash bytes, so its jobs are cut finer to still give every CU a few of them
                fine = self.prog.wgrad_jobs(mode, n_points, grad_offsets=goffs, job_bytes=self._job_bytes(mode) // HELD_JOB_DIV)
                parts = [jobs[jobs["w_off"] >= split], fine[fine["w_off"] < split]]
            entry = []
            for part in parts:
                part = np.ascontiguousarray(part)
                weights = ((part["n_nt"] + part["n_kt"]).astype(np.int64) * (part["blk1"] - part["blk0"]))
                if len(part):       # a bucket's slabs are numbered from 0 in its own workspace
                    tl = slab_tiles(part)
                    part["p_tile"] = np.cumsum(tl) - tl
                entry.append((L.to_device_bytes(part, device) if len(part) else None, len(part), weights, part))
            self._jobs[jkey] = entry
        return entry

    def split_grads(self, flat: torch.Tensor) -> List[torch.Tensor]:
        offs, _ = self.prog.grad_offsets()
        return [flat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, self.prog.params)]
