"""Lane-level numpy emulation of the MLP machine (hn_mlp.hip) — test infrastructure.

It re-implements, with the MFMA operand/accumulator lane maps documented for gfx950, exactly what
the device code does with the HOST tables (packing descriptors, op lists, slot layout, dW jobs):
weight packing, the forward machine, the transposed stashes, the backward-data machine and the
weight-gradient kernel.  Values stay in float64 (no bf16 rounding) so results must equal a plain
numpy MLP to ~1e-12: any layout / ordering mistake in the host compiler shows up as O(1) error
without needing a GPU.
"""
import numpy as np
from hypernerf_torch_amd import _lib as L

CHUNK = 32
LANES = np.arange(64)
R = LANES & 31
H = LANES >> 5


def rho(i, h):
    return (i & 3) + 8 * (i >> 2) + 4 * h


def pi16(h, j):
    return 8 * (j >> 2) + 4 * h + (j & 3)


# ---- MFMA semantics ---------------------------------------------------------------------------
def mfma_bf16(a, b, c):
    """a,b: (64,8); c: (64,16).  A[i][k=8h+j] = a[lane(i,h)][j]; B[k][n] = b[lane(n,h)][j]."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for l in range(64):
        for j in range(8):
            A[l & 31, 8 * (l >> 5) + j] = a[l, j]
            B[8 * (l >> 5) + j, l & 31] = b[l, j]
    D = A @ B
    out = c.copy()
    for l in range(64):
        for q in range(16):
            out[l, q] += D[rho(q, l >> 5), l & 31]
    return out


def mfma_f32(a, b, c):
    """a,b: (64,) ; A[i][k=h], B[k=h][n]."""
    A = np.zeros((32, 2)); B = np.zeros((2, 32))
    for l in range(64):
        A[l & 31, l >> 5] = a[l]
        B[l >> 5, l & 31] = b[l]
    D = A @ B
    out = c.copy()
    for l in range(64):
        for q in range(16):
            out[l, q] += D[rho(q, l >> 5), l & 31]
    return out


class Mode:
    def __init__(self, bf16):
        self.bf16 = bf16
        self.steps32 = 2 if bf16 else 16
        self.units32 = 2 if bf16 else 4
        self.tile_units = 2 if bf16 else 4
        self.elems = 8 if bf16 else 4     # values per lane per 1-KiB unit

    def zero_frags(self, n):
        return np.zeros((n, 64, 8)) if self.bf16 else np.zeros((n, 64))


# ---- pack kernel ---------------------------------------------------------------------------------
def pack_units(mode, units, params):
    out = np.zeros((len(units), 64, mode.elems))
    for ui, u in enumerate(units):
        if u["w_id"] < 0:
            continue
        W = params[u["w_id"]]
        for l in range(64):
            row, h = l & 31, l >> 5
            for e in range(mode.elems):
                k = u["k0"] + pi16(h, e) if mode.bf16 else rho(u["k0"] + e, h)
                if u["transposed"]:
                    sr, sc = u["r0"] + k, u["c0"] + row
                else:
                    sr, sc = u["r0"] + row, u["c0"] + k
                if 0 <= sr < u["r_end"] and 0 <= sc < u["c_end"]:
                    out[ui, l, e] = W.reshape(-1)[sr * u["ld"] + sc]
    return out


def pack_bias(descs, params, total):
    out = np.zeros(total)
    for d in descs:
        if d["w_id"] >= 0:
            out[d["off"]:d["off"] + d["n"]] = params[d["w_id"]].reshape(-1)[:d["n"]]
    return out


class WStream:
    def __init__(self, units):
        self.units = units
        self.ctr = 0

    def take(self, n):
        if (self.ctr % CHUNK) + n > CHUNK:
            self.ctr = (self.ctr + CHUNK - 1) // CHUNK * CHUNK
        p = self.ctr
        self.ctr += n
        return self.units[p:p + n]


def mma_block32(mode, acc, units, frags):
    """acc (64,16) += W[32x32] . in ; units: UNITS32 packed units ; frags: STEPS32 fragments."""
    if mode.bf16:
        for u in range(2):
            acc = mfma_bf16(units[u], frags[u], acc)
    else:
        for u in range(4):
            for e in range(4):
                acc = mfma_f32(units[u][:, e], frags[4 * u + e], acc)
    return acc


def gemm_blocks(mode, acc, ws, frags, k32):
    if k32 == 0:
        return acc
    un = ws.take(k32 * mode.units32)
    for k in range(k32):
        acc = mma_block32(mode, acc, un[k * mode.units32:(k + 1) * mode.units32],
                          frags[k * mode.steps32:(k + 1) * mode.steps32])
    return acc


def acc_to_frags(mode, acc):
    if mode.bf16:
        return np.stack([acc[:, 0:8], acc[:, 8:16]])       # (2,64,8)
    return acc.T.copy()                                     # (16,64)


def transpose_tile(mode, frags):
    z = np.zeros((64, 16))
    if mode.bf16:
        for u in range(2):
            ident = np.zeros((64, 8))
            for l in range(64):
                for j in range(8):
                    ident[l, j] = 1.0 if (l & 31) == 16 * u + pi16(l >> 5, j) else 0.0
            z = mfma_bf16(frags[u], ident, z)
    else:
        for q in range(16):
            ident = np.array([1.0 if (l & 31) == rho(q, l >> 5) else 0.0 for l in range(64)])
            z = mfma_f32(frags[q], ident, z)
    return z


def stash_slot(r, h, u):
    """hn_stash_slot (hn_mlp.hip): 16-byte slot of lane (r, h) inside unit u of a bf16 stash tile."""
    return 32 * h + (r ^ (4 * h + 8 * u))


def dw_tr_offsets(lane):
    """hn_dw_tr_offsets (hn_mlp.hip): byte offsets of a lane's two transposed reads (jh = 0, 1) inside a tile."""
    G, i = lane >> 4, lane & 15
    q, pp = i >> 2, i & 3
    u, hh, h, g = G & 1, G >> 1, pp & 1, pp >> 1
    return (u * 1024 + stash_slot(8 * hh + q, h, u) * 16 + 8 * g,
            u * 1024 + stash_slot(8 * hh + 4 + q, h, u) * 16 + 8 * g)


def ds_read_b64_tr_b16(img, addr):
    """gfx950 semantics (cdna_hip_programming.md T10): per group of 16 lanes, lane 4q+p supplies the address of row q,
    columns 4p..4p+3 (8 bytes) of a 4 x 16 block; lane i receives column i, row q in element q.  img: one entry per
    16-bit element; addr: byte address per lane.  Also checks the bank rule the layout was designed for: the 32 lanes
    of a half must touch 64 distinct banks (bank = (addr / 4) % 64, two banks per 8-byte piece)."""
    out = np.zeros((64, 4))
    for half in range(2):
        banks = set()
        for l in range(32 * half, 32 * half + 32):
            assert addr[l] % 8 == 0
            banks.update({(addr[l] // 4) % 64, (addr[l] // 4 + 1) % 64})
        assert len(banks) == 64, "transposed stash read is not bank-conflict free"
    for G in range(4):
        for i in range(16):
            p, e = i >> 2, i & 3
            for q in range(4):
                src = addr[16 * G + 4 * q + p]
                out[16 * G + i, q] = img[src // 2 + e]
    return out


# ---- HN_MODE_BF16_S8: 1-KiB tiles of 8-bit elements (csrc/hn_mlp.hip: hn_stash8_slot, hn_dw8_offset, hn_dw8_feature) ----
def stash8_slot(r, h):
    return (r & 3) + 4 * h + 8 * ((r >> 2) & 1) + 16 * (r >> 3)


def dw8_offset(lane):
    G, i = lane >> 4, lane & 15
    return stash8_slot(8 * (G >> 1) + (i >> 1), i & 1) * 16 + 8 * (G & 1)


def dw8_feature(c):
    return rho(8 * (c >> 4) + (c & 7), (c >> 3) & 1)


def ds_read_b64_tr_b8(img, addr):
    """gfx950 semantics as measured by tools/tr_b8_probe.hip: per group of 16 lanes, 8 rows x 16 bytes, row b = the 8 bytes
    at the address of lane 2b followed by the 8 bytes at the address of lane 2b+1; lane i receives column i (byte b of
    its result = row b).  img: one entry per byte.  Same bank rule as the 16-bit form: the 32 lanes of a half must touch
    64 distinct banks."""
    out = np.zeros((64, 8))
    for half in range(2):
        banks = set()
        for l in range(32 * half, 32 * half + 32):
            assert addr[l] % 8 == 0
            banks.update({(addr[l] // 4) % 64, (addr[l] // 4 + 1) % 64})
        assert len(banks) == 64, "transposed 8-bit stash read is not bank-conflict free"
    for G in range(4):
        for i in range(16):
            for b in range(8):
                out[16 * G + i, b] = img[addr[16 * G + 2 * b + (i >> 3)] + (i & 7)]
    return out


def store_tile(mode, z):
    """-> (TILE_UNITS, 64, elems) as written to the stash."""
    if mode.bf16:
        return np.stack([z[:, 0:8], z[:, 8:16]])
    return np.stack([z[:, 4 * g:4 * g + 4] for g in range(4)])


# ---- features ---------------------------------------------------------------------------------------
COMPS = None     # staged component table of the program under emulation (src << 16 | column)


def _source_value(e, srcs, p, ray):
    if ((e["packed"] >> 12) & 15) == 5:      # HN_FEAT_ID_DIRECT
        arr, per_ray = srcs[(e["packed"] >> 8) & 15]
        return arr[ray if per_ray else p, (int(e["packed"]) >> 24) & 255]
    c = int(COMPS[e["packed"] & 255])
    src = srcs[c >> 16]
    arr, per_ray = src[0], src[1]
    if len(src) > 2 and src[2] is not None:          # gathered per-ray source: row = idx[ray]
        return arr[int(src[2][ray]), c & 0xffff]
    return arr[ray if per_ray else p, c & 0xffff]


def feature_value(e, srcs, p, ray):
    kind = (e["packed"] >> 12) & 15
    if kind == 0:
        return 0.0
    x = _source_value(e, srcs, p, ray)
    if kind in (1, 5):
        return x
    arg = e["freq"] * x
    if kind == 4:
        arg = arg + 0.5 * 3.1415926
    return np.cos(arg) if kind == 3 else np.sin(arg)


def feature_grad(e, srcs, p, ray):
    kind = (e["packed"] >> 12) & 15
    if kind in (1, 5):
        return 1.0
    x = _source_value(e, srcs, p, ray)
    arg = e["freq"] * x
    if kind == 4:
        arg = arg + 0.5 * 3.1415926
    return -e["freq"] * np.sin(arg) if kind == 3 else e["freq"] * np.cos(arg)


def make_group(mode, ft, srcs, p_of_lane, ray_of_lane, valid):
    fr = mode.zero_frags(2 * mode.steps32)
    for l in range(64):
        h = l >> 5
        if mode.bf16:
            for s in range(4):
                for j in range(8):
                    v = feature_value(ft[16 * s + pi16(h, j)], srcs, p_of_lane[l], ray_of_lane[l])
                    fr[s, l, j] = v     # padded lanes replicate the last point, like the device
        else:
            for s in range(32):
                v = feature_value(ft[32 * (s >> 4) + rho(s & 15, h)], srcs, p_of_lane[l], ray_of_lane[l])
                fr[s, l] = v
    return fr


class Stash:
    def __init__(self, mode, prog, n_points):
        self.mode = mode
        self.offs, sb, mb = prog.layout(1 if mode.bf16 else 0, n_points)
        self.tile_bytes = 2048 if mode.bf16 else 4096
        self.tiles = {}     # (byte offset) -> (TILE_UNITS,64,elems)
        self.masks = {}

    def put_frags(self, slot, blk, t, frags):
        """hn_stash: fp32 transposes the tile through the matrix core and stores [g][lane][4]; bf16 stores the operand
        fragments as they are, lane (r, h) of unit u at 16-byte slot stash_slot(r, h, u) of the unit."""
        off, nt = self.offs[slot]
        key = off + (blk * nt + t) * self.tile_bytes
        if not self.mode.bf16:
            self.tiles[key] = store_tile(self.mode, transpose_tile(self.mode, frags))
            return
        img = np.zeros(1024)                         # the 2-KiB tile, one entry per bf16 element
        for u in range(2):
            for l in range(64):
                a = u * 1024 + stash_slot(l & 31, l >> 5, u) * 16
                img[a // 2:a // 2 + 8] = frags[u][l]
        self.tiles[key] = img

    def get_tile(self, byte_off):
        """The operand fragments the weight-gradient kernel builds from one stashed tile (DwFrag::load)."""
        if not self.mode.bf16:
            return self.tiles[byte_off]
        img = self.tiles[byte_off]
        out = np.zeros((2, 64, 8))
        for mm in range(2):
            for jh in range(2):
                addr = [dw_tr_offsets(l)[jh] + 256 * mm for l in range(64)]
                got = ds_read_b64_tr_b16(img, addr)
                out[mm][:, 4 * jh:4 * jh + 4] = got
        return out

    def put_mask(self, slot, blk, d, bits):
        off, nt = self.offs[slot]
        self.masks[(off, blk * nt + d)] = bits.copy()

    def get_mask(self, slot, blk, d):
        off, nt = self.offs[slot]
        return self.masks[(off, blk * nt + d)]


def bias_acc(bias, off, t):
    acc = np.zeros((64, 16))
    for l in range(64):
        for i in range(16):
            acc[l, i] = bias[off + 32 * t + rho(i, l >> 5)]
    return acc


def run_forward(prog, mode, tables, params, srcs, n_points, spr, dst_widths, training=True):
    global COMPS
    COMPS = tables["comps"]
    units = pack_units(mode, tables["fwd_units"], params)
    bias = pack_bias(tables["bias"], params, prog.bias_len)
    feat = tables["feat"]
    outs = [np.zeros((n_points, w)) for w in dst_widths]
    # a source without data is published by the program itself (HN_OP_OUT w7): the published components live in the
    # block's staged-component store on the device; here they alias the output tensor they are also written to
    srcs = list(srcs)
    for i, s_ in enumerate(srcs):
        if s_ is None and any(ly.out is not None and ly.out.publish is not None and ly.out.publish[0] == i
                              for ly in prog.layers):
            pub = [ly.out for ly in prog.layers if ly.out is not None and ly.out.publish is not None
                   and ly.out.publish[0] == i]
            assert all(o.col == o.publish[1] for o in pub), "emulator: published columns must equal output columns"
            srcs[i] = (outs[pub[0].dst], False)
    stash = Stash(mode, prog, n_points)
    nblk = (n_points + 31) // 32
    for blk in range(nblk):
        p0 = blk * 32 + R
        valid = p0 < n_points
        p = np.where(valid, p0, n_points - 1)
        ray = p // spr
        ws = WStream(units)
        cur = mode.zero_frags(8 * mode.steps32)
        accL = None
        for w in prog.fwd_ops:
            code = w[0]
            if code == 1:
                k32, ng, nt = w[1] & 255, (w[1] >> 8) & 255, (w[1] >> 16) & 255
                act, flags = (w[1] >> 24) & 15, (w[1] >> 28) & 15
                aux = []
                for g in range(ng):
                    fr = make_group(mode, feat[w[3] + 64 * g: w[3] + 64 * g + 64], srcs, p, ray, valid)
                    aux.append(fr)
                    if training and w[6] >= 0:
                        stash.put_frags(w[6], blk, 2 * g, fr[:mode.steps32])
                        stash.put_frags(w[6], blk, 2 * g + 1, fr[mode.steps32:])
                nxt = mode.zero_frags(8 * mode.steps32)
                bits = np.zeros(64, dtype=np.uint64)
                for t in range(nt):
                    acc = bias_acc(bias, w[2], t)
                    acc = gemm_blocks(mode, acc, ws, cur, k32)
                    for g in range(ng):
                        acc = gemm_blocks(mode, acc, ws, aux[g], 2)
                    accL = acc.copy()
                    pos = acc > 0
                    if act == 1:
                        # device mask word: element i of tile (2d+q) is bit 31-(16q+i); set = gradient dropped.
                        # Padded points keep their (finite) activations: their dZ is zero throughout the backward.
                        for i in range(16):
                            bits |= ((~pos[:, i]).astype(np.uint64) << np.uint64(31 - (16 * (t & 1) + i)))
                        acc = np.where(pos, acc, 0.0)
                    nxt[t * mode.steps32:(t + 1) * mode.steps32] = acc_to_frags(mode, acc)
                    if (t & 1) or t == nt - 1:
                        if training and w[4] >= 0:
                            stash.put_mask(w[4], blk, t >> 1, bits)
                        bits = np.zeros(64, dtype=np.uint64)
                    if training and w[5] >= 0:
                        stash.put_frags(w[5], blk, t, nxt[t * mode.steps32:(t + 1) * mode.steps32])
                if not (flags & 1):
                    cur[:nt * mode.steps32] = nxt[:nt * mode.steps32]
            elif code == 4:
                for l in range(32):          # h == 0 lanes
                    if not valid[l]:
                        continue
                    for i in range(w[3]):
                        y = accL[l, i]
                        if w[4] == 1:
                            y = 1.0 / (1.0 + np.exp(-y))
                        if w[5] >= 0:
                            arr, per_ray = srcs[w[5]][0], srcs[w[5]][1]
                            y = y + arr[ray[l] if per_ray else p[l], w[6] + i]
                        outs[w[1]][p[l], w[2] + i] = y
                if w[7] > 0:                 # published components: same values, staged per block on the device
                    for i in range(w[3]):
                        c = int(COMPS[w[7] - 1 + i])
                        assert srcs[c >> 16][0] is outs[w[1]] and (c & 0xffff) == w[2] + i
            elif code == 5:
                n, nt = w[3], w[4]
                for l in range(64):
                    if not valid[l]:
                        continue
                    h = l >> 5
                    for t in range(nt):
                        if mode.bf16:
                            for s in range(2):
                                for j in range(8):
                                    row = 32 * t + 16 * s + pi16(h, j)
                                    if row < n:
                                        outs[w[1]][p[l], w[2] + row] = cur[2 * t + s, l, j]
                        else:
                            for q in range(16):
                                row = 32 * t + rho(q, h)
                                if row < n:
                                    outs[w[1]][p[l], w[2] + row] = cur[16 * t + q, l]
    return outs, stash


def run_backward(prog, mode, tables, params, srcs, n_points, spr, stash):
    global COMPS
    COMPS = tables["comps"]
    units = pack_units(mode, tables["bwd_units"], list(params) + [prog.selection_matrix().astype(np.float64)])
    feat = tables["feat"]
    dsrc = np.zeros((n_points, max(1, prog.n_dsrc)))
    nblk = (n_points + 31) // 32
    for blk in range(nblk):
        p0 = blk * 32 + R
        valid = p0 < n_points
        p = np.where(valid, p0, n_points - 1)
        ray = p // spr
        ws = WStream(units)
        cur = mode.zero_frags(8 * mode.steps32)
        cur2 = mode.zero_frags(mode.steps32)
        dacc = np.zeros((64, 16))      # source gradients of the block: row = dsrc column, col = point
        for w in prog.bwd_ops:
            code = w[0]
            if code == 1:
                n, to2 = w[3] & 255, (w[3] >> 8) & 1
                tmp = mode.zero_frags(mode.steps32)
                arr = srcs[w[1]][0] if (w[1] >= 0 and srcs[w[1]] is not None) else None
                from_dacc, q = (w[3] >> 9) & 1, (w[3] >> 10) & 3
                for l in range(32):
                    if not valid[l]:
                        continue
                    for i in range(n):
                        g = arr[p[l], w[2] + i] if arr is not None else 0.0
                        if from_dacc:            # rows 8q + i sit in registers 4q + i of the h == 0 lanes
                            assert rho(4 * q + i, 0) == 8 * q + i
                            g = g + dacc[l, 4 * q + i]
                        if w[4] == 1:
                            y = srcs[w[5]][0][p[l], w[6] + i]
                            g = g * y * (1.0 - y)
                        if mode.bf16:
                            tmp[0, l, i] = g
                        else:
                            tmp[i, l] = g
                if w[7] >= 0:
                    stash.put_frags(w[7], blk, 0, tmp)
                if to2:
                    cur2 = tmp
                else:
                    cur[:mode.steps32] = tmp
            elif code == 2:
                n, nt = w[3], w[4]
                arr, _ = srcs[w[1]]
                for t in range(nt):
                    v = np.zeros((64, 16))
                    bits = stash.get_mask(w[5], blk, t >> 1) if w[5] >= 0 else None
                    for l in range(64):
                        for i in range(16):
                            row = 32 * t + rho(i, l >> 5)
                            keep = True if bits is None else not ((int(bits[l]) >> (31 - (16 * (t & 1) + i))) & 1)
                            if valid[l] and keep and row < n:
                                v[l, i] = arr[p[l], w[2] + row]
                    cur[t * mode.steps32:(t + 1) * mode.steps32] = acc_to_frags(mode, v)
                    if w[7] >= 0:
                        stash.put_frags(w[7], blk, t, cur[t * mode.steps32:(t + 1) * mode.steps32])
            elif code == 3:
                k32, k32b, nt = w[1] & 255, (w[1] >> 8) & 255, (w[1] >> 16) & 255
                nxt = mode.zero_frags(8 * mode.steps32)
                for t in range(nt):
                    acc = np.zeros((64, 16))
                    acc = gemm_blocks(mode, acc, ws, cur, k32)
                    if k32b:
                        acc = gemm_blocks(mode, acc, ws, cur2, 1)
                    if w[4] >= 0:
                        bits = stash.get_mask(w[4], blk, t >> 1)
                        for l in range(64):
                            for i in range(16):
                                if (int(bits[l]) >> (31 - (16 * (t & 1) + i))) & 1:
                                    acc[l, i] = 0.0
                    nxt[t * mode.steps32:(t + 1) * mode.steps32] = acc_to_frags(mode, acc)
                    if w[5] >= 0:
                        stash.put_frags(w[5], blk, t, nxt[t * mode.steps32:(t + 1) * mode.steps32])
                cur[:nt * mode.steps32] = nxt[:nt * mode.steps32]
            elif code == 4:
                k32, k32b, ng = w[1] & 255, (w[1] >> 8) & 255, (w[1] >> 16) & 255
                for tt in range(2 * ng):
                    if not (w[2] >> tt) & 1:       # no feature of this tile has a gradient: not in the stream either
                        continue
                    acc = np.zeros((64, 16))
                    acc = gemm_blocks(mode, acc, ws, cur, k32)
                    if k32b:
                        acc = gemm_blocks(mode, acc, ws, cur2, 1)
                    ft = feat[w[3] + 32 * tt: w[3] + 32 * tt + 32]
                    G = np.zeros((64, 16))
                    for l in range(64):
                        for i in range(16):
                            e = ft[rho(i, l >> 5)]
                            # identity-only tiles (bit 8 + tt clear) take W^T dZ as it is
                            G[l, i] = acc[l, i] * (feature_grad(e, srcs, p[l], ray[l]) if (w[2] >> (8 + tt)) & 1 else 1.0)
                    # dacc[slot][point] += S[slot][feature] . G[feature][point]  (selection units of the stream)
                    sel = ws.take(mode.units32)
                    dacc = mma_block32(mode, dacc, sel, acc_to_frags(mode, G))
        for l in range(64):
            if valid[l]:
                for i in range(16):
                    slot = rho(i, l >> 5)
                    if slot < prog.n_dsrc:
                        dsrc[p[l], slot] = dacc[l, i]
    return dsrc


def run_wgrad(prog, mode, jobs, stash, n_grad):
    """hn_wgrad_kernel: per job, 4 waves on a gn x gk grid, each owning a tn x tk tile rectangle."""
    grads = np.zeros(n_grad)
    tb = 2048 if mode.bf16 else 4096
    for jb in jobs:
        gn, gk, bps = int(jb["pad"]) & 255, (int(jb["pad"]) >> 8) & 255, (int(jb["pad"]) >> 16) & 255
        assert gn * gk <= 8 and bps >= 1
        tn, tk = -(-jb["n_nt"] // gn), -(-jb["n_kt"] // gk)
        assert tn <= 4 and tk <= 2
        # KiB per LDS stage: what the build's ring holds (2 x 64 KiB since round 5; _lib mirrors the kernel's constants)
        assert bps * (jb["n_nt"] + jb["n_kt"]) * (2 if mode.bf16 else 4) <= L.WGRAD_MAX_STAGE_KB
        assert (jb["blk0"] % bps) == 0
        for wave in range(8):
            wn, wk = wave // gk, wave % gk
            n0, k0 = wn * tn, wk * tk
            my_n = min(tn, jb["n_nt"] - n0) if wn < gn else 0
            my_k = min(tk, jb["n_kt"] - k0)
            acc = [[np.zeros((64, 16)) for _ in range(4)] for _ in range(4)]
            accb = [np.zeros((64, 16)) for _ in range(4)]
            has_bias = jb["b_off"] >= 0 and my_n > 0     # tile i's bias goes to the wave with wk == i % gk
            for b in range(jb["blk0"], jb["blk1"]):
                za = [stash.get_tile(int(jb["z_off"]) + (b * jb["z_nt"] + jb["z_t0"] + n0 + i) * tb)
                      for i in range(max(my_n, 0))]
                def x_tile(kt):        # k-tiles 0 .. n_kt1-1 from the first X slot, the rest from the second
                    if kt < jb["n_kt1"]:
                        return stash.get_tile(int(jb["x_off"]) + (b * jb["x_nt"] + jb["x_t0"] + kt) * tb)
                    return stash.get_tile(int(jb["x2_off"]) + (b * jb["x2_nt"] + jb["x2_t0"] + kt - jb["n_kt1"]) * tb)
                xb = [x_tile(k0 + j) for j in range(max(my_k, 0))]
                for i in range(max(my_n, 0)):
                    for j in range(max(my_k, 0)):
                        if mode.bf16:
                            for v in range(2):
                                acc[i][j] = mfma_bf16(za[i][v], xb[j][v], acc[i][j])
                        else:
                            for g in range(4):
                                for e in range(4):
                                    acc[i][j] = mfma_f32(za[i][g][:, e], xb[j][g][:, e], acc[i][j])
                    if has_bias and (i % gk) == wk:
                        if mode.bf16:
                            for v in range(2):
                                accb[i] = mfma_bf16(za[i][v], np.ones((64, 8)), accb[i])
                        else:
                            for g in range(4):
                                for e in range(4):
                                    accb[i] = mfma_f32(za[i][g][:, e], np.ones(64), accb[i])
            for i in range(max(my_n, 0)):
                for j in range(max(my_k, 0)):
                    for l in range(64):
                        c, h = l & 31, l >> 5
                        for q in range(16):
                            row = jb["r0"] + 32 * (n0 + i) + rho(q, h)
                            col = jb["c0"] + 32 * (k0 + j) + c
                            if 0 <= row < jb["r_end"] and 0 <= col < jb["c_end"] and jb["w_off"] >= 0:
                                grads[jb["w_off"] + row * jb["ld"] + col] += acc[i][j][l, q]
                if has_bias and (i % gk) == wk:
                    for l in (0, 32):
                        for q in range(16):
                            row = jb["r0"] + 32 * (n0 + i) + rho(q, l >> 5)
                            if 0 <= row < jb["r_end"]:
                                grads[jb["b_off"] + row] += accb[i][l, q]
    return grads
