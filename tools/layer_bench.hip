// Stand-alone micro-benchmark (diagnostic, not part of the product): a chain of L plain 256 -> 256 ReLU layers of the
// forward MLP machine — register-resident transposed activations, weight stream through an LDS ring filled by
// LDS-DMA, ReLU masks and transposed stash written per tile — in several structural variants, to decide the shape
// of the bf16 machine kernels:
//   NB     32-point blocks per wave (1: 8 waves x 32 points, 2 waves/SIMD; 2: 4 waves x 64 points, 1 wave/SIMD)
//   PIPE   products of tile t interleaved slot by slot with the epilogue + stash of tile t-1
//   RING   chunks of 32 KiB in the LDS ring (RING-1 in flight)
//   CNT    counted vmcnt at the chunk barrier (stash stores issued since do not have to drain)
// Build: hipcc -O3 --offload-arch=gfx950 -o layer_bench layer_bench.hip ; run: ./layer_bench [points] [layers]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <string>
#include <utility>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#define DEV __device__ __forceinline__
// compile-time loop: the body is instantiated per index, so register arrays are always statically indexed (a
// `#pragma unroll` loop whose body holds the ring's barrier + switch is sometimes left rolled, and its arrays go to scratch)
template <int... I, class F>
DEV void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
DEV void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

DEV int rho(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }
DEV int pi16(int h, int j) { return 8 * (j >> 2) + 4 * h + (j & 3); }
// two floats -> one register of two bf16 (ONE v_cvt_pk_bf16_f32; element-wise (__bf16) casts cost a cvt + a merge each)
DEV unsigned pack2(float a, float b) {
  const bf16x2 v = __builtin_convertvector((f32x2){a, b}, bf16x2);
  return __builtin_bit_cast(unsigned, v);
}
DEV bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
DEV f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

struct Args {
  const char* wstream;   // L * 8 tiles * 16 units * 1 KiB
  const float* bias;     // L * 256
  char* stash;           // [L][block][8 tiles][2 KiB]
  uint32_t* masks;       // [L][block][4][64]
  __bf16* out;           // [points][256] (natural feature order)
  int n_points, n_layers, train;
};

DEV void wait_vmcnt(int n) {
  if (n >= 24) { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); return; }
  switch (n) {
#define C(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19)
    C(20) C(21) C(22) C(23)
#undef C
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// SPREAD: the DMA pieces of a chunk are not issued in a burst behind the chunk barrier (where all 8 waves stop issuing
//         MFMAs at the same time) but one at a time from `poll()` calls placed in the MFMA stream
// WSRC:   0 ring, 1 static chunk (no DMA, no barrier), 2 registers, 3 barriers without DMA, 4 DMA without barriers
template <int WAVES, int RING, bool CNT, int CHUNK, bool SPREAD, int WSRC, int LOADERS = 0>
struct Ring {
  const char* g;
  char* lds;
  int nchunks, wave, lane;
  int ctr;       // next unit
  int ops;       // vector-memory instructions this wave has issued so far
  int pend, pend_chunk;
  unsigned long long hist;  // `ops` (16 bits each) right after the DMA of the most recently issued chunks, latest lowest
  static constexpr int PIECES = LOADERS ? CHUNK / LOADERS : CHUNK / WAVES;
  DEV void piece(int c, int i) {
    if (WSRC == 3) return;
    if (LOADERS && wave < WAVES) return;          // compute waves issue no DMA
    const char* src = g + (size_t)c * (CHUNK * 1024);
    char* dst = lds + (c % RING) * (CHUNK * 1024);
    const int unit = LOADERS ? (wave - WAVES) + i * LOADERS : wave + i * WAVES;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + unit * 1024 + lane * 16),
                                     (__attribute__((address_space(3))) void*)(dst + unit * 1024), 16, 0, 0);
  }
  DEV void issue(int c) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) piece(c, i);
    ops += PIECES;
    hist = (hist << 16) | (unsigned)(ops & 0xffff);
  }
  DEV void poll() {
    if (!SPREAD || WSRC == 1 || WSRC == 2) return;
    if (pend > 0) {
      piece(pend_chunk, PIECES - pend);
      ops += 1;
      --pend;
      if (pend == 0) hist = (hist << 16) | (unsigned)(ops & 0xffff);
    }
  }
  DEV void start() {
    ctr = 0; ops = 0; hist = 0; pend = 0; pend_chunk = 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int c = 0; c < RING - 1 && c < nchunks; ++c) issue(c);
  }
  DEV const char* take(int n) {
    if (WSRC == 1 || WSRC == 2) {
      const char* p = lds + (ctr & (CHUNK - 1)) * 1024;
      ctr = (ctr + n) & (CHUNK - 1);
      return p;
    }
    if ((ctr & (CHUNK - 1)) + n > CHUNK) ctr = (ctr + CHUNK - 1) & ~(CHUNK - 1);
    if ((ctr & (CHUNK - 1)) == 0) {
      const int c = ctr / CHUNK;
      while (pend > 0) poll();
      // chunk c is the oldest in flight: everything issued up to its mark must have landed
      int younger = nchunks - 1 - c;            // chunks issued after chunk c
      if (younger > RING - 2) younger = RING - 2;
      const int mark = (int)((hist >> (16 * younger)) & 0xffff);
      if (LOADERS && wave < WAVES) { /* the loader waves wait for the DMA */ }
      else if (CNT) wait_vmcnt((ops - mark) & 0xffff);
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (WSRC != 4) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (c + RING - 1 < nchunks) {
        if (SPREAD) { pend = PIECES; pend_chunk = c + RING - 1; }
        else issue(c + RING - 1);
      }
    }
    const char* p = lds + (((ctr / CHUNK) % RING) * CHUNK + (ctr & (CHUNK - 1))) * 1024;
    ctr += n;
    return p;
  }
};

template <int WSRC>
DEV bf16x8 read_a(const char* p, int salt) {
  if constexpr (WSRC == 2) {
    u32x4 v = {0x3c003c00u + salt, 0x3c003c00u, 0xbc003c00u, 0x3c00bc00u};
    asm volatile("" : "+v"(v));
    return __builtin_bit_cast(bf16x8, v);
  } else {
    return *reinterpret_cast<const bf16x8*>(p);
  }
}
DEV void init_acc(f32x16& acc, const float* bias, int t, int h) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 32 * t + 8 * g + 4 * h);
    acc[4 * g] = b[0]; acc[4 * g + 1] = b[1]; acc[4 * g + 2] = b[2]; acc[4 * g + 3] = b[3];
  }
}

// one share of a finished tile's epilogue (see hn_mlp.hip hn_epilogue_share)
template <int EPI, int STORE>
DEV void epi_share(int k, int tp, f32x16& a, u32x4* frag, unsigned& bits, const f32x16& z, u32x4* zo, char* out_base,
                   uint32_t* mask_base, int lane, int& ops) {
  if (k < 8) {
    const int i = 2 * k;
    if (EPI >= 2) bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(a[i]), 31);
    if (EPI >= 2) bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(a[i + 1]), 31);
    if (EPI >= 1) frag[i >> 3][(i & 7) >> 1] = pack2(__int_as_float(max(__float_as_int(a[i]), 0)), __int_as_float(max(__float_as_int(a[i + 1]), 0)));
    else frag[i >> 3][(i & 7) >> 1] = pack2(a[i], a[i + 1]);
  } else if (EPI >= 3 && k < 12) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int i = 4 * (k - 8) + 2 * e;
      zo[i >> 3][(i & 7) >> 1] = pack2(z[i], z[i + 1]);
    }
    if (EPI >= 4 && STORE >= 1 && k == 10) {
      char* dst = out_base + (size_t)tp * 2048 + lane * 16;
      if (STORE == 1) __builtin_nontemporal_store(zo[0], reinterpret_cast<u32x4*>(dst));
      else *reinterpret_cast<u32x4*>(dst) = zo[0];
      ops += 1;
    }
  } else if (EPI >= 2 && k == 12) {
    char* dst = out_base + (size_t)tp * 2048 + lane * 16;
    if (EPI >= 4) {
      if (STORE == 0) {
        __builtin_nontemporal_store(zo[0], reinterpret_cast<u32x4*>(dst));
        __builtin_nontemporal_store(zo[1], reinterpret_cast<u32x4*>(dst + 1024));
        ops += 2;
      } else {
        if (STORE == 1) __builtin_nontemporal_store(zo[1], reinterpret_cast<u32x4*>(dst + 1024));
        else *reinterpret_cast<u32x4*>(dst + 1024) = zo[1];
        ops += 1;
      }
    } else if (EPI == 3) {
      asm volatile("" :: "v"(zo[0]), "v"(zo[1]));
    }
    if (tp & 1) {
      mask_base[(tp >> 1) * 64] = bits;
      bits = 0;
      ops += 1;
    }
  }
}

// WSRC: 0 = weight ring (LDS-DMA + chunk barriers), 1 = one static chunk in LDS (no DMA, no barriers), 2 = A fragments
//       from registers (no LDS reads at all)
// EPI:  0 = acc -> fragments by plain packing, 1 = + ReLU, 2 = + mask words (stored), 3 = + transposes (not stored),
//       4 = + stash stores (the full training epilogue)
template <int NB, int WAVES, bool PIPE, int RING, bool CNT, int WSRC, int EPI, int CHUNK, bool SPREAD, int STORE, int LOADERS>
__global__ __launch_bounds__((WAVES + LOADERS) * 64, LOADERS ? 3 : (NB == 2 ? 1 : 2)) void layer_chain(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  Ring<WAVES, RING, CNT, CHUNK, SPREAD, WSRC, LOADERS> ws;
  ws.g = a.wstream; ws.lds = smem; ws.nchunks = a.n_layers * 128 / CHUNK; ws.wave = wave; ws.lane = lane;
  float* bias_lds = reinterpret_cast<float*>(smem + RING * CHUNK * 1024);
  for (int i = threadIdx.x; i < a.n_layers * 256; i += blockDim.x) bias_lds[i] = a.bias[i];
  __syncthreads();

  if (LOADERS && wave >= WAVES) {       // dedicated loader waves: all DMA, all vmcnt waits, no arithmetic
    ws.start();
    for (int c = 0; c < ws.nchunks; ++c) ws.take(CHUNK);
    return;
  }
  const int nblk_total = (a.n_points + 31) / 32;
  const int blk0 = (blockIdx.x * WAVES + wave) * NB;
  u32x4 cur[NB][16], nxt[NB][16];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        // input: +-[0.5, 1) from a hash of (point, feature pair): two bf16 per word
        const unsigned p = (blk0 + b) * 32 + r, f = 16 * u + pi16(h, 2 * jj);
        unsigned x = p * 2654435761u ^ (f * 40503u + 12345u);
        x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
        cur[b][u][jj] = (x & 0x807f807fu) | 0x3f003f00u;
      }
  bf16x8 id[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int j = 0; j < 8; ++j) id[u][j] = (r == 16 * u + pi16(h, j)) ? (__bf16)1.0f : (__bf16)0.0f;

  if (WSRC == 0 || WSRC == 4) ws.start();
  else { ws.ctr = 0; ws.ops = 0; ws.hist = 0; ws.pend = 0; ws.pend_chunk = 0;
         for (int i = 0; i < CHUNK / WAVES; ++i) { const int unit = wave + i * WAVES;
           __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.wstream + unit * 1024 + lane * 16),
                                            (__attribute__((address_space(3))) void*)(smem + unit * 1024), 16, 0, 0); }
         asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
  constexpr bool train = EPI >= 2;
  for (int l = 0; l < a.n_layers; ++l) {
    const float* bias = bias_lds + l * 256;
    char* out_base[NB];
    uint32_t* mask_base[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      out_base[b] = a.stash + ((size_t)l * nblk_total + blk0 + b) * (8 * 2048);
      mask_base[b] = a.masks + ((size_t)l * nblk_total + blk0 + b) * 256 + lane;
    }
    if constexpr (PIPE) {
      constexpr int SH = 13 * NB;             // epilogue shares per tile
      constexpr int PER = (SH + 15) / 16;     // shares per slot
      f32x16 acc[2][NB];
      f32x16 z[NB];
      u32x4 zo[NB][2];
      unsigned bits[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) bits[b] = 0;
      static_for<9>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        const char* wl = nullptr;
        if (t < 8) wl = ws.take(16) + lane * 16;
        const int tp = t - 1;
        bf16x8 q[2];
        if (t < 8) {
#pragma unroll
          for (int b = 0; b < NB; ++b) init_acc(acc[t & 1][b], bias, t, h);
          q[0] = read_a<WSRC>(wl, 0);
          q[1] = read_a<WSRC>(wl + 1024, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (t < 8) {
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[t & 1][b] = mfma(q[u & 1], as_frag(cur[b][u]), acc[t & 1][b]);
            if (u + 2 < 16) q[u & 1] = read_a<WSRC>(wl + (u + 2) * 1024, u);
          }
          if (t > 0) {
#pragma unroll
            for (int e = 0; e < PER; ++e) {
              const int s = u * PER + e;
              if (s < SH) {
                const int b = s % NB, k = s / NB;
                if (EPI >= 3 && k == 4) z[b] = mfma(as_frag(nxt[b][tp * 2]), id[0], f32x16{0});
                if (EPI >= 3 && k == 8) z[b] = mfma(as_frag(nxt[b][tp * 2 + 1]), id[1], z[b]);
                epi_share<EPI, STORE>(k, tp, acc[tp & 1][b], &nxt[b][tp * 2], bits[b], z[b], zo[b], out_base[b], mask_base[b], lane, ws.ops);
              }
            }
          }
          if (t < 8 && (u == 13 || u == 15)) ws.poll();
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    } else {
      unsigned bits[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) bits[b] = 0;
      static_for<8>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        const char* wl = ws.take(16) + lane * 16;
        f32x16 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) init_acc(acc[b], bias, t, h);
        bf16x8 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = read_a<WSRC>(wl + u * 1024, u);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] = mfma(q[u % 4], as_frag(cur[b][u]), acc[b]);
          if (u + 4 < 16) q[u % 4] = read_a<WSRC>(wl + (u + 4) * 1024, u);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            if (train) bits[b] = __builtin_amdgcn_alignbit(bits[b], __float_as_uint(acc[b][i]), 31);
            if (train) bits[b] = __builtin_amdgcn_alignbit(bits[b], __float_as_uint(acc[b][i + 1]), 31);
            if (EPI >= 1) nxt[b][t * 2 + (i >> 3)][(i & 7) >> 1] = pack2(__int_as_float(max(__float_as_int(acc[b][i]), 0)),
                                                           __int_as_float(max(__float_as_int(acc[b][i + 1]), 0)));
            else nxt[b][t * 2 + (i >> 3)][(i & 7) >> 1] = pack2(acc[b][i], acc[b][i + 1]);
          }
          if (train) {
            if (t & 1) { mask_base[b][(t >> 1) * 64] = bits[b]; bits[b] = 0; ws.ops += 1; }
          }
          if (EPI >= 3) {
            f32x16 z = mfma(as_frag(nxt[b][t * 2]), id[0], f32x16{0});
            z = mfma(as_frag(nxt[b][t * 2 + 1]), id[1], z);
            char* dst = out_base[b] + (size_t)t * 2048 + lane * 16;
#pragma unroll
            for (int v = 0; v < 2; ++v) {
              u32x4 o;
#pragma unroll
              for (int j = 0; j < 4; ++j) o[j] = pack2(z[8 * v + 2 * j], z[8 * v + 2 * j + 1]);
              if (EPI >= 4) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(dst + v * 1024));
              else asm volatile("" :: "v"(o));
            }
            if (EPI >= 4) ws.ops += 2;
          }
        }
      });
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int u = 0; u < 16; ++u) cur[b][u] = nxt[b][u];
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int p = (blk0 + b) * 32 + r;
    if (p < a.n_points)
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) a.out[(size_t)p * 256 + 16 * u + pi16(h, j)] = as_frag(cur[b][u])[j];
  }
}


// ---------------------------------------------------------------------------------------------------------------
// WS: weights stationary in REGISTERS, activations through an LDS image (design 1).
// 8 waves, 256 points per workgroup.  Wave w owns the 32-feature output slab w of the layer: its 16 B-operand
// fragments (64 VGPRs) come straight from L2 (prefetched one layer ahead), the A operand (32 points x 16 features)
// is read from the image [feature][point] with two ds_read_b64_tr_b16 per MFMA.  D = X . W^T has points on the
// register axis and features on the lane axis, i.e. the layout of the transposed stash: no transposing MFMAs.
// The slab's outputs stay packed in registers until every wave has finished reading the image (barrier), then
// overwrite the image in place (4 ds_write_b64 per tile), barrier, next layer.
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;
DEV int img_off(int f, int pt) {       // two half images of [256 feature rows][128 points], 16-byte chunks XOR-swizzled
  const int half = pt >> 7, q = pt & 127;
  const int sw = ((f & 3) << 2) | ((f >> 2) & 3);
  return half * 65536 + 256 * f + 16 * ((q >> 3) ^ sw) + 2 * (q & 7);
}
struct WsArgs {
  const char* wstream;   // [L][8 slabs][16 k-steps][64 lanes][8 bf16]
  const float* bias;
  char* stash;
  uint32_t* masks;
  __bf16* out;
  int n_points, n_layers;
};
template <bool TRAIN>
__global__ __launch_bounds__(512, 2) void ws_chain(const WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, h = lane >> 5;
  float* bias_lds = reinterpret_cast<float*>(smem + 131072);
  for (int i = threadIdx.x; i < a.n_layers * 256; i += blockDim.x) bias_lds[i] = a.bias[i];
  const int p_base = blockIdx.x * 256;
  for (int idx = threadIdx.x; idx < 256 * 128; idx += 512) {      // image of the input: the same hash as the other variants
    const int pl = idx >> 7, fp = idx & 127;
    const unsigned p = p_base + pl, f = 2 * fp;
    unsigned x = p * 2654435761u ^ (f * 40503u + 12345u);
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    x = (x & 0x807f807fu) | 0x3f003f00u;
    *reinterpret_cast<unsigned short*>(smem + img_off(f, pl)) = (unsigned short)(x & 0xffff);
    *reinterpret_cast<unsigned short*>(smem + img_off(f + 1, pl)) = (unsigned short)(x >> 16);
  }
  __syncthreads();
  // per-lane byte offsets of the transposed reads: lane 4q+p of 16-lane group gi -> row (feature) f0+q, points 4p..4p+3
  const int gi = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int r0 = 16 * (gi & 1), hh = gi >> 1;
  const int nblk_total = (a.n_points + 31) / 32;
  u32x4 Wa[16];
  auto load_w = [&](u32x4* W, int l) {
    const char* src = a.wstream + (((size_t)l * 8 + wave) * 16) * 1024 + lane * 16;
#pragma unroll
    for (int u = 0; u < 16; ++u) W[u] = *reinterpret_cast<const u32x4*>(src + u * 1024);
  };
  load_w(Wa, 0);
  // transposed-read addresses: lane-dependent part per (feature sub-block, row tile & 3); k-step and half image are
  // immediates / scalar adds
  int abase[2][4];
#pragma unroll
  for (int part = 0; part < 2; ++part)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) abase[part][r4] = img_off(8 * hh + 4 * part + qq, r4 * 32 + r0 + 4 * pp);
  auto layer = [&](int l, u32x4* W) __attribute__((always_inline)) {
    const float bias = bias_lds[l * 256 + 32 * wave + c];
    unsigned out[8][8];
    unsigned bits = 0;
    static_for<8>([&](auto RT) __attribute__((always_inline)) {
      constexpr int rt = decltype(RT)::value;
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = bias;
      auto read_a = [&](int u) {
        // features 16u + 8hh + 4*part + qq ; points rt*32 + r0 + 4pp
        // img_off(16u + f', pt) = img_off(f', pt & 127) + 4096 u + 65536 (pt >> 7): the swizzle only sees f' & 15
        const int cst = 4096 * u + 65536 * (rt >> 2);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(smem + abase[0][rt & 3] + cst));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(smem + abase[1][rt & 3] + cst));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
      };
      bf16x8 q[2];
      q[0] = read_a(0); q[1] = read_a(1);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        acc = mfma(q[u & 1], as_frag(W[u]), acc);
        if (u + 2 < 16) q[u & 1] = read_a(u + 2);
      }
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        if (TRAIN) bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(acc[i]), 31);
        if (TRAIN) bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(acc[i + 1]), 31);
        out[rt][i >> 1] = pack2(__int_as_float(max(__float_as_int(acc[i]), 0)), __int_as_float(max(__float_as_int(acc[i + 1]), 0)));
      }
      if (TRAIN) {
        const int blk = blockIdx.x * 8 + rt;
        char* dst = a.stash + (((size_t)l * nblk_total + blk) * 8 + wave) * 2048 + lane * 16;
        __builtin_nontemporal_store((u32x4){out[rt][0], out[rt][1], out[rt][2], out[rt][3]}, reinterpret_cast<u32x4*>(dst));
        __builtin_nontemporal_store((u32x4){out[rt][4], out[rt][5], out[rt][6], out[rt][7]}, reinterpret_cast<u32x4*>(dst + 1024));
        if (rt & 1) {
          a.masks[(((size_t)l * (nblk_total / 2) + blockIdx.x * 4 + (rt >> 1)) * 8 + wave) * 64 + lane] = bits;
          bits = 0;
        }
      }
    });
    if (l + 1 < a.n_layers) load_w(W, l + 1);      // lands behind the barriers and the image writes
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // every wave has read what it needs of the image
    static_for<8>([&](auto RT) __attribute__((always_inline)) {
      constexpr int rt = decltype(RT)::value;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        *reinterpret_cast<u32x2*>(smem + img_off(32 * wave + c, rt * 32 + 8 * g + 4 * h)) = (u32x2){out[rt][2 * g], out[rt][2 * g + 1]};
      }
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // the next layer's image is complete
  };
  for (int l = 0; l < a.n_layers; ++l) layer(l, Wa);
  for (int idx = threadIdx.x; idx < 256 * 256; idx += 512) {
    const int pl = idx >> 8, f = idx & 255;
    if (p_base + pl < a.n_points)
      a.out[(size_t)(p_base + pl) * 256 + f] = __builtin_bit_cast(__bf16, *reinterpret_cast<unsigned short*>(smem + img_off(f, pl)));
  }
}
// ---------------------------------------------------------------------------------------------------------------
static float bf16_round(float x) {
  uint32_t u; memcpy(&u, &x, 4);
  u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000u;
  float y; memcpy(&y, &u, 4); return y;
}
static uint16_t bf16_bits(float x) { float y = bf16_round(x); uint32_t u; memcpy(&u, &y, 4); return (uint16_t)(u >> 16); }
static float bf16_to_f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float y; memcpy(&y, &u, 4); return y; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NB, int WAVES, bool PIPE, int RING, bool CNT, int WSRC, int EPI, int CHUNK = 32, bool SPREAD = false, int STORE = 0, int LOADERS = 0>
static double run(const char* name, Args a, int reps, std::vector<uint16_t>* out_host) {
  auto k = layer_chain<NB, WAVES, PIPE, RING, CNT, WSRC, EPI, CHUNK, SPREAD, STORE, LOADERS>;
  a.train = EPI;
  const size_t lds = (size_t)RING * CHUNK * 1024 + a.n_layers * 256 * 4;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int pts_wg = WAVES * NB * 32;
  const int grid = (a.n_points + pts_wg - 1) / pts_wg;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3((WAVES + LOADERS) * 64), lds, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3((WAVES + LOADERS) * 64), lds, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double flops = 2.0 * a.n_points * 256.0 * 256.0 * a.n_layers;
  printf("%-34s train=%d  %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 2.5 PF)\n", name, a.train, ms, flops / ms * 1e-9,
         flops / ms * 1e-9 / 2500.0 * 100.0);
  if (out_host) {
    out_host->resize((size_t)a.n_points * 256);
    CK(hipMemcpy(out_host->data(), a.out, out_host->size() * 2, hipMemcpyDeviceToHost));
  }
  return ms;
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 131072;
  const int L = argc > 2 ? atoi(argv[2]) : 8;
  const int reps = argc > 3 ? atoi(argv[3]) : 20;
  // weights: He-uniform so that activations keep their scale through the ReLU layers
  std::vector<float> W((size_t)L * 256 * 256), B((size_t)L * 256);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  const float lim = sqrtf(6.0f / 256.0f);
  for (auto& w : W) w = bf16_round(rnd() * lim);
  for (auto& b : B) b = rnd() * 0.05f;
  std::vector<uint16_t> packed((size_t)L * 8 * 16 * 512);
  for (int l = 0; l < L; ++l)
    for (int t = 0; t < 8; ++t)
      for (int u = 0; u < 16; ++u)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int r = lane & 31, h = lane >> 5;
            const int k = 16 * u + 8 * (j >> 2) + 4 * h + (j & 3);
            packed[((((size_t)l * 8 + t) * 16 + u) * 64 + lane) * 8 + j] = bf16_bits(W[((size_t)l * 256 + 32 * t + r) * 256 + k]);
          }
  std::vector<uint16_t> packed_ws((size_t)L * 8 * 16 * 512);
  for (int l = 0; l < L; ++l)
    for (int sl = 0; sl < 8; ++sl)
      for (int u = 0; u < 16; ++u)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j)
            packed_ws[((((size_t)l * 8 + sl) * 16 + u) * 64 + lane) * 8 + j] =
                bf16_bits(W[((size_t)l * 256 + 32 * sl + (lane & 31)) * 256 + 16 * u + 8 * (lane >> 5) + j]);
  char* d_wws;
  CK(hipMalloc(&d_wws, packed_ws.size() * 2)); CK(hipMemcpy(d_wws, packed_ws.data(), packed_ws.size() * 2, hipMemcpyHostToDevice));
  Args a{};
  const int nblk = (P + 31) / 32;
  char* d_w; float* d_b; char* d_s; uint32_t* d_m; __bf16* d_o;
  CK(hipMalloc(&d_w, packed.size() * 2)); CK(hipMemcpy(d_w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_b, B.size() * 4)); CK(hipMemcpy(d_b, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_s, (size_t)L * nblk * 8 * 2048 + (1 << 20)));
  CK(hipMalloc(&d_m, (size_t)L * nblk * 256 * 4 + (1 << 20)));
  CK(hipMalloc(&d_o, (size_t)(P + 512) * 256 * 2));
  a.wstream = d_w; a.bias = d_b; a.stash = d_s; a.masks = d_m; a.out = d_o; a.n_points = P; a.n_layers = L;

  // reference for the first 8 points
  std::vector<float> ref(8 * 256);
  for (int p = 0; p < 8; ++p) {
    std::vector<float> x(256), y(256);
    for (int f = 0; f < 256; f += 2) {
      unsigned v = (unsigned)p * 2654435761u ^ ((unsigned)f * 40503u + 12345u);
      v ^= v >> 13; v *= 0x5bd1e995u; v ^= v >> 15;
      v = (v & 0x807f807fu) | 0x3f003f00u;
      x[f] = bf16_to_f((uint16_t)(v & 0xffff));
      x[f + 1] = bf16_to_f((uint16_t)(v >> 16));
    }
    for (int l = 0; l < L; ++l) {
      for (int n = 0; n < 256; ++n) {
        float acc = B[(size_t)l * 256 + n];
        for (int k = 0; k < 256; ++k) acc += W[((size_t)l * 256 + n) * 256 + k] * x[k];
        y[n] = bf16_round(acc > 0 ? acc : 0);
      }
      x = y;
    }
    for (int f = 0; f < 256; ++f) ref[p * 256 + f] = x[f];
  }
  auto check = [&](const std::vector<uint16_t>& o, const char* nm) {
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < 8 * 256; ++i) {
      maxerr = fmax(maxerr, fabs(bf16_to_f(o[i]) - ref[i]));
      maxref = fmax(maxref, fabs(ref[i]));
    }
    printf("   check %-28s max|err| = %.4g (max|ref| = %.4g) %s\n", nm, maxerr, maxref, maxerr <= 0.03 * maxref ? "ok" : "MISMATCH");
  };
  auto run_ws = [&](bool train, std::vector<uint16_t>* oh) {
    WsArgs w{d_wws, d_b, d_s, d_m, d_o, P, L};
    auto k = train ? ws_chain<true> : ws_chain<false>;
    const size_t lds = 131072 + (size_t)L * 256 * 4;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int grid = (P + 255) / 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, w);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, w);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double flops = 2.0 * P * 256.0 * 256.0 * L;
    printf("%-34s train=%d  %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 2.5 PF)\n", "WS weights-in-registers", (int)train, ms,
           flops / ms * 1e-9, flops / ms * 1e-9 / 2500.0 * 100.0);
    if (oh) { oh->resize((size_t)P * 256); CK(hipMemcpy(oh->data(), d_o, oh->size() * 2, hipMemcpyDeviceToHost)); }
  };
  std::vector<uint16_t> o0, o1;
  auto same = [&](const char* nm) { size_t diff = 0; for (size_t i = 0; i < o0.size(); ++i) diff += o0[i] != o1[i]; printf("   %s vs base: %zu differing outputs\n", nm, diff); };
  run<1, 8, false, 2, false, 0, 4>("base  ring2 vmcnt0 full", a, reps, &o0); check(o0, "base");
  run_ws(true, &o1); check(o1, "WS");
  { double md = 0; for (size_t i = 0; i < o0.size(); ++i) md = fmax(md, fabs(bf16_to_f(o0[i]) - bf16_to_f(o1[i]))); printf("   WS vs base: max |diff| over all outputs = %.4g\n", md); }
  for (int rep = 0; rep < 4; ++rep) {
    run<1, 8, false, 2, false, 0, 4>("base  ring2 vmcnt0 full", a, reps, nullptr);
    run<1, 8, true, 3, true, 0, 4>("pipe  ring3 counted full", a, reps, nullptr);
    run_ws(true, nullptr);
    run<1, 8, false, 2, false, 0, 1>("base  ring2 vmcnt0 epi1", a, reps, nullptr);
    run<1, 8, true, 3, true, 0, 1>("pipe  ring3 counted epi1", a, reps, nullptr);
    run_ws(false, nullptr);
  }
  return 0;
}
