// MLP machine for gfx950: weight packing, fused forward, fused backward-data, weight gradient.
// See include/hn_kernels.h for the program format and hn_common.h for the register layouts.
#include "hn_common.h"
#include <algorithm>
#include <type_traits>

// Diagnostic build only (-DHN_PROF): wave 0 of workgroup 0 logs (code, shader clock) pairs into HnMlpArgs.prof.
#ifdef HN_PROF
#define HN_STAMP(code)                                                                          \
  do {                                                                                          \
    if (prof_on && prof_n < 2040) {                                                             \
      unsigned long long t__;                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                        \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");               \
      __builtin_amdgcn_sched_barrier(0);                                                        \
      if (lane == 0) { prof_buf[2 * prof_n] = (code); prof_buf[2 * prof_n + 1] = (long long)t__; } \
      ++prof_n;                                                                                 \
    }                                                                                           \
  } while (0)
#else
#define HN_STAMP(code) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------
// weight stream through LDS (2 x 32 KiB chunks, filled by LDS-DMA, one barrier per chunk)
// ------------------------------------------------------------------------------------------------
template <int WAVES>
struct WStream {
  const char* g;  // packed units in global memory
  char* lds;      // 2 * 32 KiB
  int ctr;        // next unit
  int nchunks;
  int wave, lane;

  HN_DEV void issue(int c) {
    const char* src = g + (size_t)c * (HN_CHUNK_UNITS * 1024);
    char* dst = lds + (c & 1) * (HN_CHUNK_UNITS * 1024);
#ifndef HN_WSTREAM_ASYM
#define HN_WSTREAM_ASYM 1      /* 0: rounds 1-5a, every wave issues its eighth of a chunk's DMA (A/B knob) */
#endif
    if constexpr (WAVES == 8 && HN_WSTREAM_ASYM != 0) {
      // Waves w and w + 4 share a SIMD.  A wave sits ~100-200 cycles on every 1-KiB LDS-DMA instruction it issues (the
      // memory pipe's back-pressure, measured in hn_wgrad_kernel); with all eight waves issuing four pieces each behind
      // the chunk barrier, every SIMD's matrix pipe idled that long once per chunk.  Now the SECOND wave of every SIMD
      // issues the whole chunk (eight pieces) and the first goes from the barrier straight into its products: forward
      // -1.5 %, backward -1.3 % same box (profiles/r05_wgrad_ring.log: waves 0-3 instead -0.3 / -1.6 %, two waves x 16
      // pieces -2.0 / -1.2 %, one wave x 32 +8 / +4 %).
      if (wave >= 4) {
        // the first unit is made opaque here: otherwise the eight piece addresses of chunk 0 are hoisted out of the
        // persistent tile loop and held in 16 registers for the whole kernel (the AUXG = 3 training build spilled)
        int u0 = wave - 4;
        asm volatile("" : "+s"(u0));
#pragma unroll
        for (int i = 0; i < HN_CHUNK_UNITS / 4; ++i) {
          const int unit = u0 + i * 4;
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(src + unit * 1024 + lane * 16),
              (__attribute__((address_space(3))) void*)(dst + unit * 1024), 16, 0, 0);
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < HN_CHUNK_UNITS / WAVES; ++i) {
      const int unit = wave + i * WAVES;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + unit * 1024 + lane * 16),
          (__attribute__((address_space(3))) void*)(dst + unit * 1024), 16, 0, 0);
    }
  }
  HN_DEV void start() {
    ctr = 0;
    __syncthreads();  // previous tile's readers are done with both buffers
    issue(0);
  }
  // returns the LDS address of `n` consecutive units (never straddles a chunk; host packs alike)
  HN_DEV const char* take(int n) {
    if ((ctr & (HN_CHUNK_UNITS - 1)) + n > HN_CHUNK_UNITS) ctr = (ctr + HN_CHUNK_UNITS - 1) & ~(HN_CHUNK_UNITS - 1);
    if ((ctr & (HN_CHUNK_UNITS - 1)) == 0) {
      const int c = ctr / HN_CHUNK_UNITS;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // chunk c landed for every wave; everyone is done with chunk c-1
      if (c + 1 < nchunks) issue(c + 1);
    }
    const char* p = lds + (((ctr / HN_CHUNK_UNITS) & 1) * HN_CHUNK_UNITS + (ctr & (HN_CHUNK_UNITS - 1))) * 1024;
    ctr += n;
    return p;
  }
};

// ------------------------------------------------------------------------------------------------
// generated input features
// ------------------------------------------------------------------------------------------------
// Per 32-point block the source components a program reads (HnMlpArgs.comps) are staged once into LDS
// (srcv[ci][point]); a feature then costs two LDS reads (table entry, value) instead of a chain of global loads.
// A source without a pointer is skipped: its components are published by the program itself (HN_OP_OUT w7).  A per-ray
// source with a gather index reads row gather_idx[ray] (the GLO lookup); an index outside the table stages NaN.
// x / 2pi as an unevaluated sum hi + lo (the product x * C_HI exactly, plus the tail of 1/2pi): what lets the bf16
// forward take sin(2^k x) at full fp32 accuracy — 2^k * hi is exact, so is its fract (see hn_features4)
#define HN_INV2PI_HI 0.15915494309189535f
#define HN_INV2PI_LO ((float)(0.15915494309189535 - (double)0.15915494309189535f))
HN_DEV void hn_rev_split(float x, float& hi, float& lo) {
  hi = __fmul_rn(x, HN_INV2PI_HI);
  lo = __builtin_fmaf(x, HN_INV2PI_LO, __builtin_fmaf(x, HN_INV2PI_HI, -hi));
}
// `rev` (bf16 forward only, else nullptr): planes [hi | lo] of n_trig x 32 floats behind the wave's value plane, for
// the first n_trig staged components (the host orders components so that every encoded one comes first).
// `with_lo` false (a program whose planes would not fit into LDS otherwise): only hi is staged, the features read a
// shared all-zero plane as lo — the accuracy of the one-FMA form (hi alone carries x / 2pi to 2^-24 relative).
HN_DEV void hn_stage_sources(float* srcv, const HnMlpArgs& a, int p, int ray, int lane, float* rev = nullptr,
                             int n_trig = 0, bool with_lo = true) {
  const int r = lane & 31, h = lane >> 5;
  for (int ci = h; ci < a.n_comps; ci += 2) {
    const int c = a.comps[ci];
    const int sid = c >> 16, col = c & 0xffff;
    HnSrc s = a.src[0];
    if (sid == 1) s = a.src[1];
    if (sid == 2) s = a.src[2];
    if (sid == 3) s = a.src[3];
    float x = 0.0f;      // a source without a pointer: published later by the program itself; until then its planes
                         // read as 0 (padding features point at component 0, and 0 * stale-NaN would poison a tile)
    if (s.ptr != nullptr) {
      long long row = s.per_ray ? ray : p;
      bool ok = true;
      if (s.gather_idx != nullptr) {
        row = s.gather_idx[ray];
        ok = row >= 0 && row < s.gather_rows;
      }
      x = ok ? s.ptr[(size_t)row * s.ld + col] : __builtin_nanf("");
    }
    srcv[ci * 32 + r] = x;
    if (rev != nullptr && ci < n_trig) {
      float hi, lo;
      hn_rev_split(x, hi, lo);
      rev[ci * 32 + r] = hi;
      if (with_lo) rev[(n_trig + ci) * 32 + r] = lo;
    }
  }
  if (rev != nullptr && !with_lo && h == 0) rev[n_trig * 32 + r] = 0.0f;      // the shared zero plane
}

// (Never a gathered source: machine.Program refuses to leave identity features of a source in `no_direct` unstaged.
// Handling the gather here as well makes hipcc address the kernel arguments dynamically and copy all 2.5 KB of them
// to scratch, in every forward kernel.)
HN_DEV float hn_direct_source(const HnFeat e, const HnMlpArgs& a, int p, int ray) {
  const int sid = (e.packed >> 8) & 15, col = (e.packed >> 24) & 255;
  const float* base = a.src[0].ptr;
  int ld = a.src[0].ld, pr = a.src[0].per_ray;
  if (sid == 1) { base = a.src[1].ptr; ld = a.src[1].ld; pr = a.src[1].per_ray; }
  if (sid == 2) { base = a.src[2].ptr; ld = a.src[2].ld; pr = a.src[2].per_ray; }
  if (sid == 3) { base = a.src[3].ptr; ld = a.src[3].ld; pr = a.src[3].per_ray; }
  return base[(size_t)(pr ? ray : p) * ld + col];
}

// sin(2*pi*t): one v_fract + one v_sin, any magnitude of t
HN_DEV float hn_sin_rev(float t) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(t)); }
#define HN_INV_2PI 0.15915494309189535f

// Derived feature tables of the bf16 kernels (built in LDS at kernel start from HnFeat): everything a feature needs
// without decoding its kind.
// Backward (hn_derive_feat, derivative only, argument accuracy uncritical): x = *(srcv + off),
//   d value / dx = idmask ? 1 : 2pi scale * sin_rev(scale * x + phase + 1/4), scale = f / 2pi, phase 0 | 1/4.
// Forward (hn_derive_feat_fwd): value = idmask ? x : sin_rev(fract(f * hi) + (f * lo + phase)) with (hi, lo) = x / 2pi
//   from the staging planes: `off` = byte offset of x | byte offset of hi << 16 (lo sits n_trig planes further), scale =
//   f itself.  For the power-of-two frequencies of posenc_orig f * hi and its fract are EXACT, the remaining sum is
//   below 1.3: the argument of the hardware sine is good to ~1e-7 revolutions at every octave, where the one-FMA form
//   (scale * x + phase, |t| up to 82 revolutions at f = 512) is good to 8e-6.  That error, rounded to bf16 and carried
//   through the ReLUs, was 2/3 of the bf16 mode's gradient error against the bf16-operand oracle (4.8e-2 -> 1.6e-2
//   relative L2 with precise sines).  SIN: phase 0 ; COS / SINP: 1/4 ; ZERO: scale = phase = 0.
struct HnDFeat { unsigned off; float scale, phase; unsigned idmask; };
HN_DEV HnDFeat hn_derive_feat(const HnFeat e) {
  const int kind = (e.packed >> 12) & 15;
  HnDFeat d;
  d.off = (unsigned)(e.packed & 255) * 128u;
  const bool trig = kind == HN_FEAT_SIN || kind == HN_FEAT_COS || kind == HN_FEAT_SINP;
  d.scale = trig ? e.freq * 0.15915494309189535f : 0.0f;
  d.phase = (kind == HN_FEAT_COS || kind == HN_FEAT_SINP) ? 0.25f : 0.0f;
  d.idmask = (kind == HN_FEAT_ID || kind == HN_FEAT_ID_DIRECT) ? 0xffffffffu : 0u;
  return d;
}
HN_DEV HnDFeat hn_derive_feat_fwd(const HnFeat e, int n_comps, int n_trig, bool with_lo) {
  const int kind = (e.packed >> 12) & 15, ci = e.packed & 255;
  HnDFeat d;
  const bool trig = (kind == HN_FEAT_SIN || kind == HN_FEAT_COS || kind == HN_FEAT_SINP) && ci < n_trig;
  // two staged values per feature: (A, B) = (hi, lo) of x / 2pi for a trigonometric feature, (x, x) for the others
  const unsigned oa = trig ? (unsigned)(n_comps + ci) * 128u : (unsigned)ci * 128u;
  const unsigned ob = trig ? (unsigned)(n_comps + n_trig + (with_lo ? ci : 0)) * 128u : oa;
  d.off = oa | ob << 16;
  d.scale = trig ? e.freq : 0.0f;
  d.phase = (kind == HN_FEAT_COS || kind == HN_FEAT_SINP) ? 0.25f : 0.0f;
  d.idmask = (kind == HN_FEAT_ID || kind == HN_FEAT_ID_DIRECT) ? 0xffffffffu : 0u;
  return d;
}
HN_DEV float hn_bfi(unsigned m, float a, float b) {     // m ? a : b, bit-wise (one v_bfi_b32)
  return __uint_as_float((m & __float_as_uint(a)) | (~m & __float_as_uint(b)));
}
// values of 4 consecutive forward-table entries for the lane's point: 4 x 16-byte table reads issued together, then
// the 8 staged values (A, B; addresses from the entries) together, then 6 VALU per feature:
//   value = idmask ? A : sin(fract(f * A) + (f * B + phase))        [revolutions; the sum stays below 1.3, inside the
// hardware sine's domain, so no second fract].  `srcv_r` = the wave's staged components + 4 * point-in-block.
HN_DEV void hn_features4(const HnDFeat* tp, const char* srcv_r, float* out) {
  u32x4 t[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) t[j] = reinterpret_cast<const u32x4*>(tp)[j];
  __builtin_amdgcn_sched_barrier(0);
  float va[4], vb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    va[j] = *reinterpret_cast<const float*>(srcv_r + (t[j][0] & 0xffffu));
    vb[j] = *reinterpret_cast<const float*>(srcv_r + (t[j][0] >> 16));
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float f = __uint_as_float(t[j][1]);
    const float big = __builtin_amdgcn_fractf(__fmul_rn(f, va[j]));                      // exact for f = 2^k
    const float sv = __builtin_amdgcn_sinf(__fadd_rn(big, __builtin_fmaf(f, vb[j], __uint_as_float(t[j][2]))));
    out[j] = hn_bfi(t[j][3], va[j], sv);
  }
}
HN_DEV void hn_feature_grads4(const HnDFeat* tp, const char* srcv_r, float* out) {
  u32x4 t[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) t[j] = reinterpret_cast<const u32x4*>(tp)[j];
  __builtin_amdgcn_sched_barrier(0);
  float x[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) x[j] = *reinterpret_cast<const float*>(srcv_r + t[j][0]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float sc = __uint_as_float(t[j][1]);
    const float g = (sc * 6.283185307179586f) * hn_sin_rev(__builtin_fmaf(sc, x[j], __uint_as_float(t[j][2]) + 0.25f));
    out[j] = hn_bfi(t[j][3], 1.0f, g);
  }
}

// bf16 mode evaluates every trigonometric kind with ONE transcendental and no branch:
//   sin(f x) = sin_rev(f x / 2pi), cos(f x) = sin(f x + pi/2) = sin_rev(f x / 2pi + 1/4)
// fp32 (parity) mode calls the precise sinf/cosf the reference's CPU path uses.
// DIRECT: the program holds identity features read straight from global memory (HN_LAYER_DIRECT).
template <bool BF16, bool DIRECT>
HN_DEV float hn_feature(const HnFeat e, const float* srcv, int r, const HnMlpArgs& a, int p, int ray) {
  const int kind = (e.packed >> 12) & 15;
  if constexpr (DIRECT) {
    if (kind == HN_FEAT_ID_DIRECT) return hn_direct_source(e, a, p, ray);
  }
  const float x = srcv[(e.packed & 255) * 32 + r];
  if constexpr (BF16) {
    const float ph = (kind == HN_FEAT_COS || kind == HN_FEAT_SINP) ? 0.25f : 0.0f;
    const float s = hn_sin_rev(__builtin_fmaf(e.freq * HN_INV_2PI, x, ph));
    return kind == HN_FEAT_ID ? x : (kind == HN_FEAT_ZERO ? 0.0f : s);
  } else {
    if (kind == HN_FEAT_ZERO) return 0.0f;
    if (kind == HN_FEAT_ID) return x;
    float arg = __fmul_rn(e.freq, x);
    if (kind == HN_FEAT_SINP) arg = __fadd_rn(arg, 0.5f * 3.1415926f);
    return kind == HN_FEAT_COS ? cosf(arg) : sinf(arg);
  }
}

// d value / d x  of a generated feature
template <bool BF16>
HN_DEV float hn_feature_grad(const HnFeat e, const float* srcv, int r) {
  const int kind = (e.packed >> 12) & 15;
  const float x = srcv[(e.packed & 255) * 32 + r];
  if constexpr (BF16) {
    // d sin(f x) = f sin_rev(t + 1/4) ; d cos(f x) = d sin(f x + pi/2) = f sin_rev(t + 1/2)
    const float ph = (kind == HN_FEAT_COS || kind == HN_FEAT_SINP) ? 0.5f : 0.25f;
    const float g = e.freq * hn_sin_rev(__builtin_fmaf(e.freq * HN_INV_2PI, x, ph));
    return (kind == HN_FEAT_ID || kind == HN_FEAT_ID_DIRECT) ? 1.0f : g;
  } else {
    if (kind == HN_FEAT_ID || kind == HN_FEAT_ID_DIRECT) return 1.0f;
    float arg = __fmul_rn(e.freq, x);
    if (kind == HN_FEAT_SINP) arg = __fadd_rn(arg, 0.5f * 3.1415926f);
    if (kind == HN_FEAT_COS) return -e.freq * sinf(arg);
    return e.freq * cosf(arg);
  }
}

// fragments of one group of 64 generated features
template <bool DIRECT>
HN_DEV void hn_make_group(bf16x8* out, const HnFeat* ft, const HnDFeat* dft, const float* srcv, int lane,
                          const HnMlpArgs& a, int p, int ray) {
  const int h = lane >> 5, r = lane & 31;
  if constexpr (DIRECT) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        out[s][j] = (__bf16)hn_feature<true, true>(ft[16 * s + hn_pi16(h, j)], srcv, r, a, p, ray);
    }
  } else {
    // Batched: the 8 entries of a fragment are two runs of 4 consecutive table entries (pi16); per run two LDS
    // round trips (entries, then source values) instead of eight (left to itself the compiler waits for every
    // read before it issues the next: 5.5k cycles per group in the in-kernel trace).
    const char* srcv_r = reinterpret_cast<const char*>(srcv) + 4 * r;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float v[8];
      hn_features4(dft + 16 * s + 4 * h, srcv_r, v);
      hn_features4(dft + 16 * s + 8 + 4 * h, srcv_r, v + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) out[s][j] = (__bf16)v[j];
    }
  }
}
template <bool DIRECT>
HN_DEV void hn_make_group(float* out, const HnFeat* ft, const HnDFeat*, const float* srcv, int lane, const HnMlpArgs& a,
                          int p, int ray) {
  const int h = lane >> 5, r = lane & 31;
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    out[s] = hn_feature<false, DIRECT>(ft[32 * (s >> 4) + hn_rho(s & 15, h)], srcv, r, a, p, ray);
    if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);
  }
}

HN_DEV void hn_init_acc(f32x16& acc, const float* bias, int t, int h) {
  if (bias != nullptr) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 32 * t + 8 * g + 4 * h);
      acc[4 * g] = b[0]; acc[4 * g + 1] = b[1]; acc[4 * g + 2] = b[2]; acc[4 * g + 3] = b[3];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  }
}

// acc += W[tile][32*K32 features] . in   (K32 consecutive blocks of the stream)
// bf16: the A fragments (one ds_read_b128 per MFMA) are read HN_PF units ahead of the MFMA that consumes them, so
// the LDS latency hides behind the MFMAs in between instead of being paid before every pair.  3, not 4: with four
// fragment buffers the forward kernel is 4 VGPRs short in its 16-MFMA tile product, spills two register pairs around it
// and reloads them at its end — and the next tile's first LDS read into those registers then waits (vmcnt(0), WAW) for
// the reload AND every stash store in front of it.  Same speed either way (measured on one box), no scratch traffic
// in any MFMA block this way (tools/scratch_report.py).
constexpr int HN_PF = 3;
template <bool BF16, int K32>
HN_DEV void hn_gemm_blocks(f32x16& acc, const typename ModeT<BF16>::Frag* in, WStream<ModeT<BF16>::WAVES>& ws) {
  using M = ModeT<BF16>;
  const char* w = ws.take(K32 * M::UNITS32);
  if constexpr (BF16) {
    constexpr int N = 2 * K32;
    constexpr int D = N < HN_PF ? N : HN_PF;
    const char* wl = w + ws.lane * 16;
    bf16x8 q[D];
#pragma unroll
    for (int u = 0; u < D; ++u) q[u] = *reinterpret_cast<const bf16x8*>(wl + u * 1024);
#pragma unroll
    for (int u = 0; u < N; ++u) {
      acc = hn_mfma_bf16(q[u % D], in[u], acc);
      if (u + D < N) q[u % D] = *reinterpret_cast<const bf16x8*>(wl + (u + D) * 1024);
    }
    // pin that order (left alone, the scheduler pulls every read back next to its MFMA to save registers)
    __builtin_amdgcn_sched_group_barrier(0x100, D, 0);
#pragma unroll
    for (int u = 0; u < N; ++u) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (u + D < N) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  } else {
#pragma unroll
    for (int k = 0; k < K32; ++k) hn_mma_block32(acc, w + k * M::UNITS32 * 1024, in + k * M::STEPS32, ws.lane);
  }
}
template <bool BF16>
HN_DEV void hn_gemm_k(f32x16& acc, const typename ModeT<BF16>::Frag* in, int K32, WStream<ModeT<BF16>::WAVES>& ws) {
  if (K32 == 8) hn_gemm_blocks<BF16, 8>(acc, in, ws);
  else if (K32 == 4) hn_gemm_blocks<BF16, 4>(acc, in, ws);
  else if (K32 == 2) hn_gemm_blocks<BF16, 2>(acc, in, ws);
  else if (K32 == 1) hn_gemm_blocks<BF16, 1>(acc, in, ws);
}

// address of tile 0 of a stash slot for one 32-point block (looked up once per op, outside the tile loops: the slot
// table sits in the kernel arguments and a lookup is two dependent scalar loads)
template <bool BF16, int S8 = 0>
HN_DEV char* hn_slot_base(const HnMlpArgs& a, int off_kib, int nt, int blk) {
  // `off_kib`: the slot's offset for block 0 in KiB, resolved by the host for this launch's point count (the op
  // word itself: no table lookup — a kernel-argument lookup is two dependent scalar loads per slot and layer)
  if (off_kib < 0) return nullptr;
  constexpr int TU = S8 ? 1 : ModeT<BF16>::TILE_UNITS;
  return reinterpret_cast<char*>(a.stash) + (size_t)(unsigned)off_kib * 1024 + (size_t)blk * nt * (TU * 1024);
}
// mask words of a block: `off256` = the slot's byte offset / 256 (resolved by the host), nt words per lane and block
HN_DEV uint32_t* hn_mask_base(const HnMlpArgs& a, int off256, int nt, int blk, int lane) {
  (void)lane;      // added at the access: the base stays wave-uniform (scalar registers, no VGPR pair per slot)
  return a.masks + (size_t)(unsigned)off256 * 64 + (size_t)blk * nt * 64;
}
// Store one 32-feature tile of a block as tile t of a stash slot.
// bf16: the operand fragments AS THEY ARE (points on lanes) — no transposing MFMAs, no second conversion: unit u of the
// tile holds fragment u, lane (r, h) at 16-byte slot hn_stash_slot(r, h, u) of the unit; the weight-gradient kernel
// turns "points on lanes" into "features on lanes" on its LDS read (ds_read_b64_tr_b16, DwFrag<true>::load), and the
// slot permutation is what makes those reads bank-conflict free.  The 64 lanes still fill exactly one 1-KiB unit.
// fp32 (parity mode; no 32-bit transposing read exists): transposed through the matrix core as before.
HN_DEV int hn_stash_slot(int r, int h, int u) { return 32 * h + (r ^ (4 * h + 8 * u)); }
// HN_MODE_BF16_S8: a tile is ONE 1-KiB unit, lane (r, h) owns 16 bytes of it — byte i = element i of its accumulator
// column (feature rho(i, h) of point r), i.e. the two fragments back to back — at 16-byte slot hn_stash8_slot(r, h).
// The weight-gradient kernel's ds_read_b64_tr_b8 takes, per group of 16 lanes, 8 rows (points) x 16 bytes, a row being
// the 8-byte halves supplied by a lane pair: with this slot order the 32 halves of a 32-lane access tile one 256-byte
// window (64 distinct banks), and 4 consecutive lanes still store 64 contiguous bytes.
HN_DEV int hn_stash8_slot(int r, int h) { return (r & 3) + 4 * h + 8 * ((r >> 2) & 1) + 16 * (r >> 3); }
// two bf16 of one register -> the two 8-bit values in half `hi` of `old` (S8 = 1: e4m3, 2: e5m2); out-of-range values
// clamp to the largest finite code (MODE.FP16_OVFL is set at the top of the S8 kernels), never to NaN / inf
template <int S8>
HN_DEV int hn_cvt2_f8(unsigned w, int old, bool hi) {
  const float lo_f = __uint_as_float(w << 16), hi_f = __uint_as_float(w & 0xffff0000u);
  if constexpr (S8 == 1) return hi ? __builtin_amdgcn_cvt_pk_fp8_f32(lo_f, hi_f, old, true)
                                   : __builtin_amdgcn_cvt_pk_fp8_f32(lo_f, hi_f, old, false);
  else return hi ? __builtin_amdgcn_cvt_pk_bf8_f32(lo_f, hi_f, old, true)
                 : __builtin_amdgcn_cvt_pk_bf8_f32(lo_f, hi_f, old, false);
}
// the same tile straight from the fp32 accumulator (8 conversions instead of 16 shifts / masks + 8 conversions)
template <int S8>
HN_DEV void hn_stash8_acc(const f32x16& a, char* slot_base, int t, int lane) {
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int w = 0;
    if constexpr (S8 == 1) {
      w = __builtin_amdgcn_cvt_pk_fp8_f32(a[4 * k], a[4 * k + 1], w, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(a[4 * k + 2], a[4 * k + 3], w, true);
    } else {
      w = __builtin_amdgcn_cvt_pk_bf8_f32(a[4 * k], a[4 * k + 1], w, false);
      w = __builtin_amdgcn_cvt_pk_bf8_f32(a[4 * k + 2], a[4 * k + 3], w, true);
    }
    o[k] = (unsigned)w;
  }
  char* dst8 = slot_base + (size_t)t * 1024 + hn_stash8_slot(lane & 31, lane >> 5) * 16;
  __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(dst8));
}
template <bool BF16, int S8 = 0>
HN_DEV void hn_stash(const typename ModeT<BF16>::Frag* fr, char* slot_base, int t, int lane) {
  using M = ModeT<BF16>;
  if constexpr (BF16 && S8 != 0) {
    const u32x4 f0 = *reinterpret_cast<const u32x4*>(&fr[0]), f1 = *reinterpret_cast<const u32x4*>(&fr[1]);
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      int lo = 0, hi = 0;
      lo = hn_cvt2_f8<S8>(f0[2 * k], lo, false);
      lo = hn_cvt2_f8<S8>(f0[2 * k + 1], lo, true);
      hi = hn_cvt2_f8<S8>(f1[2 * k], hi, false);
      hi = hn_cvt2_f8<S8>(f1[2 * k + 1], hi, true);
      o[k] = (unsigned)lo;
      o[2 + k] = (unsigned)hi;
    }
    char* dst8 = slot_base + (size_t)t * 1024 + hn_stash8_slot(lane & 31, lane >> 5) * 16;
    __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(dst8));
    return;
  }
  char* dst = slot_base + (size_t)t * (M::TILE_UNITS * 1024);
  if constexpr (BF16) {
    const int off0 = hn_stash_slot(lane & 31, lane >> 5, 0) * 16;        // unit 1: the same slot with bit 3 flipped
    __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(&fr[0]), reinterpret_cast<u32x4*>(dst + off0));
    __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(&fr[1]), reinterpret_cast<u32x4*>(dst + 1024 + (off0 ^ 128)));
  } else {
    const f32x16 z = hn_transpose_tile(fr, lane);
    hn_store_tile(z, dst, lane, (typename M::Frag*)nullptr);
  }
}

// ReLU masks: one 32-bit word per lane and PAIR of tiles; element i of tile (2d + q) is bit 31 - (16 q + i), and a
// set bit means "gradient dropped".  bf16 mode shifts in the sign bit of the pre-activation (one v_alignbit), fp32
// (parity) mode the exact reference predicate !(x > 0).
template <bool BF16>
HN_DEV unsigned hn_push_mask(unsigned bits, float x) {
  if constexpr (BF16) return __builtin_amdgcn_alignbit(bits, __float_as_uint(x), 31);
  return (bits << 1) | (x > 0.0f ? 0u : 1u);
}
// all-ones where element (q, i) of the mask word is kept, zero where it is dropped
HN_DEV int hn_keep_mask(unsigned nbits, int q, int i) { return (int)(nbits << (16 * q + i)) >> 31; }

// Op words are read through the constant address space: scalar loads (s_load_dwordx8 through the scalar cache), and
// the compiler knows no store of the kernel can change them (a plain pointer made it re-load a word with a vector
// load + vmcnt(0) after every stash store).
typedef const __attribute__((address_space(4))) int* HnOpPtr;
HN_DEV HnOpPtr hn_op_words(const int* ops, int op) {
  return (HnOpPtr)(uintptr_t)(ops + (size_t)op * HN_OP_WORDS);
}
// the 8 words of one op in scalar registers; the words of op+1 are fetched at the START of op (one s_load_dwordx8
// whose latency hides behind the layer) instead of at the top of the next iteration, where nothing could hide it
struct HnOpWords {
  int v[HN_OP_WORDS];
  HN_DEV int operator[](int i) const { return v[i]; }
};
HN_DEV HnOpWords hn_load_op(const int* ops, int op, int n_ops) {
  HnOpWords w;
  const HnOpPtr p = hn_op_words(ops, op < n_ops ? op : n_ops - 1);
#pragma unroll
  for (int i = 0; i < HN_OP_WORDS; ++i) w.v[i] = p[i];
  if (op >= n_ops) w.v[0] = -1;     // no such op
  return w;
}


// ------------------------------------------------------------------------------------------------
// Software-pipelined hidden layer for the 128-wide layers (template rgb branch, warp field): 4 input tiles -> 4 output
// tiles, bias, ReLU, commit.  A 128-wide tile has only 8 MFMAs, so in the generic loop its epilogue, mask and
// transposed stash (which the matrix pipe idles through) outweigh the products.  Here the products of tile t and the
// epilogue + stash of tile t-1 are written slot by slot in ONE basic block (two accumulators alternate; sched_barrier
// keeps the slots apart).  Same take() sequence as the generic path, so barrier counts match across waves.
// (The same body for the 256-wide layers does not fit 256 registers next to the generic path: DESIGN.md §8.)
// ------------------------------------------------------------------------------------------------
// one share of the previous tile's epilogue + stash, written so that share k only needs shares < k:
//   k = 0..7   elements 2k, 2k+1: mask bit, ReLU, pack to bf16 (fragment k>>2 complete after k = 3 / 7)
//   k = 8      the two 16-byte stash stores of the finished fragments (+ the mask word of a finished tile pair)
template <bool TRAIN, int NT, int S8>
HN_DEV void hn_epilogue_share(int k, int tp, f32x16& a, bf16x8* frag, unsigned& bits, char* out_base,
                              uint32_t* mask_base, int lane) {
  if (k < 8) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int i = 2 * k + e;
      if (TRAIN) bits = hn_push_mask<true>(bits, a[i]);
      frag[i >> 3][i & 7] = (__bf16)__int_as_float(max(__float_as_int(a[i]), 0));
    }
  } else if (TRAIN && k == 8) {
    hn_stash<true, S8>(frag, out_base, tp, lane);
    if ((tp & 1) || tp == NT - 1) {
      mask_base[(tp >> 1) * 64 + lane] = (tp & 1) ? bits : bits << 16;
      bits = 0;
    }
  }
}

template <int K32, int NT, bool TRAIN, int S8>
HN_DEV void hn_layer_pipelined(bf16x8* cur, bf16x8* nxt, const float* bias, WStream<ModeT<true>::WAVES>& ws,
                               char* out_base, uint32_t* mask_base, int lane) {
  constexpr int N = 2 * K32;            // MFMAs (= issue slots) per tile
  constexpr int D = 2;                  // fragment reads in flight
  constexpr int SHARES = 9;
  constexpr int PER_SLOT = (SHARES + N - 1) / N;      // 2 for 8 slots
  const int h = lane >> 5;
  f32x16 acc[2];
  unsigned bits = 0;
#pragma unroll
  for (int t = 0; t <= NT; ++t) {
    const char* wl = nullptr;
    if (t < NT) wl = ws.take(N) + lane * 16;        // may pass the chunk barrier (a branch): kept out of the block
    const int tp = t - 1;
    bf16x8 q[D];
    if (t < NT) {
      hn_init_acc(acc[t & 1], bias, t, h);
#pragma unroll
      for (int u = 0; u < D; ++u) q[u] = *reinterpret_cast<const bf16x8*>(wl + u * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < N; ++u) {
      // ---- slot u: one product of tile t, one fragment read ahead, shares of tile t-1's epilogue ----
      if (t < NT) {
        acc[t & 1] = hn_mfma_bf16(q[u % D], cur[u], acc[t & 1]);
        if (u + D < N) q[u % D] = *reinterpret_cast<const bf16x8*>(wl + (u + D) * 1024);
      }
      if (t > 0) {
#pragma unroll
        for (int e = 0; e < PER_SLOT; ++e) {
          const int k = u * PER_SLOT + e;
          if (k < SHARES)
            hn_epilogue_share<TRAIN, NT, S8>(k, tp, acc[tp & 1], nxt + tp * 2, bits, out_base, mask_base, lane);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < NT * 2; ++i) cur[i] = nxt[i];
}

// ------------------------------------------------------------------------------------------------
// forward machine
// ------------------------------------------------------------------------------------------------
// AUXG: generated-feature groups (64 features each) a layer may have.  Their fragments stay in registers across the
// layer's tile loop, so the common programs (<= 2 groups) get a build that does not pay for the third.
// EXTRA: the program holds identity features read directly from global memory (HN_LAYER_DIRECT) or HN_OP_OUT_WIDE ops
// — stand-alone modules with wide raw inputs / outputs; never a render-level program.  Compiled out, the bf16 kernels
// need 237 (AUXG 2) / 253 (AUXG 3) registers and no scratch (12 / 72 B/lane with them).
// TRAIN: masks and stashes are written (HnMlpArgs.training); the inference build carries none of that code.
// S8: HN_MODE_BF16_S8 — the stash is written as e4m3, 1 KiB per tile (hn_stash); nothing else changes.
template <bool BF16, int AUXG, bool EXTRA, bool TRAIN, bool S8 = false>
__global__ __launch_bounds__(ModeT<BF16>::WAVES * 64, BF16 ? 2 : 1) void hn_mlp_fwd_kernel(const HnMlpArgs a) {
  using M = ModeT<BF16>;
  constexpr int SX = S8 ? 1 : 0;
  static_assert(!(S8 && (EXTRA || !TRAIN || !BF16)), "the 8-bit stash exists for the render-level bf16 training builds only");
  if constexpr (S8) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");     // FP16_OVFL: 8-bit conversions clamp
  using Frag = typename M::Frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  constexpr int PTS = M::WAVES * 32;
  const int ntiles = (a.n_points + PTS - 1) / PTS;
  hn_timeline_begin(a.timeline);
#ifdef HN_PROF
  long long* prof_buf = reinterpret_cast<long long*>(a.prof);
  const bool prof_on = prof_buf != nullptr && blockIdx.x == 0 && wave == 0;
  int prof_n = 0;
#endif

  WStream<M::WAVES> ws;
  ws.g = reinterpret_cast<const char*>(a.wstream);
  ws.lds = smem;
  ws.nchunks = a.n_chunks;
  ws.wave = wave;
  ws.lane = lane;

  Frag cur[8 * M::STEPS32];
  Frag nxt[8 * M::STEPS32];

  // biases and the feature table live in LDS for the whole kernel: no global loads inside the MFMA loops
  float* bias_lds = reinterpret_cast<float*>(smem + 2 * HN_CHUNK_UNITS * 1024);
  HnFeat* feat_lds = reinterpret_cast<HnFeat*>(bias_lds + ((a.n_bias + 3) & ~3));
  HnDFeat* dfeat_lds = reinterpret_cast<HnDFeat*>(feat_lds + ((a.n_feat + 1) & ~1));      // bf16 kernels only
  // per wave: value plane (n_comps x 32 floats) and, bf16 only, the (hi, lo) planes of x / 2pi for the first n_trig
  const int n_trig = BF16 ? a.n_trig_comps : 0;
  const bool with_lo = a.trig_lo_planes != 0;
  const int n_planes = BF16 ? a.n_comps + n_trig + (with_lo ? n_trig : 1) : a.n_comps;
  float* srcv = reinterpret_cast<float*>(BF16 ? reinterpret_cast<char*>(dfeat_lds + a.n_feat)
                                              : reinterpret_cast<char*>(dfeat_lds)) + wave * (n_planes * 32);
  float* rev = BF16 ? srcv + a.n_comps * 32 : nullptr;
  for (int i = threadIdx.x; i < a.n_bias; i += blockDim.x) bias_lds[i] = a.bias[i];
  for (int i = threadIdx.x; i < a.n_feat; i += blockDim.x) {
    const HnFeat e = a.feat[i];
    feat_lds[i] = e;
    if constexpr (BF16) dfeat_lds[i] = hn_derive_feat_fwd(e, a.n_comps, n_trig, with_lo);
  }
  __syncthreads();

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int blk = tile * M::WAVES + wave;  // 32-point block of this wave
    const int p0 = blk * 32 + r;
    const bool valid = p0 < a.n_points;
    const int p = valid ? p0 : a.n_points - 1;
    const int ray = p / a.samples_per_ray;
    const bool wave_valid = blk * 32 < a.n_points;  // wave-uniform: the block holds at least one point
    hn_stage_sources(srcv, a, p, ray, lane, rev, n_trig, with_lo);
    ws.start();

    HnOpWords w_next = hn_load_op(a.ops, 0, a.n_ops);
    for (int op = 0; op < a.n_ops; ++op) {
      const HnOpWords w = w_next;
      w_next = hn_load_op(a.ops, op + 1, a.n_ops);
      const int code = w[0];
      if (code == HN_OP_LAYER) {
        HN_STAMP(100 + op);
        const int K32 = w[1] & 255, nG = (w[1] >> 8) & 255, NT = (w[1] >> 16) & 255;
        const int act = (w[1] >> 24) & 15, flags = (w[1] >> 28) & 15;
        const float* bias = bias_lds + w[2];
        const bool do_mask = TRAIN && w[4] >= 0 && wave_valid;
        const bool do_stash = TRAIN && w[5] >= 0 && wave_valid;
        char* out_base = do_stash ? hn_slot_base<BF16, SX>(a, w[5], NT, blk) : nullptr;
        char* aux_base = (TRAIN && wave_valid) ? hn_slot_base<BF16, SX>(a, w[6], 2 * nG, blk) : nullptr;
        uint32_t* mask_base = do_mask ? hn_mask_base(a, w[4], (NT + 1) >> 1, blk, lane) : nullptr;
        const bool has_out = w_next[0] == HN_OP_OUT;    // head layer: its <=4 outputs leave from the accumulator
        const HnOpWords out_w = w_next;
        Frag aux[AUXG * 2 * M::STEPS32];
#pragma unroll
        for (int g = 0; g < AUXG; ++g) {
          if (g < nG) {
            if (EXTRA && (flags & HN_LAYER_DIRECT))
              hn_make_group<true>(aux + g * 2 * M::STEPS32, feat_lds + w[3] + 64 * g, dfeat_lds + w[3] + 64 * g, srcv,
                                  lane, a, p, ray);
            else
              hn_make_group<false>(aux + g * 2 * M::STEPS32, feat_lds + w[3] + 64 * g, dfeat_lds + w[3] + 64 * g, srcv,
                                   lane, a, p, ray);
            if (aux_base != nullptr) {
              hn_stash<BF16, SX>(aux + g * 2 * M::STEPS32, aux_base, 2 * g, lane);
              hn_stash<BF16, SX>(aux + (g * 2 + 1) * M::STEPS32, aux_base, 2 * g + 1, lane);
            }
          }
        }
        unsigned bits = 0;
        HN_STAMP(1);
        if constexpr (BF16) {
          // plain 128 -> 128 ReLU layers (template rgb branch, warp field) take the pipelined body (adding the 64-wide
          // shape as well tips the register allocator into spilling inside the generic path: measured 7 % slower)
          const bool plain = nG == 0 && act == HN_ACT_RELU && !(flags & HN_LAYER_NO_COMMIT) && !has_out &&
                             wave_valid && do_stash == do_mask;
          if (plain && K32 == 4 && NT == 4) {
            if (do_stash) hn_layer_pipelined<4, 4, true, SX>(cur, nxt, bias, ws, out_base, mask_base, lane);
            else hn_layer_pipelined<4, 4, false, SX>(cur, nxt, bias, ws, out_base, mask_base, lane);
            continue;
          }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (t < NT) {
            f32x16 acc;
            hn_init_acc(acc, bias, t, h);
            hn_gemm_k<BF16>(acc, cur, K32, ws);
#pragma unroll
            for (int g = 0; g < AUXG; ++g)
              if (g < nG) hn_gemm_blocks<BF16, 2>(acc, aux + g * 2 * M::STEPS32, ws);
            HN_STAMP(2);
            if (act == HN_ACT_RELU) {
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                bits = hn_push_mask<BF16>(bits, acc[i]);
                acc[i] = __int_as_float(max(__float_as_int(acc[i]), 0));  // relu on the bit pattern: one v_max_i32
              }
            }
            if (has_out && t == 0 && h == 0 && valid) {
              // the OUT op that follows a head layer: <= 4 fp32 columns straight from the accumulator
              const HnDst d = a.dst[out_w[1]];
              const int n = out_w[3];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                if (i < n) {
                  float y = acc[i];
                  if (out_w[4] == 1) y = 1.0f / (1.0f + expf(-y));
                  if (out_w[5] >= 0) {
                    const HnSrc sr = a.src[out_w[5]];
                    y = __fadd_rn(sr.ptr[(size_t)(sr.per_ray ? ray : p) * sr.ld + out_w[6] + i], y);
                  }
                  d.ptr[(size_t)p * d.ld + out_w[2] + i] = y;
                }
              }
            }
            if (has_out && t == 0 && h == 0 && out_w[7] > 0) {
              // publish the head's results as staged components of this block (every lane: padded points too)
              const int n = out_w[3];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                if (i < n) {
                  float y = acc[i];
                  if (out_w[4] == 1) y = 1.0f / (1.0f + expf(-y));
                  if (out_w[5] >= 0) {
                    const HnSrc sr = a.src[out_w[5]];
                    y = __fadd_rn(sr.ptr[(size_t)(sr.per_ray ? ray : p) * sr.ld + out_w[6] + i], y);
                  }
                  const int ci = out_w[7] - 1 + i;
                  srcv[ci * 32 + r] = y;
                  if (BF16 && ci < n_trig) {
                    float hi, lo;
                    hn_rev_split(y, hi, lo);
                    rev[ci * 32 + r] = hi;
                    if (with_lo) rev[(n_trig + ci) * 32 + r] = lo;
                  }
                }
              }
            }
            hn_acc_to_frags(acc, nxt + t * M::STEPS32);
#ifdef HN_PROF
            if (prof_on) {   // make the stamp wait for the tile's results
              float x__ = acc[15];
              asm volatile("v_mov_b32 %0, %0" : "+v"(x__)::"memory");
            }
#endif
            HN_STAMP(3);
            if ((t & 1) || t == NT - 1) {
              // a plain (cached) store: the backward machine stalls on these words at the top of every layer, and
              // unlike the once-streamed stash they are small enough (70 MB per step) to survive in L2 / MALL
              if (do_mask) mask_base[(t >> 1) * 64 + lane] = (t & 1) ? bits : bits << 16;
              bits = 0;
            }
            if (do_stash) {
              if constexpr (SX != 0) hn_stash8_acc<SX>(acc, out_base, t, lane);
              else hn_stash<BF16, SX>(nxt + t * M::STEPS32, out_base, t, lane);
            }
            HN_STAMP(4);
          }
        }
        if (!(flags & HN_LAYER_NO_COMMIT)) {
#pragma unroll
          for (int t = 0; t < 8; ++t)
            if (t < NT)
#pragma unroll
              for (int s = 0; s < M::STEPS32; ++s) cur[t * M::STEPS32 + s] = nxt[t * M::STEPS32 + s];
        }
      } else if (code == HN_OP_OUT) {
        // handled inside the LAYER op it follows (the raw accumulator is only alive there)
      } else if (EXTRA && code == HN_OP_OUT_WIDE) {
        const int n = w[3], NT = w[4];
        if (valid) {
          const HnDst d = a.dst[w[1]];
#pragma unroll
          for (int t = 0; t < 8; ++t)
            if (t < NT) {
              if constexpr (BF16) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                  for (int j = 0; j < 8; ++j) {
                    const int row = 32 * t + 16 * s + hn_pi16(h, j);
                    if (row < n) d.ptr[(size_t)p * d.ld + w[2] + row] = (float)cur[2 * t + s][j];
                  }
              } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                  const int row = 32 * t + hn_rho(q, h);
                  if (row < n) d.ptr[(size_t)p * d.ld + w[2] + row] = cur[16 * t + q];
                }
              }
            }
        }
      }
    }
  }
  hn_timeline_end(a.timeline);
}

// Backward machine: if the NEXT op is a layer with a ReLU mask, start loading its mask words (raw, not complemented)
// now.  Returns whether it did.  Words of tile pairs the layer does not have read as 0 (= keep everything).
HN_DEV bool hn_prefetch_masks(const HnMlpArgs& a, const HnOpWords& wn, int blk, int lane, bool wave_valid,
                              unsigned* out) {
  if (wn[0] != HN_BOP_LAYER || wn[4] < 0) return false;
  const int NT = (wn[1] >> 16) & 255;
  const uint32_t* mb = hn_mask_base(a, wn[4], (NT + 1) >> 1, blk, lane);
#pragma unroll
  for (int dd = 0; dd < 4; ++dd)
    out[dd] = (2 * dd < NT) ? (wave_valid ? mb[dd * 64 + lane] : 0xffffffffu) : 0u;
  return true;
}

// ------------------------------------------------------------------------------------------------
// backward-data machine
// ------------------------------------------------------------------------------------------------
// WIDE: the program holds HN_BOP_LOAD_WIDE ops (stand-alone modules with > 4 output columns; never a render-level
// program).  Compiled out, the bf16 kernel needs no scratch at all (188 B/lane with it: the wide load's 16-value gather
// per tile pushes the allocator over 256 registers in the prologue).
// S8: HN_MODE_BF16_S8 — the machine carries 2^dz_scale_log2 * dZ (exact) and stashes it as e5m2, 1 KiB per tile; source
// and embedding gradients are scaled back on their way out.
template <bool BF16, bool WIDE, bool S8 = false>
__global__ __launch_bounds__(ModeT<BF16>::WAVES * 64, BF16 ? 2 : 1) void hn_mlp_bwd_kernel(const HnMlpArgs a) {
  using M = ModeT<BF16>;
  constexpr int SZ = S8 ? 2 : 0;
  static_assert(!(S8 && WIDE) && !(S8 && !BF16), "the 8-bit stash exists for the render-level bf16 builds only");
  if constexpr (S8) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");     // FP16_OVFL: 8-bit conversions clamp
  const float dz_scale = S8 ? ldexpf(1.0f, a.dz_scale_log2) : 1.0f;
  const float dz_unscale = S8 ? ldexpf(1.0f, -a.dz_scale_log2) : 1.0f;
  using Frag = typename M::Frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  constexpr int PTS = M::WAVES * 32;
  const int ntiles = (a.n_points + PTS - 1) / PTS;
  HnFeat* feat_lds = reinterpret_cast<HnFeat*>(smem + 2 * HN_CHUNK_UNITS * 1024);
  HnDFeat* dfeat_lds = reinterpret_cast<HnDFeat*>(feat_lds + ((a.n_feat + 1) & ~1));      // bf16 kernel only
  float* srcv = reinterpret_cast<float*>(BF16 ? reinterpret_cast<char*>(dfeat_lds + a.n_feat)
                                              : reinterpret_cast<char*>(dfeat_lds)) + wave * (a.n_comps * 32);
  hn_timeline_begin(a.timeline);
#ifdef HN_PROF
  long long* prof_buf = reinterpret_cast<long long*>(a.prof);
  const bool prof_on = prof_buf != nullptr && blockIdx.x == 0 && wave == 0;
  int prof_n = 0;
#endif
  for (int i = threadIdx.x; i < a.n_feat; i += blockDim.x) {
    const HnFeat e = a.feat[i];
    feat_lds[i] = e;
    if constexpr (BF16) dfeat_lds[i] = hn_derive_feat(e);
  }
  __syncthreads();

  WStream<M::WAVES> ws;
  ws.g = reinterpret_cast<const char*>(a.wstream);
  ws.lds = smem;
  ws.nchunks = a.n_chunks;
  ws.wave = wave;
  ws.lane = lane;

  Frag cur[8 * M::STEPS32];
  Frag nxt[8 * M::STEPS32];
  Frag cur2[M::STEPS32];

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int blk = tile * M::WAVES + wave;
    const int p0 = blk * 32 + r;
    const bool valid = p0 < a.n_points;
    const int p = valid ? p0 : a.n_points - 1;
    const int ray = p / a.samples_per_ray;
    const bool wave_valid = blk * 32 < a.n_points;
    f32x16 dacc;  // source gradients of the block: row rho(i,h) = dsrc column, col r = point
    hn_init_acc(dacc, nullptr, 0, h);
    hn_stage_sources(srcv, a, p, ray, lane);
    ws.start();

    HnOpWords w_next = hn_load_op(a.ops, 0, a.n_ops);
    unsigned mnext[4] = {0u, 0u, 0u, 0u};
    bool mnext_ready = false;
    for (int op = 0; op < a.n_ops; ++op) {
      const HnOpWords w = w_next;
      w_next = hn_load_op(a.ops, op + 1, a.n_ops);
      const int code = w[0];
      HN_STAMP(100 * code + op);
      if (code != HN_BOP_LAYER && code != HN_BOP_AUX) mnext_ready = false;   // LOAD ops do not prefetch
      if (code == HN_BOP_LOAD) {
        // dZ (<= 4 columns) of an output layer -> one 32-feature tile
        const int n = w[3] & 255;
        const bool to2 = (w[3] >> 8) & 1;
        float d[4] = {0.f, 0.f, 0.f, 0.f};
        if ((w[3] >> 9) & 1) {
          // the gradient later ops left in the source-gradient accumulators for the components this head published:
          // slots 8q + i sit in registers 4q + i of the h == 0 lanes (row rho(4q + i, 0) = 8q + i)
          const int q = (w[3] >> 10) & 3;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            d[i] = q == 0 ? dacc[i] : (q == 1 ? dacc[4 + i] : (q == 2 ? dacc[8 + i] : dacc[12 + i]));
        }
        if (h == 0 && valid) {
          const HnSrc s = a.src[w[1] < 0 ? 0 : w[1]];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (i < n) {
              float g = (w[1] >= 0 && s.ptr != nullptr) ? s.ptr[(size_t)p * s.ld + w[2] + i] : 0.0f;
              if constexpr (S8) g *= dz_scale;
              if ((w[3] >> 9) & 1) g += d[i];
              if (w[4] == 1) {
                const HnSrc ys = a.src[w[5]];
                const float y = ys.ptr[(size_t)p * ys.ld + w[6] + i];
                g = g * y * (1.0f - y);
              }
              d[i] = g;
            } else {
              d[i] = 0.0f;
            }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) d[i] = 0.0f;
        }
        Frag tmp[M::STEPS32];
        hn_zero_frags(tmp, M::STEPS32);
        if constexpr (BF16) {
#pragma unroll
          for (int i = 0; i < 4; ++i) tmp[0][i] = (__bf16)d[i];  // h==0, j<4 <-> features 0..3
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) tmp[i] = d[i];             // step q, h==0 <-> feature q
        }
        if (a.training && w[7] >= 0 && wave_valid) hn_stash<BF16, SZ>(tmp, hn_slot_base<BF16, SZ>(a, w[7], 1, blk), 0, lane);
#pragma unroll
        for (int s = 0; s < M::STEPS32; ++s) {
          if (to2) cur2[s] = tmp[s];
          else cur[s] = tmp[s];
        }
      } else if (WIDE && code == HN_BOP_LOAD_WIDE) {
        const int n = w[3], NT = w[4];
        const HnSrc s = a.src[w[1]];
        unsigned nbits = 0xffffffffu;  // complement of the mask word: set = keep
        char* dz_base = (a.training && wave_valid) ? hn_slot_base<BF16>(a, w[7], NT, blk) : nullptr;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (t < NT) {
            if (w[5] >= 0 && !(t & 1)) {  // output activation was relu: dZ = dY * relu'
              nbits = wave_valid ? ~hn_mask_base(a, w[5], (NT + 1) >> 1, blk, lane)[(t >> 1) * 64 + lane] : 0u;
            }
            f32x16 v;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int row = 32 * t + hn_rho(i, h);
              const bool keep = hn_keep_mask(nbits, t & 1, i) != 0;
              v[i] = (valid && keep && row < n) ? s.ptr[(size_t)p * s.ld + w[2] + row] : 0.0f;
            }
            hn_acc_to_frags(v, cur + t * M::STEPS32);
            if (dz_base != nullptr) hn_stash<BF16>(cur + t * M::STEPS32, dz_base, t, lane);
          }
        }
      } else if (code == HN_BOP_LAYER) {
        const int K32 = w[1] & 255, K32b = (w[1] >> 8) & 255, NT = (w[1] >> 16) & 255;
        const bool has_mask = w[4] >= 0;
        const bool do_stash = a.training && w[5] >= 0 && wave_valid;
        char* dz_base = do_stash ? hn_slot_base<BF16, SZ>(a, w[5], NT, blk) : nullptr;
        unsigned nbits = 0xffffffffu;
        unsigned mbits[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};  // complemented words: set = keep
        if (has_mask) {
          if (mnext_ready) {   // fetched while the previous op ran (hn_prefetch_masks): no memory wait here
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) mbits[dd] = ~mnext[dd];
          } else {             // all relu masks of the layer up front: one VMEM wait per layer, none per tile
            const uint32_t* mb = hn_mask_base(a, w[4], (NT + 1) >> 1, blk, lane);
#pragma unroll
            for (int dd = 0; dd < 4; ++dd)
              if (2 * dd < NT)
                mbits[dd] = wave_valid ? ~mb[dd * 64 + lane] : 0u;
          }
        }
        mnext_ready = false;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (t < NT) {
            f32x16 acc;
            hn_init_acc(acc, nullptr, t, h);
            hn_gemm_k<BF16>(acc, cur, K32, ws);
            if (K32b) hn_gemm_blocks<BF16, 1>(acc, cur2, ws);
            // the next layer's mask words, issued behind this tile's chunk barrier so that they have the rest of
            // the layer to arrive (the backward machine used to stall ~3 us on them at the top of every layer)
            if (t == 0) mnext_ready = hn_prefetch_masks(a, w_next, blk, lane, wave_valid, mnext);
            HN_STAMP(2);
            if (!(t & 1)) nbits = mbits[t >> 1];
            // dZ of padded points is zero from the LOAD ops on and stays zero: no `valid` select here
#pragma unroll
            for (int i = 0; i < 16; ++i)
              acc[i] = __int_as_float(__float_as_int(acc[i]) & hn_keep_mask(nbits, t & 1, i));
            hn_acc_to_frags(acc, nxt + t * M::STEPS32);
            if (do_stash) {
              if constexpr (SZ != 0) hn_stash8_acc<SZ>(acc, dz_base, t, lane);
              else hn_stash<BF16, SZ>(nxt + t * M::STEPS32, dz_base, t, lane);
            }
            HN_STAMP(4);
          }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (t < NT)
#pragma unroll
            for (int s = 0; s < M::STEPS32; ++s) cur[t * M::STEPS32 + s] = nxt[t * M::STEPS32 + s];
      } else if (code == HN_BOP_AUX) {
        // gradient of generated features: per 32-feature tile tmp = W_aux^T . dZ, then the chain rule
        const int K32 = w[1] & 255, K32b = (w[1] >> 8) & 255, nG = (w[1] >> 16) & 255;
        // w2: bit tt = tile tt holds a feature that differentiates into a source (the others are neither in the weight
        // stream nor computed: encoders of plain inputs in front of a GLO row, padding); bit 8 + tt = one of them is
        // trigonometric (a tile of identity features only — GLO rows — takes W^T dZ as it is: d feature / dx = 1)
        const int tiles = w[2];
        bool first = true;
        mnext_ready = false;
        for (int tt = 0; tt < 2 * nG; ++tt) {
          if (!((tiles >> tt) & 1)) continue;
          f32x16 acc;
          hn_init_acc(acc, nullptr, 0, h);
          hn_gemm_k<BF16>(acc, cur, K32, ws);
          if (K32b) hn_gemm_blocks<BF16, 1>(acc, cur2, ws);
          if (first) mnext_ready = hn_prefetch_masks(a, w_next, blk, lane, wave_valid, mnext);
          first = false;
          // chain rule per feature, then the reduction over the features of each source component as one more
          // matrix product: dacc[slot][point] += S[slot][feature] . G[feature][point]  (S: 0/1 selection block that
          // the host put in the weight stream right behind this tile's weights).  No LDS accumulators: an LDS
          // atomic after an LDS-DMA makes the compiler drain vmcnt, i.e. the weight prefetch.
          const HnFeat* ft = feat_lds + w[3] + 32 * tt;
          if ((tiles >> (8 + tt)) & 1) {
            if constexpr (BF16) {
              // accumulator register i is feature rho(i, h): four runs of 4 consecutive table entries
              const HnDFeat* dft = dfeat_lds + w[3] + 32 * tt + 4 * h;
              const char* srcv_r = reinterpret_cast<const char*>(srcv) + 4 * r;
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) {
                float g[4];
                hn_feature_grads4(dft + 8 * q4, srcv_r, g);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * q4 + e] *= g[e];
              }
            } else {
#pragma unroll
              for (int i = 0; i < 16; ++i) acc[i] *= hn_feature_grad<BF16>(ft[hn_rho(i, h)], srcv, r);
            }
          }
          Frag gfr[M::STEPS32];
          hn_acc_to_frags(acc, gfr);
          hn_gemm_blocks<BF16, 1>(dacc, gfr, ws);
          HN_STAMP(5);
        }
      }
    }
    // source gradients of this block -> global
    if (a.n_dsrc > 0 && a.dsrc != nullptr && p0 < a.n_points) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int slot = hn_rho(i, h);
        if (slot < a.n_dsrc) a.dsrc[(size_t)p0 * a.n_dsrc + slot] = S8 ? dacc[i] * dz_unscale : dacc[i];
      }
    }
    // GLOEmbed backward (modules.py:155-167 under autograd): the block's 32 points belong to one ray, so the
    // gradient of the gathered row is the sum over the lanes of each half, added once per block and component
    if (a.embed_reg_mask != 0 && wave_valid) {
      const long long erow = a.embed_idx[(blk * 32) / a.samples_per_ray];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (a.embed_reg_mask & (1 << i)) {
          float v = valid ? (S8 ? dacc[i] * dz_unscale : dacc[i]) : 0.0f;
#pragma unroll
          for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
          const int col = a.embed_col[hn_rho(i, h)];
          if (r == 0 && col >= 0) {
            // (the opt-in 8-bit-stash build keeps the atomics: the host never hands it a partial buffer, and the extra
            // pointer pushed that build's allocation into scratch)
            if (!S8 && a.embed_partial != nullptr) a.embed_partial[(size_t)blk * a.embed_dim + col] = v;
            else if (erow >= 0 && erow < a.embed_rows) atomicAdd(a.embed_grad + (size_t)erow * a.embed_dim + col, v);
          }
        }
      }
    }
  }
  hn_timeline_end(a.timeline);
}

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
#include "hn_pack.h"
template <bool BF16>
__global__ void hn_pack_kernel(const HnPackUnit* units, int n_units, const float* const* ptrs, char* out,
                               const HnPackBias* bias, int n_bias, float* bias_out) {
  hn_pack_one<BF16>(units, n_units, ptrs, out, bias, n_bias, bias_out, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
}
template <bool BF16>
__global__ void hn_pack_multi_kernel(const HnPackTable tab) {
  hn_pack_block<BF16>(tab, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// weight gradient: dW[n][k] += sum_p dZ[p][n] X[p][k] straight from the transposed stashes
// ------------------------------------------------------------------------------------------------
template <class F>
HN_DEV void static_for4(F&& f) {
  f(std::integral_constant<int, 0>{});
  f(std::integral_constant<int, 1>{});
  f(std::integral_constant<int, 2>{});
  f(std::integral_constant<int, 3>{});
}
template <bool BF16>
struct DwFrag;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
// byte offsets inside a 2-KiB bf16 stash tile of this lane's transposed reads (hn_stash's layout).  The MFMA operand
// of the dW product wants "feature on the lane, 8 points in the registers"; the stash has "point on the lane, 8
// features in the registers".  ds_read_b64_tr_b16 gathers, per group of 16 lanes, 4 rows x 16 columns and hands lane
// i column i: rows = 4 points, columns = 16 features (four 8-byte pieces of 4 features each).  Lane 4q + p of group
// G supplies the piece (unit G&1, lane half p&1, register half p>>1) of point 16 mm + 8 (G>>1) + 4 jh + q for read
// (mm, jh); it receives feature 16 (G&1) + (its index in the group) at those 4 points.  Reads jh = 0, 1 make the 8
// points of one operand (k = 8 hh + 4 jh + q), mm = 0, 1 the two operands of a 32-point block.  With the slot
// permutation of hn_stash_slot the 32 lanes of a half hit 64 distinct banks.
HN_DEV void hn_dw_tr_offsets(int lane, int& o0, int& o1) {
  const int G = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int u = G & 1, hh = G >> 1, h = pp & 1, g = pp >> 1;
  o0 = u * 1024 + hn_stash_slot(8 * hh + q, h, u) * 16 + 8 * g;
  o1 = u * 1024 + hn_stash_slot(8 * hh + 4 + q, h, u) * 16 + 8 * g;
}
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
// The four transposed reads of one tile + the wait for them, as ONE asm statement.  (The compiler builtin of
// ds_read_b64_tr_b16 carries no memory operand, so next to LDS-DMA in flight hipcc guards every one of them with
// s_waitcnt vmcnt(0) — which drains the whole stage ring: measured 0.70 -> 1.00 ms on the weight-gradient launch.  The
// asm form is invisible to that rule; the data dependence of the MFMAs on its outputs orders them behind the wait.)
// a0 / a1: LDS byte addresses of the lane's reads jh = 0 / 1 in tile 0; OFF: byte offset of the tile.
template <int OFF>
HN_DEV void hn_tr_tile(bf16x8* v, unsigned a0, unsigned a1) {
  u32x2 l0, h0, l1, h1;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %4 offset:%6\n\t"
      "ds_read_b64_tr_b16 %1, %5 offset:%6\n\t"
      "ds_read_b64_tr_b16 %2, %4 offset:%7\n\t"
      "ds_read_b64_tr_b16 %3, %5 offset:%7\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1)
      : "v"(a0), "v"(a1), "n"(OFF), "n"(OFF + 256)
      : "memory");
  const u32x4 w0 = {l0[0], l0[1], h0[0], h0[1]}, w1 = {l1[0], l1[1], h1[0], h1[1]};
  v[0] = __builtin_bit_cast(bf16x8, w0);
  v[1] = __builtin_bit_cast(bf16x8, w1);
}
// All operand tiles of one 32-point block of a wave's rectangle (2 X tiles, NZ dZ tiles of 2 KiB) under ONE wait: 8 + 4 NZ
// transposed reads in flight together instead of a read-wait-multiply round trip per tile (an LDS round trip next to the
// ring's DMA writes is ~300 cycles; six of them per block were the critical path of a stage).  Tiles the wave does not own
// are read all the same (never used; reads past the end of LDS return zero).
template <int NZ>
HN_DEV void hn_tr_block(bf16x8 (*x)[2], bf16x8 (*z)[2], unsigned ax0, unsigned ax1, unsigned az0, unsigned az1) {
  u32x2 r[8 + 4 * NZ];
  if constexpr (NZ == 4) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %24\n\t"
        "ds_read_b64_tr_b16 %1, %25\n\t"
        "ds_read_b64_tr_b16 %2, %24 offset:256\n\t"
        "ds_read_b64_tr_b16 %3, %25 offset:256\n\t"
        "ds_read_b64_tr_b16 %4, %24 offset:2048\n\t"
        "ds_read_b64_tr_b16 %5, %25 offset:2048\n\t"
        "ds_read_b64_tr_b16 %6, %24 offset:2304\n\t"
        "ds_read_b64_tr_b16 %7, %25 offset:2304\n\t"
        "ds_read_b64_tr_b16 %8, %26\n\t"
        "ds_read_b64_tr_b16 %9, %27\n\t"
        "ds_read_b64_tr_b16 %10, %26 offset:256\n\t"
        "ds_read_b64_tr_b16 %11, %27 offset:256\n\t"
        "ds_read_b64_tr_b16 %12, %26 offset:2048\n\t"
        "ds_read_b64_tr_b16 %13, %27 offset:2048\n\t"
        "ds_read_b64_tr_b16 %14, %26 offset:2304\n\t"
        "ds_read_b64_tr_b16 %15, %27 offset:2304\n\t"
        "ds_read_b64_tr_b16 %16, %26 offset:4096\n\t"
        "ds_read_b64_tr_b16 %17, %27 offset:4096\n\t"
        "ds_read_b64_tr_b16 %18, %26 offset:4352\n\t"
        "ds_read_b64_tr_b16 %19, %27 offset:4352\n\t"
        "ds_read_b64_tr_b16 %20, %26 offset:6144\n\t"
        "ds_read_b64_tr_b16 %21, %27 offset:6144\n\t"
        "ds_read_b64_tr_b16 %22, %26 offset:6400\n\t"
        "ds_read_b64_tr_b16 %23, %27 offset:6400\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),
          "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15]),
          "=&v"(r[16]), "=&v"(r[17]), "=&v"(r[18]), "=&v"(r[19]), "=&v"(r[20]), "=&v"(r[21]), "=&v"(r[22]), "=&v"(r[23])
        : "v"(ax0), "v"(ax1), "v"(az0), "v"(az1)
        : "memory");
  } else {
    static_assert(NZ == 2, "");
    asm volatile(
        "ds_read_b64_tr_b16 %0, %16\n\t"
        "ds_read_b64_tr_b16 %1, %17\n\t"
        "ds_read_b64_tr_b16 %2, %16 offset:256\n\t"
        "ds_read_b64_tr_b16 %3, %17 offset:256\n\t"
        "ds_read_b64_tr_b16 %4, %16 offset:2048\n\t"
        "ds_read_b64_tr_b16 %5, %17 offset:2048\n\t"
        "ds_read_b64_tr_b16 %6, %16 offset:2304\n\t"
        "ds_read_b64_tr_b16 %7, %17 offset:2304\n\t"
        "ds_read_b64_tr_b16 %8, %18\n\t"
        "ds_read_b64_tr_b16 %9, %19\n\t"
        "ds_read_b64_tr_b16 %10, %18 offset:256\n\t"
        "ds_read_b64_tr_b16 %11, %19 offset:256\n\t"
        "ds_read_b64_tr_b16 %12, %18 offset:2048\n\t"
        "ds_read_b64_tr_b16 %13, %19 offset:2048\n\t"
        "ds_read_b64_tr_b16 %14, %18 offset:2304\n\t"
        "ds_read_b64_tr_b16 %15, %19 offset:2304\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),
          "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
        : "v"(ax0), "v"(ax1), "v"(az0), "v"(az1)
        : "memory");
  }
#pragma unroll
  for (int t = 0; t < 2 + NZ; ++t)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const u32x4 w = {r[4 * t + 2 * m][0], r[4 * t + 2 * m][1], r[4 * t + 2 * m + 1][0], r[4 * t + 2 * m + 1][1]};
      if (t < 2) x[t][m] = __builtin_bit_cast(bf16x8, w);
      else z[t - 2][m] = __builtin_bit_cast(bf16x8, w);
    }
}
template <>
struct DwFrag<true> {
  bf16x8 v[2];
  template <int OFF>
  HN_DEV void load_tr(unsigned a0, unsigned a1) { hn_tr_tile<OFF>(v, a0, a1); }
  HN_DEV void load(const char*, int, int, int) {}
  HN_DEV static void mma(f32x16& acc, const DwFrag& a, const DwFrag& b) {
    acc = hn_mfma_bf16(a.v[0], b.v[0], acc);
    acc = hn_mfma_bf16(a.v[1], b.v[1], acc);
  }
  HN_DEV static void mma_ones(f32x16& acc, const DwFrag& a, int, int) {      // one accumulator per tile, every column
    bf16x8 one;
#pragma unroll
    for (int j = 0; j < 8; ++j) one[j] = (__bf16)1.0f;
    acc = hn_mfma_bf16(a.v[0], one, acc);
    acc = hn_mfma_bf16(a.v[1], one, acc);
  }
  // bias gradient on the vector pipe (round 5): s + the sum of this lane's 16 values — feature (lane & 31) of the dZ
  // tile at the 16 points of the lane's half — as eight v_dot2c_f32_bf16 with a (1, 1) operand.  The all-ones MFMAs it
  // replaces (2 per dZ tile and block, 64 more accumulator registers, and waves that own more bias tiles than their
  // SIMD partner reach every stage barrier late) cost 7 % of the launch: timing builds, profiles/r05_wgrad_end_of_job.log
  HN_DEV float add_point_sum(float s) const {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 one = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 8; j += 2) s = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{v[u][j], v[u][j + 1]}, one, s, false);
    return s;
  }
};
template <>
struct DwFrag<false> {
  f32x4 v[4];
  HN_DEV void load(const char* tile, int lane, int, int) {
#pragma unroll
    for (int g = 0; g < 4; ++g) v[g] = *reinterpret_cast<const f32x4*>(tile + g * 1024 + lane * 16);
  }
  template <int OFF>
  HN_DEV void load_tr(unsigned, unsigned) {}
  HN_DEV static void mma(f32x16& acc, const DwFrag& a, const DwFrag& b) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = hn_mfma_f32(a.v[g][e], b.v[g][e], acc);
  }
  HN_DEV static void mma_ones(f32x16& acc, const DwFrag& a, int, int) {     // fp32: one accumulator per tile, every column
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = hn_mfma_f32(a.v[g][e], 1.0f, acc);
  }
  HN_DEV float add_point_sum(float s) const {       // the lane's 16 points of feature (lane & 31)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += v[g][e];
    return s;
  }
};

// HN_MODE_BF16_S8: one ds_read_b64_tr_b8 per 16-point operand.  Per group of 16 lanes it gathers 8 rows x 16 bytes, row b =
// the 8-byte pieces supplied by lanes 2b and 2b+1, and hands lane i column i (tools/tr_b8_probe.hip).  Lane 2b + hh of
// group G supplies the half (fragment G & 1) of forward lane (point 16 mm + 8 (G >> 1) + b, half hh) for read mm; it
// receives, for those 8 points, accumulator element 8 (G & 1) + (i & 7) of half i >> 3 — feature hn_dw8_feature(c) of
// the tile, c = lane & 31: a fixed permutation, the same for both operands, undone in the epilogue's addresses.
HN_DEV int hn_dw8_offset(int lane) {
  const int G = lane >> 4, i = lane & 15;
  return hn_stash8_slot(8 * (G >> 1) + (i >> 1), i & 1) * 16 + 8 * (G & 1);
}
HN_DEV int hn_dw8_feature(int c) { return hn_rho(8 * (c >> 4) + (c & 7), (c >> 3) & 1); }
struct DwFrag8 {
  long v[2];
  template <int OFF>
  HN_DEV void load_tr(unsigned a0, unsigned) {
    u32x2 l0, l1;
    asm volatile(
        "ds_read_b64_tr_b8 %0, %2 offset:%3\n\t"
        "ds_read_b64_tr_b8 %1, %2 offset:%4\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(l0), "=&v"(l1)
        : "v"(a0), "n"(OFF), "n"(OFF + 512)
        : "memory");
    v[0] = __builtin_bit_cast(long, l0);
    v[1] = __builtin_bit_cast(long, l1);
  }
  HN_DEV void load(const char*, int, int, int) {}
  // a: dZ (e5m2), b: X (e4m3)
  HN_DEV static void mma(f32x16& acc, const DwFrag8& a, const DwFrag8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(a.v[0], b.v[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(a.v[1], b.v[1], acc, 0, 0, 0);
  }
  // bias gradient of tile `col` (0..3) into COLUMN `col` of the one shared accumulator: the B operand is all ones on the
  // lanes of that column and zero elsewhere, so four tiles' row sums live side by side in 16 registers instead of 64
  HN_DEV float add_point_sum(float s) const { return s; }      // (the 8-bit stash keeps the all-ones MFMA)
  HN_DEV static void mma_ones(f32x16& acc, const DwFrag8& a, int col, int lane) {
    const long one = ((lane & 31) == col) ? 0x3838383838383838L : 0L;       // e4m3 1.0 in every byte of column `col`
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(a.v[0], one, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(a.v[1], one, acc, 0, 0, 0);
  }
};

// All operands of one 32-point block of a wave's rectangle (2 X tiles, 4 dZ tiles of 1 KiB) with ONE wait: 12 reads in
// flight together instead of a read-wait-multiply round trip per tile (the 8-bit fragments are half the registers of
// the bf16 ones, which is what makes room for all six).  Tiles the wave does not own are read all the same (their
// values are never used; reads past the end of LDS return zero).
HN_DEV void hn_tr8_block(DwFrag8* xb, DwFrag8* za, unsigned ax, unsigned az) {
  u32x2 r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11;
  asm volatile(
      "ds_read_b64_tr_b8 %0, %12\n\t"
      "ds_read_b64_tr_b8 %1, %12 offset:512\n\t"
      "ds_read_b64_tr_b8 %2, %12 offset:1024\n\t"
      "ds_read_b64_tr_b8 %3, %12 offset:1536\n\t"
      "ds_read_b64_tr_b8 %4, %13\n\t"
      "ds_read_b64_tr_b8 %5, %13 offset:512\n\t"
      "ds_read_b64_tr_b8 %6, %13 offset:1024\n\t"
      "ds_read_b64_tr_b8 %7, %13 offset:1536\n\t"
      "ds_read_b64_tr_b8 %8, %13 offset:2048\n\t"
      "ds_read_b64_tr_b8 %9, %13 offset:2560\n\t"
      "ds_read_b64_tr_b8 %10, %13 offset:3072\n\t"
      "ds_read_b64_tr_b8 %11, %13 offset:3584\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8), "=&v"(r9),
        "=&v"(r10), "=&v"(r11)
      : "v"(ax), "v"(az)
      : "memory");
  xb[0].v[0] = __builtin_bit_cast(long, r0); xb[0].v[1] = __builtin_bit_cast(long, r1);
  xb[1].v[0] = __builtin_bit_cast(long, r2); xb[1].v[1] = __builtin_bit_cast(long, r3);
  za[0].v[0] = __builtin_bit_cast(long, r4); za[0].v[1] = __builtin_bit_cast(long, r5);
  za[1].v[0] = __builtin_bit_cast(long, r6); za[1].v[1] = __builtin_bit_cast(long, r7);
  za[2].v[0] = __builtin_bit_cast(long, r8); za[2].v[1] = __builtin_bit_cast(long, r9);
  za[3].v[0] = __builtin_bit_cast(long, r10); za[3].v[1] = __builtin_bit_cast(long, r11);
}

HN_DEV void hn_wait_vmcnt(int n) {
  // s_waitcnt needs an immediate: wait until at most n of this wave's vector-memory ops are outstanding; counts without
  // a case of their own round DOWN (waiting for more than asked is always safe)
  if (n >= 32) { asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); return; }
  if (n >= 24) { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); return; }
  if (n >= 16) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); return; }
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;       // 12 .. 15
  }
}

// One workgroup (8 waves, two per SIMD) = one job = the whole dW tile grid (<= 8x8 tiles in bf16, 4x4 in
// fp32) of one Linear-layer input segment over a range of point blocks.  The dZ and X stash tiles are fetched
// ONCE by LDS-DMA into a ring of HN_WGRAD_STAGES stages (a stage = `bps` blocks, <= 8 x HN_WGRAD_MAXSLOT KiB: 2 x 64 KiB;
// counted vmcnt; one raw barrier per stage) and shared by the 8 waves, which own a gn x gk grid of tn x tk (<= 4x2)
// tile rectangles.  HBM-bound by construction: every stash byte is read exactly once.
struct HnDwBatchTable {
  HnDwBatch b[HN_MAX_WGRAD_BATCH];
  float unscale;          // HN_MODE_BF16_S8: 2^-dz_scale_log2, applied to every sum before it is added to the gradient
  const int32_t* order;   // optional: workgroup g runs job order[g] & 0xffffff of batch order[g] >> 24
  uint64_t* timeline;     // optional: the launch times itself (hn_common.h)
  int n;
};

// (Round 5 tried 2 and 4 extra LOADER waves per workgroup that do nothing but the ring's LDS-DMA, so that the compute waves
// go from the stage barrier straight into their products: 1.00 / 0.78 ms against 0.62 — three waves per SIMD at 168
// registers + 20 B of scratch, and two waves cannot issue a stage's 32 pieces in a stage's time.  Removed.)
template <bool BF16, bool S8 = false>
__global__ __launch_bounds__(512, 2) void hn_wgrad_kernel(const HnDwBatchTable tab) {
  hn_timeline_begin(tab.timeline);
  // which batch holds this workgroup's job (<= 8 scalar compares on kernel-argument data)
  int job_id = blockIdx.x, which = 0;
  if (tab.order != nullptr) {     // host-made global order (heaviest job first across ALL batches)
    const int o = __builtin_amdgcn_readfirstlane(tab.order[blockIdx.x]);
    which = o >> 24;
    job_id = o & 0xffffff;
  } else {
#pragma unroll
    for (int i = 0; i < HN_MAX_WGRAD_BATCH - 1; ++i)
      if (which == i && i + 1 < tab.n && job_id >= tab.b[i].n_jobs) { job_id -= tab.b[i].n_jobs; which = i + 1; }
  }
  const HnDwJob* jobs = tab.b[0].jobs;
  const char* stash = reinterpret_cast<const char*>(tab.b[0].stash);
  float* grads = tab.b[0].grads;
  float* partials = tab.b[0].partials;
  int n_jobs = tab.b[0].n_jobs;
#pragma unroll
  for (int i = 1; i < HN_MAX_WGRAD_BATCH; ++i)
    if (which == i) { jobs = tab.b[i].jobs; stash = reinterpret_cast<const char*>(tab.b[i].stash); grads = tab.b[i].grads; partials = tab.b[i].partials; n_jobs = tab.b[i].n_jobs; }
  using M = ModeT<BF16>;
  using Fr = std::conditional_t<S8, DwFrag8, DwFrag<BF16>>;
  constexpr int TU = S8 ? 1 : M::TILE_UNITS;
  constexpr int TBc = TU * 1024;
  constexpr size_t TB = TBc;
#ifndef HN_WGRAD_MAXSLOT
#define HN_WGRAD_MAXSLOT 8      /* LDS-DMA pieces per wave and stage: a stage holds <= 8 x MAXSLOT KiB; _lib.WGRAD_MAXSLOT mirrors it */
#endif
#ifndef HN_WGRAD_AUX
#define HN_WGRAD_AUX 2      /* nt: every stash byte is read once, do not keep it in L2 / MALL */
#endif
#ifndef HN_WGRAD_STAGES
#define HN_WGRAD_STAGES 2      /* ring depth of the bf16 / fp32 builds; _lib.WGRAD_STAGES mirrors it (A/B knob) */
#endif
  // The ring: 2 x 64 KiB since round 5 (rounds 2-5a: 4 x 32 KiB).  Same box, config 2, the launch in ms — 5 x 32: 0.637,
  // 4 x 32: 0.615-0.62, 3 x 32: 0.58-0.59, 3 x 48: 0.585, 2 x 32: 0.61, 2 x 64: 0.564, 2 x 80: 0.566 — a stage costs its
  // barrier, its counted wait and a burst of DMA issue whatever its size, and more than one stage in flight behind the one
  // being multiplied buys nothing (profiles/r05_wgrad_ring.log).  8-bit stash: 3 x 48 KiB.
  constexpr int STAGES = S8 ? 3 : HN_WGRAD_STAGES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (job_id >= n_jobs) {
    hn_timeline_end(tab.timeline);
    return;
  }
  const HnDwJob jb = jobs[job_id];
#ifdef HN_WGRAD_JOBTIMES   // diagnostic build (tools/wg_jobtimes.py): every job stamps the 100 MHz wall clock at its entry, when its
  // first stage has landed, behind its last product and behind its flush, with the CU it ran on — the per-CU timeline of
  // the launch (ramp, flush, the gap to the next workgroup on that CU, the idle tail), i.e. the ceiling of what a persistent
  // form of this kernel could recover.  Buffer from hn_set_wgrad_prof: 8 int64 per workgroup.
  long long* jt_buf = (long long*)tab.b[HN_MAX_WGRAD_BATCH - 1].jobs;
  const bool jt_on = jt_buf != nullptr && tab.n < HN_MAX_WGRAD_BATCH && threadIdx.x == 0;
  unsigned long long jt0 = 0, jt1 = 0, jt2 = 0;
  if (jt_on) jt0 = wall_clock64();
#endif
  const int c = lane & 31, h = lane >> 5;
  const int gn = jb.pad & 255, gk = (jb.pad >> 8) & 255, bps = (jb.pad >> 16) & 255;
  const int tn = (jb.n_nt + gn - 1) / gn, tk = (jb.n_kt + gk - 1) / gk;   // tiles per wave (<= 4, <= 2)
  const int wn = wave / gk, wk = wave % gk;
  const int n0 = wn * tn, k0 = wk * tk;
  const int my_n = wn < gn ? min(tn, jb.n_nt - n0) : 0, my_k = min(tk, jb.n_kt - k0);   // may be <= 0
  const int tiles_blk = jb.n_nt + jb.n_kt;
  const int UB = TU * tiles_blk;                 // 1-KiB units per block
  const size_t stage_bytes = (size_t)bps * UB * 1024;
  const int nb = jb.blk1 - jb.blk0;
  const int nstage = (nb + bps - 1) / bps;

  f32x16 acc[4][2];
  // bias gradient db[n] = sum_p dZ[p][n].  bf16 / fp32: a running sum per dZ tile on the vector pipe (bsum[i]: this
  // lane's feature over the points of its half, DwFrag::add_point_sum).  8-bit stash: ONE all-ones MFMA accumulator,
  // tile i in its column i (DwFrag8::mma_ones).
#ifndef HN_WGRAD_BIAS_MFMA
#define HN_WGRAD_BIAS_MFMA 0      /* 1: rounds 1-4, an all-ones MFMA accumulator per dZ tile (A/B knob) */
#endif
  constexpr bool BM = S8 || (HN_WGRAD_BIAS_MFMA != 0);
  constexpr int NB = S8 ? 1 : (BM ? 4 : 1);
  f32x16 accb[NB];
  float bsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[i % NB][e] = 0.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  }
  // bias gradient (db[n] = sum_p dZ[p][n], an all-ones operand): every wave of a row of the wave grid holds the same
  // dZ tiles, so the bias of tile i goes to the wave with wk == i % gk.  (All of it on the wk == 0 waves put 50 % more
  // MFMAs on two waves — which share a SIMD — and every stage barrier waited for them: in-kernel trace)
  unsigned bias_mask = 0;
  if (jb.b_off >= 0 && my_n > 0)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < my_n && (i % gk) == wk) bias_mask |= 1u << i;

  // Per-wave LDS-DMA slots of one stage, decoded ONCE (the integer divisions by run-time tile counts would
  // otherwise cost ~600 scalar instructions per stage): slot i moves 1 KiB from
  // sbase[i] + (first block of the stage) * sstride[i]  to  stage buffer + sdst[i], if sblk[i] < blocks in stage.
  constexpr int MAXSLOT = HN_WGRAD_MAXSLOT;                     // <= 32 units per stage / 8 waves... (fp32 tiles: 4 units each)
  // offsets in KiB (every slot offset, tile and unit is a multiple of 1 KiB; 32 bits reach 4 TiB): one register per
  // DMA slot instead of a 64-bit pair — the fp32 build sits at 256 registers, and a spilled address reloaded inside the
  // stage loop drains the ring (scratch loads share the in-order vmcnt queue)
  unsigned sbase[MAXSLOT];
  int sinfo[MAXSLOT];          // block index inside the stage << 2 | kind (0: X slot 1, 1: dZ, 2: X slot 2) ; huge = unused slot
  constexpr unsigned TK = (unsigned)(TBc / 1024);      // KiB per tile
  const unsigned zstride = (unsigned)jb.z_nt * TK, xstride = (unsigned)jb.x_nt * TK, x2stride = (unsigned)jb.x2_nt * TK;
  int per_stage = 0;
#pragma unroll
  for (int i = 0; i < MAXSLOT; ++i) {
    const int q = wave + 8 * i;
    sinfo[i] = 1 << 30;
    sbase[i] = 0;
    if (q < bps * UB) {
      const int bi = q / UB, r = q % UB;
      const int tile = r / TU, u = r % TU;
      const bool isz = tile < jb.n_nt;
      const int kt = tile - jb.n_nt;                  // k-tile: the first n_kt1 from X slot 1, the rest from X slot 2
      const bool isx2 = !isz && kt >= jb.n_kt1;
      const unsigned rel = isz ? ((unsigned)bi * jb.z_nt + jb.z_t0 + tile) * TK
                         : isx2 ? ((unsigned)bi * jb.x2_nt + jb.x2_t0 + (kt - jb.n_kt1)) * TK
                                : ((unsigned)bi * jb.x_nt + jb.x_t0 + kt) * TK;
      sbase[i] = (unsigned)((isz ? jb.z_off : (isx2 ? jb.x2_off : jb.x_off)) >> 10) + rel + (unsigned)u;
      sinfo[i] = bi << 2 | (isz ? 1 : (isx2 ? 2 : 0));
      ++per_stage;
    }
  }
  auto issue = [&](int s) {
    char* dst = smem + (size_t)(s % STAGES) * stage_bytes;
    const int b0 = jb.blk0 + s * bps;
    const int nblk_s = min(bps, jb.blk1 - b0);
#pragma unroll
    for (int i = 0; i < MAXSLOT; ++i) {
      if ((sinfo[i] >> 2) < nblk_s) {
        const unsigned kind = sinfo[i] & 3;
        const char* src = stash + ((unsigned long long)(sbase[i] + (unsigned)b0 * (kind == 1 ? zstride : (kind == 2 ? x2stride : xstride))) << 10);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + (wave + 8 * i) * 1024), 16, 0,
                                         HN_WGRAD_AUX);
      }
    }
  };

  // STAGES-1 stages in flight.  A wave issues `per_stage` LDS-DMA instructions for every FULL stage and `last_cnt`
  // (<= per_stage) for the last one, which may be partial.  "Stage s has landed" = at most as many instructions
  // outstanding as were issued AFTER it — counted exactly: allowing younger * per_stage while the partial stage is
  // among the younger ones would let per_stage - last_cnt loads of stage s itself still be in flight (a race that
  // small jobs, whose stages are all issued back to back, did hit: stale LDS in the product, flaky gradients).
  const int nb_last = nb - (nstage - 1) * bps;          // blocks of the last stage
  int last_cnt = 0;
#pragma unroll
  for (int i = 0; i < MAXSLOT; ++i)
    if ((sinfo[i] >> 2) < nb_last) ++last_cnt;
#ifdef HN_PROF   // diagnostic build: wave 0 of every 97th workgroup sums the cycles of its four phases per stage
  long long* prof_buf = (long long*)tab.b[HN_MAX_WGRAD_BATCH - 1].jobs;     // set by hn_set_wgrad_prof
  const bool prof_on = prof_buf != nullptr && tab.n < HN_MAX_WGRAD_BATCH && (blockIdx.x % 97) == 0 && wave == (HN_PROF - 1);     // HN_PROF = 1 + the wave to watch
  unsigned long long tw = 0, tb = 0, ti = 0, tc = 0, t0_ = 0, t1_ = 0, t2_ = 0, t3_ = 0, t4_ = 0;
#define HN_TS(x) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory")
#endif
  int tro0 = 0, tro1 = 0;       // bf16: this lane's transposed-read offsets inside a stash tile
  if constexpr (S8) tro0 = hn_dw8_offset(lane);
  else hn_dw_tr_offsets(lane, tro0, tro1);
  for (int s0 = 0; s0 < STAGES - 1 && s0 < nstage; ++s0) issue(s0);
  for (int s = 0; s < nstage; ++s) {
    const int younger = min(STAGES - 2, nstage - 1 - s);     // stages issued after stage s that may stay in flight
    // the last stage is among them exactly when fewer than STAGES-1 stages remain behind s
    const int allowed = (younger > 0 && nstage - 1 - s <= STAGES - 2) ? (younger - 1) * per_stage + last_cnt
                                                                      : younger * per_stage;
#ifdef HN_PROF
    if (prof_on) HN_TS(t0_);
#endif
    hn_wait_vmcnt(allowed);
#ifdef HN_PROF
    if (prof_on) HN_TS(t1_);
#endif
    // raw barrier: __syncthreads() would make hipcc drain vmcnt(0) and with it the stages still in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // ... for every wave; everyone finished stage s-1
    __builtin_amdgcn_sched_barrier(0);
#ifdef HN_WGRAD_JOBTIMES
    if (jt_on && s == 0) jt1 = wall_clock64();
#endif
#ifdef HN_PROF
    if (prof_on) HN_TS(t2_);
#endif
    // refill the buffer stage s-1 used.  (Measured and dropped, profiles/r05_wgrad_ring.log: the refill behind the stage's
    // products, for all waves or for the second wave of every SIMD only — no gain at any ring depth; a piece's address kept
    // in a register pair per slot instead of decoded from the slot tables — wave 0's issue share 34 -> 25 % of a stage, the
    // stage as long as before: a wave sits ~160-290 cycles on every 1-KiB piece whatever precedes it, the memory pipe's
    // back-pressure, not instruction count; the refill issued piece by piece behind the products of each dZ tile, slot
    // tables in register lanes read back with v_readlane, rings of 2-4 stages — 0.557-0.564 ms at 3 x 32 KiB against
    // 0.545-0.557 for this form; the whole refill issued by the second wave of every SIMD, 16 pieces each — what pays in
    // the forward / backward machines' weight stream (WStream::issue) — 0.585-0.594 against 0.553-0.564: every structure
    // lands on the same ~6 TB/s or worse.)
    if (s + STAGES - 1 < nstage) issue(s + STAGES - 1);
#ifdef HN_PROF
    if (prof_on) HN_TS(t3_);
#endif
    const char* st = smem + (size_t)(s % STAGES) * stage_bytes;
    const int nblk_s = min(bps, nb - s * bps);
    for (int bi = 0; bi < nblk_s; ++bi) {
      const char* sb = st + (size_t)bi * UB * 1024;
      Fr xb[2];
      // bf16: LDS byte addresses of this lane's transposed reads in the block's first dZ tile / first X tile of the wave
      const unsigned a_blk = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)sb;
      const unsigned ax = a_blk + (unsigned)((jb.n_nt + k0) * TB), az = a_blk + (unsigned)(n0 * TB);
      if constexpr (S8) {
        DwFrag8 za8[4];
        hn_tr8_block(xb, za8, ax + tro0, az + tro0);
        static_for4([&](auto I) __attribute__((always_inline)) {
          constexpr int i = decltype(I)::value;
          if (i < my_n) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (j < my_k) DwFrag8::mma(acc[i][j], za8[i], xb[j]);
            if (bias_mask & (1u << i)) DwFrag8::mma_ones(accb[0], za8[i], i, lane);
          }
        });
        continue;
      }
#ifndef HN_WGRAD_BLOCK
#define HN_WGRAD_BLOCK 1      /* 0: rounds 1-5a, one read-wait-multiply round trip per operand tile (A/B knob) */
#endif
      if constexpr (BF16 && !S8 && HN_WGRAD_BLOCK != 0) {
        // every operand tile of the block under one LDS wait (hn_tr_block), then the products back to back
        DwFrag<true> za[4];
        bf16x8 xv[2][2], zv[4][2];      // (a wave with <= 2 dZ tiles reads two; tiles 2, 3 are then never multiplied: i < my_n below)
        if (my_n > 2) hn_tr_block<4>(xv, zv, ax + tro0, ax + tro1, az + tro0, az + tro1);
        else hn_tr_block<2>(xv, zv, ax + tro0, ax + tro1, az + tro0, az + tro1);
#pragma unroll
        for (int j = 0; j < 2; ++j) { xb[j].v[0] = xv[j][0]; xb[j].v[1] = xv[j][1]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { za[i].v[0] = zv[i][0]; za[i].v[1] = zv[i][1]; }
        static_for4([&](auto I) __attribute__((always_inline)) {
          constexpr int i = decltype(I)::value;
          if (i < my_n) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (j < my_k) Fr::mma(acc[i][j], za[i], xb[j]);
            if (bias_mask & (1u << i)) {
              if constexpr (BM) Fr::mma_ones(accb[i % NB], za[i], i, lane);
              else bsum[i] = za[i].add_point_sum(bsum[i]);
            }
          }
        });
        continue;
      }
      if constexpr (BF16) {
        if (0 < my_k) xb[0].template load_tr<0>(ax + tro0, ax + tro1);
        if (1 < my_k) xb[1].template load_tr<TBc>(ax + tro0, ax + tro1);
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (j < my_k) xb[j].load(sb + (size_t)(jb.n_nt + k0 + j) * TB, lane, 0, 0);
      }
      static_for4([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        if (i < my_n) {
          Fr za;
          if constexpr (BF16) za.template load_tr<TBc * i>(az + tro0, az + tro1);
          else za.load(sb + (size_t)(n0 + i) * TB, lane, 0, 0);
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (j < my_k) Fr::mma(acc[i][j], za, xb[j]);
          if (bias_mask & (1u << i)) {
            if constexpr (BM) Fr::mma_ones(accb[i % NB], za, i, lane);
            else bsum[i] = za.add_point_sum(bsum[i]);
          }
        }
      });
    }
#ifdef HN_PROF
    if (prof_on) { HN_TS(t4_); tw += t1_ - t0_; tb += t2_ - t1_; ti += t3_ - t2_; tc += t4_ - t3_; }
#endif
  }
#ifdef HN_WGRAD_JOBTIMES
  if (jt_on) jt2 = wall_clock64();
#endif
#ifdef HN_PROF
  if (prof_on && lane == 0) {
    long long* o = prof_buf + (blockIdx.x / 97) * 8;
    o[0] = (long long)tw; o[1] = (long long)tb; o[2] = (long long)ti; o[3] = (long long)tc; o[4] = nstage;
    o[5] = jb.n_nt * 16 + jb.n_kt; o[6] = bps; o[7] = blockIdx.x;
  }
#endif
#if defined(HN_WGRAD_EXP) && (HN_WGRAD_EXP == 1 || HN_WGRAD_EXP == 3)   // timing-only experiments (results are WRONG):
  {                                                                     // 1 = no flush at all, 3 = no dW flush
    float z = 0.0f;                                                     // (every accumulator stays live)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) z += acc[i][j][q];
#if HN_WGRAD_EXP == 1
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) z += accb[i][q] + bsum[q & 3];
#endif
    if (z == 123.456f) grads[0] = z;
  }
#if HN_WGRAD_EXP == 1
  hn_timeline_end(tab.timeline);
  return;
#endif
#endif
  // D[n][k]: lane = column k (c), register q -> row rho(q,h)
#if defined(HN_WGRAD_EXP) && HN_WGRAD_EXP == 3      // timing-only experiment: no dW flush (weight gradients are WRONG)
  if (false) {
#else
  if (jb.w_off >= 0 && partials != nullptr) {
#endif
    // the rectangle leaves as raw accumulator tiles, 16 coalesced 256-B stores per tile (hn_mlp_wgrad_reduce sums the
    // jobs' slabs and adds every element to the gradient once): a CU retires float atomics at ~5 GB/s, plain stores
    // at its full store rate — the atomic flush was 100 us of a 650-us launch at config 2
    float* P = partials + (size_t)jb.p_tile * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < my_n)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (j < my_k) {
            // tile = [register quad (4)][lane (64)][4 floats]: four 1-KiB store instructions per tile and wave
            float* T = P + (size_t)((n0 + i) * jb.n_kt + (k0 + j)) * 1024 + lane * 4;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              f32x4 v = {acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]};
              if (S8) v *= tab.unscale;
              *reinterpret_cast<f32x4*>(T + q4 * 256) = v;
            }
          }
#if defined(HN_WGRAD_EXP) && HN_WGRAD_EXP == 3
  } else if (false) {
#else
  } else if (jb.w_off >= 0) {
#endif
    float* G = grads + jb.w_off;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < my_n)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (j < my_k)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const int row = jb.r0 + 32 * (n0 + i) + (S8 ? hn_dw8_feature(hn_rho(q, h)) : hn_rho(q, h));
              const int col = jb.c0 + 32 * (k0 + j) + (S8 ? hn_dw8_feature(c) : c);
              if (row >= 0 && col >= 0 && row < jb.r_end && col < jb.c_end)
                atomicAdd(G + (size_t)row * jb.ld + col, S8 ? acc[i][j][q] * tab.unscale : acc[i][j][q]);
            }
  }
#if defined(HN_WGRAD_EXP) && HN_WGRAD_EXP == 2      // timing-only experiment: no bias flush (bias gradients are WRONG)
  bias_mask = 0;
#endif
  if (!BM && bias_mask != 0) {
    // the two halves of the wave hold the sums over the two halves of every block's points: add them, lanes 0-31 hold
    // db of row (lane) of each dZ tile
    float* Bp = partials != nullptr ? partials + (size_t)(jb.p_tile + jb.n_nt * jb.n_kt) * 1024 : nullptr;
    float* gb = grads + jb.b_off;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (bias_mask & (1u << i)) {
        const float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);
        const int row = jb.r0 + 32 * (n0 + i) + c;
        if (h == 0) {
          // bias partials: one more slab tile behind the job's dW tiles, 32 floats (rows in natural order) per dZ tile
          if (Bp != nullptr) Bp[(n0 + i) * 32 + c] = v;
          else if (row >= 0 && row < jb.r_end) atomicAdd(gb + row, v);
        }
      }
  } else if (bias_mask != 0 && c < 4) {
    float* gb = grads + jb.b_off;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if ((bias_mask & (1u << i)) && c == (S8 ? i : 0))
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = jb.r0 + 32 * (n0 + i) + (S8 ? hn_dw8_feature(hn_rho(q, h)) : hn_rho(q, h));
          if (row >= 0 && row < jb.r_end) atomicAdd(gb + row, S8 ? accb[i % NB][q] * tab.unscale : accb[i % NB][q]);
        }
  }
#ifdef HN_WGRAD_JOBTIMES
  __syncthreads();      // every wave's stores are issued
  if (jt_on) {
    unsigned hw = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long* o = jt_buf + (size_t)blockIdx.x * 8;
    o[0] = (long long)hw | ((long long)(xcc & 15u) << 32);
    o[1] = (long long)jt0; o[2] = (long long)jt1; o[3] = (long long)jt2; o[4] = (long long)wall_clock64();
    o[5] = (long long)nstage * (long long)stage_bytes; o[6] = jb.n_nt * 16 + jb.n_kt; o[7] = 1;
  }
#endif
  hn_timeline_end(tab.timeline);
}

#ifdef HN_WGRAD_PERSIST
// ------------------------------------------------------------------------------------------------
// EXPERIMENT BUILD ONLY (-DHN_WGRAD_PERSIST=1; round 6, verdict item 3 — measured and CLOSED, profiles/r06_wgrad_persistent.md):
// a persistent form of the weight-gradient launch.  It does what it was built for — per job, ramp 4.1 -> 1.5 us, workgroup
// turnover 12.8 -> 0 us, 5.8 % of launch x CUs of idle time gone (tools/wg_jobtimes.py) — and the launch is no faster
// (0.575 against 0.550 ms, same box, alternating): the stream is bound by the AGGREGATE HBM rate, a CU that pauses between
// two jobs leaves its share to the others, and hiding the pause only stretches every job's own streaming time (0.855 ->
// 0.907 of launch x CUs).  What is really lost is the idle tail behind each CU's last job (7-8 %), which this form does
// not touch.  Not compiled into the product library.
// The same weight-gradient jobs, PERSISTENT form (round 6; bf16 stash, 2-stage ring, block reads): one workgroup per CU
// walks the host-ordered job list through a device ticket.  tools/wg_jobtimes.py (profiles/r06_wgrad_jobtimes.log) had put
// numbers on what one-workgroup-per-job leaves on the table at config 2: 4.1 us of ramp per job (entry -> first stage
// landed), 1.1 us of flush, and 12.8 us between a job's exit and the next workgroup's entry on the same CU (dispatch of a
// 512-thread / 128-KiB workgroup + two dependent descriptor loads) = 7.2 % of launch x CUs.  Here the ring runs ACROSS
// jobs: while the last stage of job j is being multiplied, the first stage of job j + 1 — claimed and its descriptor
// loaded one job ahead — is already on its way into the other buffer (the slot tables of job j are dead once its last
// stage has been issued, so they are re-decoded in place: no second set of registers), and the slab stores of job j drain
// under it.  Same products, same slabs, same reduce: results bit-identical to hn_wgrad_kernel<true>.
// ------------------------------------------------------------------------------------------------
struct HnDwCur {           // what a job's stages and flush need of its descriptor + geometry: wave-uniform, 27 scalar registers
  int bps, n0, k0, my_n, my_k, UB, nb, nstage, blk0, blk1, n_nt, n_kt;
  int w_off, p_tile, b_off, r0, c0, r_end, c_end, ld;
  unsigned zstride, xstride, x2stride, bias_mask;
  const char* stash; float* grads; float* partials;
};
__global__ __launch_bounds__(512, 2) void hn_wgrad_persist_kernel(const HnDwBatchTable tab, unsigned* __restrict__ ticket,
                                                                  int total) {
  hn_timeline_begin(tab.timeline);
  using Fr = DwFrag<true>;
  constexpr int TU = ModeT<true>::TILE_UNITS;
  constexpr int TBc = TU * 1024;
  constexpr size_t TB = TBc;
  constexpr int MAXSLOT = HN_WGRAD_MAXSLOT;
  constexpr unsigned BUF = 8u * MAXSLOT * 1024u;      // bytes per ring buffer: fixed (jobs differ in their stage size)
  constexpr unsigned TK = (unsigned)(TBc / 1024);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_next = reinterpret_cast<int*>(smem + 2 * BUF);      // two words behind the ring (dynamic LDS: 2 x BUF + 64)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int G = (int)gridDim.x;
  int tro0 = 0, tro1 = 0;
  hn_dw_tr_offsets(lane, tro0, tro1);

  // per-wave LDS-DMA slots of one stage (hn_wgrad_kernel's tables)
  unsigned sbase[MAXSLOT];
  int sinfo[MAXSLOT];
  // job `g` of the launch's global order: its descriptor is read, boiled down to `q` (geometry of this wave, flush
  // destination, batch pointers) and to the slot tables — the descriptor itself is dead afterwards
  auto prepare = [&](int g, HnDwCur& q) __attribute__((always_inline)) {
    int job_id = g, which = 0;
    if (tab.order != nullptr) {
      const int o = __builtin_amdgcn_readfirstlane(tab.order[g]);
      which = o >> 24;
      job_id = o & 0xffffff;
    } else {
#pragma unroll
      for (int i = 0; i < HN_MAX_WGRAD_BATCH - 1; ++i)
        if (which == i && i + 1 < tab.n && job_id >= tab.b[i].n_jobs) { job_id -= tab.b[i].n_jobs; which = i + 1; }
    }
    const HnDwJob* jobs = tab.b[0].jobs;
    q.stash = reinterpret_cast<const char*>(tab.b[0].stash); q.grads = tab.b[0].grads; q.partials = tab.b[0].partials;
#pragma unroll
    for (int i = 1; i < HN_MAX_WGRAD_BATCH; ++i)
      if (which == i) { jobs = tab.b[i].jobs; q.stash = reinterpret_cast<const char*>(tab.b[i].stash); q.grads = tab.b[i].grads; q.partials = tab.b[i].partials; }
    const HnDwJob jd = jobs[job_id];
    const int gn = jd.pad & 255, gk = (jd.pad >> 8) & 255;
    q.bps = (jd.pad >> 16) & 255;
    const int tn = (jd.n_nt + gn - 1) / gn, tk = (jd.n_kt + gk - 1) / gk;
    const int wn = wave / gk, wk = wave % gk;
    q.n0 = wn * tn; q.k0 = wk * tk;
    q.my_n = wn < gn ? min(tn, jd.n_nt - q.n0) : 0;
    q.my_k = min(tk, jd.n_kt - q.k0);
    q.UB = TU * (jd.n_nt + jd.n_kt);
    q.nb = jd.blk1 - jd.blk0;
    q.nstage = (q.nb + q.bps - 1) / q.bps;
    q.blk0 = jd.blk0; q.blk1 = jd.blk1; q.n_nt = jd.n_nt; q.n_kt = jd.n_kt;
    q.w_off = jd.w_off; q.p_tile = jd.p_tile; q.b_off = jd.b_off; q.r0 = jd.r0; q.c0 = jd.c0; q.r_end = jd.r_end;
    q.c_end = jd.c_end; q.ld = jd.ld;
    q.zstride = (unsigned)jd.z_nt * TK; q.xstride = (unsigned)jd.x_nt * TK; q.x2stride = (unsigned)jd.x2_nt * TK;
    q.bias_mask = 0;
    if (jd.b_off >= 0 && q.my_n > 0)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < q.my_n && (i % gk) == wk) q.bias_mask |= 1u << i;
#pragma unroll
    for (int i = 0; i < MAXSLOT; ++i) {
      const int qq = wave + 8 * i;
      sinfo[i] = 1 << 30;
      sbase[i] = 0;
      if (qq < q.bps * q.UB) {
        const int bi = qq / q.UB, r = qq % q.UB;
        const int tile = r / TU, u = r % TU;
        const bool isz = tile < jd.n_nt;
        const int kt = tile - jd.n_nt;
        const bool isx2 = !isz && kt >= jd.n_kt1;
        const unsigned rel = isz ? ((unsigned)bi * jd.z_nt + jd.z_t0 + tile) * TK
                           : isx2 ? ((unsigned)bi * jd.x2_nt + jd.x2_t0 + (kt - jd.n_kt1)) * TK
                                  : ((unsigned)bi * jd.x_nt + jd.x_t0 + kt) * TK;
        sbase[i] = (unsigned)((isz ? jd.z_off : (isx2 ? jd.x2_off : jd.x_off)) >> 10) + rel + (unsigned)u;
        sinfo[i] = bi << 2 | (isz ? 1 : (isx2 ? 2 : 0));
      }
    }
  };
  // stage `s` of job `q` (whose tables are the decoded ones) into ring buffer `buf`
  auto issue = [&](const HnDwCur& q, int s, int buf) __attribute__((always_inline)) {
    char* dst = smem + (size_t)buf * BUF;
    const int b0 = q.blk0 + s * q.bps;
    const int nblk_s = min(q.bps, q.blk1 - b0);
#pragma unroll
    for (int i = 0; i < MAXSLOT; ++i) {
      if ((sinfo[i] >> 2) < nblk_s) {
        const unsigned kind = sinfo[i] & 3;
        const char* src = q.stash + ((unsigned long long)(sbase[i] + (unsigned)b0 * (kind == 1 ? q.zstride : (kind == 2 ? q.x2stride : q.xstride))) << 10);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + (wave + 8 * i) * 1024), 16, 0,
                                         HN_WGRAD_AUX);
      }
    }
  };

  if ((int)blockIdx.x >= total) {       // (the host never launches more workgroups than jobs)
    hn_timeline_end(tab.timeline);
    return;
  }
  HnDwCur cj, nj;                       // current job, next job
  prepare((int)blockIdx.x, cj);         // this workgroup's first job: one per workgroup, the tickets hand out the rest
  int rb = 0;                           // ring buffer of the current job's stage 0
  issue(cj, 0, rb);
  int parity = 0;
#ifdef HN_WGRAD_JOBTIMES
  long long* jt_buf = (long long*)tab.b[HN_MAX_WGRAD_BATCH - 1].jobs;
  const bool jt_on = jt_buf != nullptr && tab.n < HN_MAX_WGRAD_BATCH && threadIdx.x == 0;
  int jt_idx = (int)blockIdx.x;
  unsigned long long jt_prev_end = wall_clock64();
#endif
  for (;;) {
#ifdef HN_WGRAD_JOBTIMES
    unsigned long long jt0 = jt_prev_end, jt1 = 0, jt2 = 0;
#endif
    // The job after this one is claimed LATE — one stage before its descriptor is needed (stage nstage - 3's products hide
    // the atomic, stage nstage - 2 publishes it through LDS) —: a claim at the start of the job turned the ticket into a
    // round-robin assignment (every workgroup holds two jobs, a third of the list is handed out blind) and the launch ran
    // 12 % LONGER than one workgroup per job (profiles/r06_wgrad_persist_ab.log).
    const int s_pub = max(0, cj.nstage - 2);
    unsigned tk_raw = 0;
    if (s_pub == 0 && threadIdx.x == 0) tk_raw = atomicAdd(ticket, 1u);
    int nxt = total;
    bool prepared = false;
    f32x16 acc[4][2];
    float bsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    for (int s = 0; s < cj.nstage; ++s) {
      // stage s has landed: everything this wave issued is older than it or it (2-stage ring: nothing younger in flight)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (s == s_pub && threadIdx.x == 0) s_next[parity] = G + (int)tk_raw;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#ifdef HN_WGRAD_JOBTIMES
      if (jt_on && s == 0) jt1 = wall_clock64();
#endif
      if (s == s_pub) nxt = __builtin_amdgcn_readfirstlane(s_next[parity]);
      if (s + 1 == s_pub && threadIdx.x == 0) tk_raw = atomicAdd(ticket, 1u);
      if (s + 1 < cj.nstage) {
        issue(cj, s + 1, (rb + s + 1) & 1);
      }
      if (s + 2 >= cj.nstage && !prepared && nxt < total) {
        // the current job's slot tables are dead — its last stage has been issued (or is the one in LDS): the next job's
        // descriptor is read and decoded in place, one stage before it is needed where the job has two or more
        prepare(nxt, nj);
        prepared = true;
      }
      if (s + 1 >= cj.nstage && prepared) issue(nj, 0, (rb + s + 1) & 1);      // into the buffer stage s - 1 has just released
      const char* st = smem + (size_t)((rb + s) & 1) * BUF;
      const int nblk_s = min(cj.bps, cj.nb - s * cj.bps);
      for (int bi = 0; bi < nblk_s; ++bi) {
        const char* sb = st + (size_t)bi * cj.UB * 1024;
        const unsigned a_blk = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)sb;
        const unsigned ax = a_blk + (unsigned)((cj.n_nt + cj.k0) * TB), az = a_blk + (unsigned)(cj.n0 * TB);
        Fr xb[2], za[4];
        bf16x8 xv[2][2], zv[4][2];
        if (cj.my_n > 2) hn_tr_block<4>(xv, zv, ax + tro0, ax + tro1, az + tro0, az + tro1);
        else hn_tr_block<2>(xv, zv, ax + tro0, ax + tro1, az + tro0, az + tro1);
#pragma unroll
        for (int j = 0; j < 2; ++j) { xb[j].v[0] = xv[j][0]; xb[j].v[1] = xv[j][1]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { za[i].v[0] = zv[i][0]; za[i].v[1] = zv[i][1]; }
        static_for4([&](auto I) __attribute__((always_inline)) {
          constexpr int i = decltype(I)::value;
          if (i < cj.my_n) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (j < cj.my_k) Fr::mma(acc[i][j], za[i], xb[j]);
            if (cj.bias_mask & (1u << i)) bsum[i] = za[i].add_point_sum(bsum[i]);
          }
        });
      }
    }
#ifdef HN_WGRAD_JOBTIMES
    if (jt_on) jt2 = wall_clock64();
#endif
    // ---- flush of the current job (hn_wgrad_kernel's: partial slabs or float atomics) ----
    if (cj.w_off >= 0 && cj.partials != nullptr) {
      float* P = cj.partials + (size_t)cj.p_tile * 1024;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < cj.my_n)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (j < cj.my_k) {
              float* T = P + (size_t)((cj.n0 + i) * cj.n_kt + (cj.k0 + j)) * 1024 + lane * 4;
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 v = {acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]};
                *reinterpret_cast<f32x4*>(T + q4 * 256) = v;
              }
            }
    } else if (cj.w_off >= 0) {
      float* Gw = cj.grads + cj.w_off;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < cj.my_n)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (j < cj.my_k)
#pragma unroll
              for (int q = 0; q < 16; ++q) {
                const int row = cj.r0 + 32 * (cj.n0 + i) + hn_rho(q, h);
                const int col = cj.c0 + 32 * (cj.k0 + j) + c;
                if (row >= 0 && col >= 0 && row < cj.r_end && col < cj.c_end) atomicAdd(Gw + (size_t)row * cj.ld + col, acc[i][j][q]);
              }
    }
    if (cj.bias_mask != 0) {
      float* Bp = cj.partials != nullptr ? cj.partials + (size_t)(cj.p_tile + cj.n_nt * cj.n_kt) * 1024 : nullptr;
      float* gb = cj.grads + cj.b_off;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (cj.bias_mask & (1u << i)) {
          const float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);
          const int row = cj.r0 + 32 * (cj.n0 + i) + c;
          if (h == 0) {
            if (Bp != nullptr) Bp[(cj.n0 + i) * 32 + c] = v;
            else if (row >= 0 && row < cj.r_end) atomicAdd(gb + row, v);
          }
        }
    }
#ifdef HN_WGRAD_JOBTIMES
    if (jt_on) {
      unsigned hw = 0, xcc = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      long long* o = jt_buf + (size_t)jt_idx * 8;
      jt_prev_end = wall_clock64();
      o[0] = (long long)hw | ((long long)(xcc & 15u) << 32);
      o[1] = (long long)jt0; o[2] = (long long)jt1; o[3] = (long long)jt2; o[4] = (long long)jt_prev_end;
      o[5] = (long long)cj.nb * cj.UB * 1024; o[6] = cj.n_nt * 16 + cj.n_kt; o[7] = 1;
    }
    jt_idx = nxt;
#endif
    if (!prepared) break;
    rb = (rb + cj.nstage) & 1;
    cj = nj;
    parity ^= 1;
  }
  // the workgroup that leaves last re-arms the tickets for the next launch
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(ticket + 1, 1u) == (unsigned)G - 1) {
      ticket[0] = 0u;
      ticket[1] = 0u;
    }
  }
  hn_timeline_end(tab.timeline);
}

#endif  // HN_WGRAD_PERSIST

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static int hn_grid_for(int n_points, int pts_per_wg) {
  int tiles = (n_points + pts_per_wg - 1) / pts_per_wg;
  const int cap = 256 * 4;  // persistent: at most 4 tiles' worth of workgroups per CU queue
  return tiles < cap ? tiles : cap;
}

extern "C" int hn_version(void) { return HN_VERSION; }

extern "C" int hn_abi_sizes(int32_t* out, int n) {
  const int32_t v[8] = {(int32_t)sizeof(HnMlpArgs),      (int32_t)sizeof(HnPackUnit), (int32_t)sizeof(HnPackBias),
                        (int32_t)sizeof(HnDwJob),        (int32_t)sizeof(HnCompositeArgs), (int32_t)sizeof(HnFeat),
                        (int32_t)sizeof(HnSlot),         (int32_t)sizeof(HnSrc)};
  static_assert(sizeof(HnDwBatch) == 40, "HnDwBatch layout");
  static_assert(sizeof(HnSrc) == 32, "HnSrc layout");
  for (int i = 0; i < n && i < 8; ++i) out[i] = v[i];
  return 8;
}

// Build-time tuning knobs this library was compiled with, in the order HN_BUILD_CONFIG_* of include/hn_kernels.h.  The
// host mirrors (hypernerf_torch_amd/_lib.py) are READ from here at load time: a library prebuilt with other knobs (the
// A/B tools) can no longer run under host tables cut for the defaults.
#define HN_STR2(x) #x
#define HN_STR(x) HN_STR2(x)
// the same values as text inside the binary: a host that must know them BEFORE it may dlopen the library (Python's import
// of the package, which a rebuild can follow) finds them by scanning the file for the marker
extern "C" __attribute__((used)) const char hn_build_config_text[] =
    "HN_BUILD_CONFIG:" HN_STR(HN_WGRAD_STAGES) "," HN_STR(HN_WGRAD_MAXSLOT) "," HN_STR(HN_WGRAD_BIAS_MFMA) "," HN_STR(HN_CHUNK_UNITS) ","
    HN_STR(HN_WGRAD_BLOCK) "," HN_STR(HN_WSTREAM_ASYM) "," HN_STR(HN_BF16_WAVES) "," HN_STR(HN_WGRAD_AUX) ";";
extern "C" int hn_build_config(int32_t* out, int n) {
  const int32_t v[HN_BUILD_CONFIG_N] = {HN_WGRAD_STAGES, HN_WGRAD_MAXSLOT, HN_WGRAD_BIAS_MFMA, HN_CHUNK_UNITS,
                                        HN_WGRAD_BLOCK,  HN_WSTREAM_ASYM,  HN_BF16_WAVES,      HN_WGRAD_AUX};
  for (int i = 0; i < n && i < HN_BUILD_CONFIG_N; ++i) out[i] = v[i];
  return HN_BUILD_CONFIG_N;
}

static void hn_allow_big_lds() {
  // per device: hipFuncSetAttribute applies to the device that is current at the call (a process driving several GPUs
  // must raise the limit on each of them before its first launch there)
  static bool done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (done[dev]) return;
  done[dev] = true;
  const int big = 160 * 1024;
#define HN_BIG(k) (void)hipFuncSetAttribute((const void*)(k), hipFuncAttributeMaxDynamicSharedMemorySize, big)
#ifdef HN_WGRAD_PERSIST
  HN_BIG(hn_wgrad_persist_kernel);
#endif
  HN_BIG((hn_mlp_fwd_kernel<true, 2, false, true>)); HN_BIG((hn_mlp_fwd_kernel<true, 2, false, false>));
  HN_BIG((hn_mlp_fwd_kernel<true, 2, true, true>)); HN_BIG((hn_mlp_fwd_kernel<true, 2, true, false>));
  HN_BIG((hn_mlp_fwd_kernel<true, 3, false, true>)); HN_BIG((hn_mlp_fwd_kernel<true, 3, false, false>));
  HN_BIG((hn_mlp_fwd_kernel<true, 3, true, true>)); HN_BIG((hn_mlp_fwd_kernel<true, 3, true, false>));
  HN_BIG((hn_mlp_fwd_kernel<false, 3, true, true>)); HN_BIG((hn_mlp_fwd_kernel<false, 3, true, false>));
  HN_BIG((hn_mlp_fwd_kernel<true, 2, false, true, true>)); HN_BIG((hn_mlp_fwd_kernel<true, 3, false, true, true>));
  HN_BIG((hn_mlp_bwd_kernel<true, false, true>)); HN_BIG((hn_wgrad_kernel<true, true>));
#undef HN_BIG
  (void)hipFuncSetAttribute((const void*)hn_mlp_bwd_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  (void)hipFuncSetAttribute((const void*)hn_mlp_bwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  (void)hipFuncSetAttribute((const void*)hn_mlp_bwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  (void)hipFuncSetAttribute((const void*)hn_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  (void)hipFuncSetAttribute((const void*)hn_wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
}

extern "C" int hn_pack_units(int mode, const HnPackUnit* units, int n_units, const float* const* ptrs, void* wstream,
                             const HnPackBias* bias, int n_bias, float* bias_out, hnStream_t stream) {
  if (n_units < 0 || n_bias < 0) return -1;
  const int total = n_units + n_bias;
  if (total == 0) return 0;
  const int grid = (total + 3) / 4;
  if (mode == HN_MODE_BF16 || mode == HN_MODE_BF16_S8)
    hipLaunchKernelGGL(hn_pack_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, units, n_units, ptrs,
                       (char*)wstream, bias, n_bias, bias_out);
  else if (mode == HN_MODE_F32)
    hipLaunchKernelGGL(hn_pack_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, units, n_units, ptrs,
                       (char*)wstream, bias, n_bias, bias_out);
  else
    return -2;
  HN_CHECK_LAUNCH();
  return 0;
}

extern "C" int hn_pack_units_multi(int mode, const HnPackJob* jobs, int n_jobs, hnStream_t stream) {
  HnPackTable tab = {};
  int blocks = 0;
  const int rc = hn_pack_table_fill(jobs, n_jobs, tab, blocks);
  if (rc != 0) return rc;
  if (blocks == 0) return 0;
  if (mode == HN_MODE_BF16 || mode == HN_MODE_BF16_S8)
    hipLaunchKernelGGL(hn_pack_multi_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tab);
  else if (mode == HN_MODE_F32)
    hipLaunchKernelGGL(hn_pack_multi_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tab);
  else
    return -2;
  HN_CHECK_LAUNCH();
  return 0;
}

static int hn_check_args(const HnMlpArgs* a) {
  if (a == nullptr) return -1;
  if (a->n_points <= 0 || a->samples_per_ray <= 0 || a->n_ops <= 0 || a->n_chunks <= 0) return -2;
  if (a->ops == nullptr || a->wstream == nullptr) return -3;
  if (a->mode != HN_MODE_BF16 && a->mode != HN_MODE_F32 && a->mode != HN_MODE_BF16_S8) return -4;
  // the 8-bit stash exists in the render-level training builds only (no stand-alone-module paths)
  if (a->mode == HN_MODE_BF16_S8 && a->training && a->wide_ops != 0) return -4;
  if (a->mode == HN_MODE_BF16_S8 && (a->dz_scale_log2 < -60 || a->dz_scale_log2 > 60)) return -5;
  if (a->n_dsrc < 0 || a->n_dsrc > HN_DSRC_COMPS) return -5;
  if (a->n_bias < 0 || a->n_feat < 0 || a->n_comps < 0 || a->n_comps > HN_MAX_COMPS) return -5;
  if (a->max_groups < 0 || a->max_groups > 3) return -5;
  if (a->n_comps > 0 && a->comps == nullptr) return -3;
  return 0;
}

// Host-only: the extent of `stash` and `masks` an op program touches for n_points points (the kernels do no
// bounds checks of their own; a host that did not compile the program itself sizes its buffers with this).
extern "C" int hn_mlp_workspace_bytes(const int32_t* ops_host, int n_ops, int backward, int mode, int64_t n_points,
                                      int64_t* stash_bytes, int64_t* mask_bytes) {
  if (ops_host == nullptr || stash_bytes == nullptr || mask_bytes == nullptr) return -1;
  if (n_ops <= 0 || n_points <= 0) return -2;
  if (mode != HN_MODE_BF16 && mode != HN_MODE_F32 && mode != HN_MODE_BF16_S8) return -4;
  const int64_t nblk = (n_points + 31) / 32;
  const int64_t tile = (mode == HN_MODE_BF16_S8 ? 1 : mode == HN_MODE_BF16 ? ModeT<true>::TILE_UNITS : ModeT<false>::TILE_UNITS) * 1024;
  int64_t sb = 0, mb = 0;
  auto stash = [&](int off_kib, int nt) {
    if (off_kib >= 0) sb = std::max<int64_t>(sb, (int64_t)(unsigned)off_kib * 1024 + nblk * nt * tile);
  };
  auto mask = [&](int off256, int nt_tiles) {
    if (off256 >= 0) mb = std::max<int64_t>(mb, (int64_t)(unsigned)off256 * 256 + nblk * ((nt_tiles + 1) >> 1) * 256);
  };
  for (int i = 0; i < n_ops; ++i) {
    const int32_t* w = ops_host + (size_t)i * HN_OP_WORDS;
    if (!backward) {
      if (w[0] == HN_OP_LAYER) {
        const int nG = (w[1] >> 8) & 255, NT = (w[1] >> 16) & 255;
        mask(w[4], NT);
        stash(w[5], NT);
        stash(w[6], 2 * nG);
      } else if (w[0] != HN_OP_OUT && w[0] != HN_OP_OUT_WIDE) {
        return -7;
      }
    } else {
      if (w[0] == HN_BOP_LOAD) {
        stash(w[7], 1);
      } else if (w[0] == HN_BOP_LOAD_WIDE) {
        mask(w[5], w[4]);
        stash(w[7], w[4]);
      } else if (w[0] == HN_BOP_LAYER) {
        const int NT = (w[1] >> 16) & 255;
        mask(w[4], NT);
        stash(w[5], NT);
      } else if (w[0] != HN_BOP_AUX) {
        return -7;
      }
    }
  }
  *stash_bytes = sb;
  *mask_bytes = mb;
  return 0;
}

extern "C" int hn_mlp_forward(const HnMlpArgs* a, hnStream_t stream) {
  int rc = hn_check_args(a);
  if (rc) return rc;
  hn_allow_big_lds();
  constexpr int WB = ModeT<true>::WAVES;
  if (a->n_trig_comps < 0 || a->n_trig_comps > a->n_comps) return -5;
  const bool bf = a->mode == HN_MODE_BF16 || a->mode == HN_MODE_BF16_S8;
  const size_t planes = bf
                            ? (size_t)a->n_comps + (size_t)a->n_trig_comps + (a->trig_lo_planes ? (size_t)a->n_trig_comps : 1)
                            : (size_t)a->n_comps;
  const size_t lds = 2 * HN_CHUNK_UNITS * 1024 + (size_t)((a->n_bias + 3) & ~3) * 4 +
                     (size_t)((a->n_feat + 1) & ~1) * 8 + (bf ? (size_t)a->n_feat * 16 : 0) +
                     (size_t)8 * planes * 32 * 4;
  if (lds > 158 * 1024) return -6;
  const dim3 grid_b(hn_grid_for(a->n_points, WB * 32)), blk_b(WB * 64);
  const hipStream_t st = (hipStream_t)stream;
#define HN_FWD(G, E, T) hipLaunchKernelGGL((hn_mlp_fwd_kernel<true, G, E, T>), grid_b, blk_b, lds, st, *a)
  if (a->mode == HN_MODE_BF16_S8 && a->training) {
    if (a->max_groups <= 2) hipLaunchKernelGGL((hn_mlp_fwd_kernel<true, 2, false, true, true>), grid_b, blk_b, lds, st, *a);
    else hipLaunchKernelGGL((hn_mlp_fwd_kernel<true, 3, false, true, true>), grid_b, blk_b, lds, st, *a);
  } else if (bf) {
    const bool extra = a->wide_ops != 0, train = a->training != 0;
    if (a->max_groups <= 2) {
      if (extra) { if (train) HN_FWD(2, true, true); else HN_FWD(2, true, false); }
      else { if (train) HN_FWD(2, false, true); else HN_FWD(2, false, false); }
    } else {
      if (extra) { if (train) HN_FWD(3, true, true); else HN_FWD(3, true, false); }
      else { if (train) HN_FWD(3, false, true); else HN_FWD(3, false, false); }
    }
  } else {
    const dim3 grid_f(hn_grid_for(a->n_points, 128)), blk_f(256);
    if (a->training) hipLaunchKernelGGL((hn_mlp_fwd_kernel<false, 3, true, true>), grid_f, blk_f, lds, st, *a);
    else hipLaunchKernelGGL((hn_mlp_fwd_kernel<false, 3, true, false>), grid_f, blk_f, lds, st, *a);
  }
#undef HN_FWD
  HN_CHECK_LAUNCH();
  return 0;
}

extern "C" int hn_mlp_backward(const HnMlpArgs* a, hnStream_t stream) {
  int rc = hn_check_args(a);
  if (rc) return rc;
  hn_allow_big_lds();
  const size_t flds = (size_t)((a->n_feat + 1) & ~1) * 8;
  if (flds > 64 * 1024) return -6;
  if (a->mode == HN_MODE_BF16 || a->mode == HN_MODE_BF16_S8) {
    constexpr int WB = ModeT<true>::WAVES;
    const size_t lds = 2 * HN_CHUNK_UNITS * 1024 + flds + (size_t)a->n_feat * 16 + (size_t)WB * a->n_comps * 32 * 4;
    if (a->mode == HN_MODE_BF16_S8 && a->training)
      hipLaunchKernelGGL((hn_mlp_bwd_kernel<true, false, true>), dim3(hn_grid_for(a->n_points, WB * 32)), dim3(WB * 64), lds,
                         (hipStream_t)stream, *a);
    else if (a->wide_ops & 1)
      hipLaunchKernelGGL((hn_mlp_bwd_kernel<true, true>), dim3(hn_grid_for(a->n_points, WB * 32)), dim3(WB * 64), lds,
                         (hipStream_t)stream, *a);
    else
      hipLaunchKernelGGL((hn_mlp_bwd_kernel<true, false>), dim3(hn_grid_for(a->n_points, WB * 32)), dim3(WB * 64), lds,
                         (hipStream_t)stream, *a);
  } else {
    const size_t lds = 2 * HN_CHUNK_UNITS * 1024 + flds + (size_t)4 * a->n_comps * 32 * 4;
    hipLaunchKernelGGL((hn_mlp_bwd_kernel<false, true>), dim3(hn_grid_for(a->n_points, 128)), dim3(256), lds,
                       (hipStream_t)stream, *a);
  }
  HN_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// second half of a batched weight-gradient launch that flushed to partial slabs: one workgroup per destination tile
// ------------------------------------------------------------------------------------------------
struct HnDwReduceTable {
  float* partials[HN_MAX_WGRAD_BATCH];
  float* grads[HN_MAX_WGRAD_BATCH];
};
// ADAM = true (hn_mlp_wgrad_reduce_adam): the launch that completes the gradient also CONSUMES it — a workgroup that has
// summed its destination tile applies torch.optim.Adam's update to those elements in registers and writes p / m / v (the
// gradient buffer is read once more — whatever other launches added to it — and left zeroed): no gradient write-back, no
// second pass over the arena.  Elements no tile / bias record / table row covers (`rest`: parameters outside the launch's
// programs, alignment gaps) are updated by extra workgroups at the end of the grid, so every element of the arena is
// updated exactly once (the host proves the partition: machine.adam_rest_ranges, tests/test_host_api.py).
struct HnAdamFuseDev {
  float* p; float* g; float* m; float* v;
  const float* hyper; float* step;
  const HnAdamRange* rest;
  int n_rest, zero_grad;
};
#ifndef HN_REDUCE_SPLIT
#define HN_REDUCE_SPLIT 2      /* parts a destination tile's slab list is summed in (workgroup = SPLIT x 256 threads): 1 / 2 / 4 = 24.1 / 21.7 / 26.5 us per launch at config 2 (profiles/r06_reduce_variants.log) */
#endif
#ifndef HN_REDUCE_INFLIGHT
#define HN_REDUCE_INFLIGHT 8   /* slab loads a thread keeps in flight */
#endif
template <bool S8, bool ADAM>
__global__ __launch_bounds__(256 * HN_REDUCE_SPLIT) void hn_wgrad_reduce_kernel(const HnDwReduceTile* __restrict__ tiles, int n_tiles,
                                                              const uint32_t* __restrict__ list, const HnDwReduceTable tab,
                                                              const HnEmbedReduce em, const HnAdamFuseDev A) {
  __shared__ HnAdamConsts s_k;
  HnAdamConsts K = {};
  if (ADAM) {
    // one thread per block pays the bias-correction arithmetic (two powf, an rsqrtf); every block reads step[0] before it
    // does anything else, the block that finishes last advances it (hn_adam_ticket)
    if (threadIdx.x == 0) s_k = hn_adam_consts(A.hyper, A.step);
    __syncthreads();
    K = s_k;
  }
  // the gradient element at `G + off` is complete with `add`: plain path adds it (one writer per element), ADAM path
  // updates the parameter behind it
  auto finish = [&](float* Gp, float add) {
    if (ADAM) {
      const size_t i = (size_t)(Gp - A.g);
      const float g = *Gp + add;
      hn_adam_update(K, A.p[i], g, A.m[i], A.v[i]);
      *Gp = A.zero_grad ? 0.0f : g;
    } else {
#if defined(HN_REDUCE_ATOMIC) && HN_REDUCE_ATOMIC
      atomicAdd(Gp, add);
#else
      *Gp += add;      // one writer per element in this launch, every other launch is ordered before or behind it
#endif
    }
  };
  // (with HN_REDUCE_SPLIT > 1 the table-row, bias and rest paths run on the first 256 threads; every barrier below is
  // reached by all threads of the workgroup)
  const bool first256 = threadIdx.x < 256;
  if ((int)blockIdx.x < em.rows) {
    // one table row (these workgroups come FIRST in the grid: they are the long ones).  Thread x takes rays x, x + 256,
    // ... of every program in turn and, for the rays that carry this row, their blocks in order (a fixed order; four index
    // loads in flight — a thread that looked up the ray of every BLOCK in turn paid 24 dependent L2 latencies at config 2);
    // the 256 partial sums are added by a fixed shuffle tree per wave, the four waves' sums in wave order by threads
    // 0 .. dim-1, which add the row to the gradient (one writer per element)
    __shared__ float red[4][32];
    __shared__ int s_match[4][1024];
    const long long row = blockIdx.x;
    // (round 6) Wave w takes programs w, w + 4, ... — the programs' load chains run side by side.  Per program and 1024
    // rays: (1) sixteen ray-index loads per lane in flight, (2) the rays that carry this row are COMPACTED into an LDS
    // list in ray order (ballot + prefix count: deterministic), (3) the list is walked eight rays at a time, lane e
    // loading element e = (block of the ray, column) of each ray's partial rows — eight independent loads in flight, then
    // eight adds in list order.  The scan-and-accumulate form this replaces branched on every ray index and waited for
    // each matching ray's loads in turn: ~20 serialized memory round trips per row and program, the tail of the launch
    // (10 of its 35 us at config 2).  A fixed order throughout: the row's gradient stays bit-reproducible.
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    float colsum = 0.0f;          // lanes c < dim: column c of this wave's programs
    int* mlist = s_match[wave];
#ifdef HN_REDUCE_EXP_NOEMBED      /* timing-only experiment: the table rows do no work (their gradient is WRONG) */
    for (int sIdx = wave; first256 && sIdx < 0; sIdx += 4) {
#else
    for (int sIdx = wave; first256 && sIdx < em.n_src; sIdx += 4) {
#endif
      const float* __restrict__ P = em.partial[sIdx];
      const int64_t* __restrict__ idx = em.idx[sIdx];
      const int nb = em.n_blocks[sIdx], spr = em.samples_per_ray[sIdx];
      const int bpr = spr / 32;                         // blocks per ray (this path: samples_per_ray % 32 == 0)
      const int n_rays = nb / bpr;
      const int E = bpr * em.dim;                       // floats per ray: [block][column]
      float accv = 0.0f;                                // E <= 64: lane e sums element e over the row's rays
      float accw[32];                                   // E > 64 (spr > 256 at dim 8): per-column sums, lanes walk the blocks
#pragma unroll
      for (int c = 0; c < 32; ++c) accw[c] = 0.0f;
      for (int r0 = 0; r0 < n_rays; r0 += 1024) {
        long long id[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int ray = r0 + 64 * u + lane;
          id[u] = ray < n_rays ? (long long)idx[ray] : -1ll;
        }
        int nm = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const bool ok = id[u] == row;
          const unsigned long long m = __ballot(ok);
          if (ok) mlist[nm + __popcll(m & ((1ull << lane) - 1ull))] = r0 + 64 * u + lane;
          nm += __popcll(m);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        if (E <= 64) {
          for (int i0 = 0; i0 < nm; i0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              v[j] = 0.0f;
              if (i0 + j < nm && lane < E) v[j] = P[(size_t)mlist[i0 + j] * E + lane];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) accv += v[j];            // list order = ray order
          }
        } else {
          for (int i = 0; i < nm; ++i)
            for (int k = lane; k < bpr; k += 64) {
              const size_t bb = (size_t)mlist[i] * bpr + k;
#pragma unroll
              for (int c = 0; c < 32; ++c)
                if (c < em.dim) accw[c] += P[bb * em.dim + c];
            }
        }
        __builtin_amdgcn_wave_barrier();                          // the list is reused by the next 1024 rays
      }
      if (E <= 64) {
        // column c = the sum over the ray's blocks k of lane k * dim + c, blocks in order
        float t = 0.0f;
        for (int k = 0; k < bpr; ++k) {
          const float o = __shfl(accv, (k * em.dim + lane) & 63, 64);
          if (lane < em.dim) t += o;
        }
        colsum += t;
      } else {
#pragma unroll
        for (int c = 0; c < 32; ++c) {
          float v = accw[c];
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
          if (lane == c) colsum += v;
        }
      }
    }
    if (first256 && lane < 32) red[wave][lane] = (lane < em.dim && ((em.col_mask >> lane) & 1u)) ? colsum : 0.0f;
    __syncthreads();
    const int c = threadIdx.x;
    if (c < em.dim && ((em.col_mask >> c) & 1u)) {
      const float tot = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
      if (ADAM) finish(em.grad + row * em.dim + c, tot);
      else em.grad[row * em.dim + c] += tot;
    }
  } else if ((int)blockIdx.x < em.rows + n_tiles) {
    const HnDwReduceTile t = tiles[blockIdx.x - em.rows];
    if (t.ld == 0) {
      // bias record: col0 = dZ tiles of the rectangle; thread x = row 32 (x >> 5) + (x & 31) of it.  A job's bias slab
      // holds 32 floats per dZ tile, rows in natural order
      const int i_n = threadIdx.x >> 5, rr = threadIdx.x & 31;
      const int row = t.row0 + 32 * i_n + rr;
      if (first256 && !(i_n >= t.col0 || row < 0 || row >= t.r_end)) {
        const int off = i_n * 32 + rr;
        float sb = 0.0f;
        for (int k0 = 0; k0 < t.count; k0 += 8) {         // eight slab loads in flight (a load-add chain of up to 40 before)
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            v[u] = 0.0f;
            if (k0 + u < t.count) {
              const uint32_t e = list[t.first + k0 + u];
              float* P = tab.partials[0];
#pragma unroll
              for (int i = 1; i < HN_MAX_WGRAD_BATCH; ++i)
                if ((int)(e >> 28) == i) P = tab.partials[i];
              v[u] = P[(size_t)(e & 0x0fffffffu) * 1024 + off];
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) sb += v[u];          // fixed order: slab k0, k0 + 1, ...
        }
        float* Gb = tab.grads[0];
#pragma unroll
        for (int i = 1; i < HN_MAX_WGRAD_BATCH; ++i)
          if (t.batch == i) Gb = tab.grads[i];
        finish(Gb + t.w_off + row, sb);
      }
    } else {
      // a slab tile is [register quad q4 (4)][lane (64)][4 floats]; thread x sums the float4 at x: registers 4 q4 .. 4 q4 + 3
      // of one lane.  The slab loads are issued eight at a time (a dependent load-add chain pays one memory latency per
      // slab); with HN_REDUCE_SPLIT > 1 the workgroup has SPLIT x 256 threads and part j sums slabs j, j + SPLIT, ... —
      // SPLIT times the bytes in flight per tile, 1 / SPLIT of the chain — and part 0 adds the parts' sums in part order
      // (a fixed order: the gradient stays bit-reproducible).
      // With SPLIT > 1 every part ALSO finishes 4 / SPLIT of the thread's four elements (the parts' sums meet in LDS), so
      // the tail of a tile — the read-modify-write of the gradient or, ADAM, of p / m / v — is one element per thread
      // at SPLIT 4; its loads are issued BEFORE the slab loop (their latency hides behind it).
      constexpr int SP = HN_REDUCE_SPLIT, EPT = 4 / SP;      // elements a thread finishes
      static_assert(SP == 1 || SP == 2 || SP == 4, "HN_REDUCE_SPLIT");
      __shared__ f32x4 s_part[SP > 1 ? SP * 256 : 1];
      const int part = threadIdx.x >> 8, x = threadIdx.x & 255;
      float* G = tab.grads[0];
#pragma unroll
      for (int i = 1; i < HN_MAX_WGRAD_BATCH; ++i)
        if (t.batch == i) G = tab.grads[i];
      G += t.w_off;
      const int q4 = x >> 6, lane = x & 63, h = lane >> 5, c = lane & 31;
      const int col = t.col0 + (S8 ? hn_dw8_feature(c) : c);
      const bool col_ok = col >= 0 && col < t.c_end;
      float* gp[EPT];
      float g4[EPT], p4[EPT], m4[EPT], v4[EPT];
#pragma unroll
      for (int j = 0; j < EPT; ++j) {
        const int e = part * EPT + j;
        const int rr = hn_rho(4 * q4 + e, h);
        const int row = t.row0 + (S8 ? hn_dw8_feature(rr) : rr);
        gp[j] = (col_ok && row >= 0 && row < t.r_end) ? G + (size_t)row * t.ld + col : nullptr;
        if (ADAM && gp[j] != nullptr) {
          const size_t i = (size_t)(gp[j] - A.g);
          g4[j] = *gp[j]; p4[j] = A.p[i]; m4[j] = A.m[i]; v4[j] = A.v[i];
        }
      }
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      constexpr int NF = HN_REDUCE_INFLIGHT;
      for (int k0 = part; k0 < t.count; k0 += NF * SP) {
        f32x4 v[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) {
          v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          const int k = k0 + u * SP;
          if (k < t.count) {
            const uint32_t e = list[t.first + k];
            float* P = tab.partials[0];
#pragma unroll
            for (int i = 1; i < HN_MAX_WGRAD_BATCH; ++i)
              if ((int)(e >> 28) == i) P = tab.partials[i];
            v[u] = *reinterpret_cast<const f32x4*>(P + (size_t)(e & 0x0fffffffu) * 1024 + x * 4);
          }
        }
#pragma unroll
        for (int u = 0; u < NF; ++u) s += v[u];          // fixed order: slab k0, k0 + SPLIT, ...
      }
      float tot[EPT];
      if (SP > 1) {
        s_part[part * 256 + x] = s;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
          const int e = part * EPT + j;
          float a = s_part[x][e];
#pragma unroll
          for (int q = 1; q < SP; ++q) a += s_part[q * 256 + x][e];      // part order: fixed
          tot[j] = a;
        }
      } else {
#pragma unroll
        for (int j = 0; j < EPT; ++j) tot[j] = s[j];
      }
#pragma unroll
      for (int j = 0; j < EPT; ++j) {
        if (gp[j] == nullptr) continue;
        if (ADAM) {
          g4[j] += tot[j];
          hn_adam_update(K, p4[j], g4[j], m4[j], v4[j]);
        } else {
#if defined(HN_REDUCE_ATOMIC) && HN_REDUCE_ATOMIC
          atomicAdd(gp[j], tot[j]);
#else
          *gp[j] += tot[j];
#endif
        }
      }
      if (ADAM) {
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
          if (gp[j] == nullptr) continue;
          const size_t i = (size_t)(gp[j] - A.g);
          A.p[i] = p4[j]; A.m[i] = m4[j]; A.v[i] = v4[j];
          *gp[j] = A.zero_grad ? 0.0f : g4[j];
        }
      }
    }
  } else if (ADAM) {
    // the rest of the arena: elements nothing in this launch adds to (their gradient is whatever the buffer holds)
    const HnAdamRange r = A.rest[blockIdx.x - em.rows - n_tiles];
    for (int i = threadIdx.x; first256 && i < r.len; i += 256) finish(A.g + r.start + i, 0.0f);
  }
  if (ADAM) hn_adam_ticket(A.step, K.t);
}

static int hn_launch_reduce(int mode, const HnDwReduceTile* tiles, int n_tiles, const uint32_t* list,
                            const HnDwBatch* batches, int n_batches, const HnEmbedReduce* embed, const HnAdamFuse* adam,
                            hnStream_t stream) {
  if (n_tiles < 0 || n_batches < 0 || n_batches > HN_MAX_WGRAD_BATCH) return -1;
  HnEmbedReduce em = {};
  if (embed != nullptr) {
    em = *embed;
    if (em.n_src < 0 || em.n_src > HN_MAX_WGRAD_BATCH || em.rows < 0 || em.dim < 1 || em.dim > 32) return -2;
    if (em.n_src > 0 && em.grad == nullptr) return -3;
    for (int i = 0; i < em.n_src; ++i)
      if (em.partial[i] == nullptr || em.idx[i] == nullptr || em.n_blocks[i] < 0 || em.samples_per_ray[i] < 32) return -3;
    if (em.n_src == 0) em.rows = 0;
  }
  HnAdamFuseDev A = {};
  if (adam != nullptr) {
    if (adam->n < 1 || adam->n_rest < 0) return -2;
    if (adam->params == nullptr || adam->grads == nullptr || adam->exp_avg == nullptr || adam->exp_avg_sq == nullptr ||
        adam->hyper == nullptr || adam->step == nullptr || (adam->n_rest > 0 && adam->rest == nullptr))
      return -3;
    A.p = adam->params; A.g = adam->grads; A.m = adam->exp_avg; A.v = adam->exp_avg_sq;
    A.hyper = adam->hyper; A.step = adam->step; A.rest = adam->rest; A.n_rest = adam->n_rest; A.zero_grad = adam->zero_grad;
    // every destination of the launch must lie in the arena the optimizer steps
    if (em.rows > 0 && (em.grad < adam->grads || em.grad + (size_t)em.rows * em.dim > adam->grads + adam->n)) return -9;
  }
  const int total = n_tiles + em.rows + A.n_rest;
  if (total == 0) return 0;
  if (n_tiles > 0 && (tiles == nullptr || list == nullptr || batches == nullptr || n_batches < 1)) return -3;
  HnDwReduceTable tab = {};
  for (int i = 0; i < n_batches; ++i) {
    if (n_tiles > 0 && (batches[i].partials == nullptr || batches[i].grads == nullptr)) return -3;
    if (adam != nullptr && batches[i].grads != adam->grads) return -9;
    tab.partials[i] = batches[i].partials;
    tab.grads[i] = batches[i].grads;
  }
  const int m = mode & 255;
  const dim3 grid(total), blk(256 * HN_REDUCE_SPLIT);
  if (m == HN_MODE_BF16_S8) {
    if (adam != nullptr) hipLaunchKernelGGL((hn_wgrad_reduce_kernel<true, true>), grid, blk, 0, (hipStream_t)stream, tiles, n_tiles, list, tab, em, A);
    else hipLaunchKernelGGL((hn_wgrad_reduce_kernel<true, false>), grid, blk, 0, (hipStream_t)stream, tiles, n_tiles, list, tab, em, A);
  } else if (m == HN_MODE_BF16 || m == HN_MODE_F32) {
    if (adam != nullptr) hipLaunchKernelGGL((hn_wgrad_reduce_kernel<false, true>), grid, blk, 0, (hipStream_t)stream, tiles, n_tiles, list, tab, em, A);
    else hipLaunchKernelGGL((hn_wgrad_reduce_kernel<false, false>), grid, blk, 0, (hipStream_t)stream, tiles, n_tiles, list, tab, em, A);
  } else {
    return -2;
  }
  HN_CHECK_LAUNCH();
  return 0;
}

extern "C" int hn_mlp_wgrad_reduce(int mode, const HnDwReduceTile* tiles, int n_tiles, const uint32_t* list,
                                   const HnDwBatch* batches, int n_batches, const HnEmbedReduce* embed,
                                   hnStream_t stream) {
  return hn_launch_reduce(mode, tiles, n_tiles, list, batches, n_batches, embed, nullptr, stream);
}

extern "C" int hn_mlp_wgrad_reduce_adam(int mode, const HnDwReduceTile* tiles, int n_tiles, const uint32_t* list,
                                        const HnDwBatch* batches, int n_batches, const HnEmbedReduce* embed,
                                        const HnAdamFuse* adam, hnStream_t stream) {
  if (adam == nullptr) return -3;
  return hn_launch_reduce(mode, tiles, n_tiles, list, batches, n_batches, embed, adam, stream);
}

#if defined(HN_PROF) || defined(HN_WGRAD_JOBTIMES)
static void* hn_wgrad_prof = nullptr;
extern "C" void hn_set_wgrad_prof(void* p) { hn_wgrad_prof = p; }     // diagnostic builds only: (64, 8) int64 buffer
#endif
// HN_MODE_BF16 / HN_MODE_F32: bits 8..15 of the mode word state the LDS stage (KiB) the HOST cut the jobs' `bps` for
// (0 = not stated).  A stage longer than the ring this library was built with would never be loaded in full — the job
// tables live in device memory, so this is the one place the mismatch can be refused: -8.
static int hn_wgrad_stage_check(int mode_word) {
  if ((mode_word & 255) == HN_MODE_BF16_S8) return 0;      // (bits 8.. carry the dZ scale there; its ring is fixed)
  const int kb = (mode_word >> 8) & 255;
  if (kb > 8 * HN_WGRAD_MAXSLOT || kb * HN_WGRAD_STAGES > 160) return -8;
  return 0;
}
static int hn_launch_wgrad(int mode_word, HnDwBatchTable& tab, int total, hnStream_t stream) {
  hn_allow_big_lds();
  // HN_WGRAD_STAGES stages of <= 8 x HN_WGRAD_MAXSLOT KiB (2 x 64 KiB: bf16 32 tiles of 2 KiB, fp32 16 of 4 KiB); 8-bit stash 3 x 48 KiB
  size_t lds = (size_t)HN_WGRAD_STAGES * 8 * HN_WGRAD_MAXSLOT * 1024;
  if (lds > 160 * 1024) lds = 160 * 1024;
  const int mode = mode_word & 255, dz_log2 = mode_word >> 8;     // HN_MODE_BF16_S8 | dz_scale_log2 << 8
  tab.unscale = 1.0f;
  if (mode == HN_MODE_BF16_S8) {
    if (dz_log2 < -60 || dz_log2 > 60) return -2;
    tab.unscale = ldexpf(1.0f, -dz_log2);
    lds = 3 * 48 * 1024;
    hipLaunchKernelGGL((hn_wgrad_kernel<true, true>), dim3(total), dim3(512), lds, (hipStream_t)stream, tab);
  } else if (mode == HN_MODE_BF16)
    hipLaunchKernelGGL(hn_wgrad_kernel<true>, dim3(total), dim3(512), lds, (hipStream_t)stream, tab);
  else if (mode == HN_MODE_F32)
    hipLaunchKernelGGL(hn_wgrad_kernel<false>, dim3(total), dim3(512), lds, (hipStream_t)stream, tab);
  else
    return -2;
  HN_CHECK_LAUNCH();
  return 0;
}

extern "C" int hn_mlp_wgrad(int mode, const HnDwJob* jobs, int n_jobs, const void* stash, float* grads,
                            hnStream_t stream) {
  if (hn_wgrad_stage_check(mode) != 0) return -8;
  if (n_jobs < 0) return -1;
  if (n_jobs == 0) return 0;
  if (jobs == nullptr || stash == nullptr || grads == nullptr) return -3;
  HnDwBatchTable tab = {};
  tab.b[0].jobs = jobs; tab.b[0].stash = stash; tab.b[0].grads = grads; tab.b[0].n_jobs = n_jobs;
  tab.n = 1;
  return hn_launch_wgrad(mode, tab, n_jobs, stream);
}

extern "C" int hn_mlp_wgrad_batched(int mode, const HnDwBatch* batches, int n_batches, const int32_t* order_dev,
                                    hnStream_t stream) {
  return hn_mlp_wgrad_batched_t(mode, batches, n_batches, order_dev, nullptr, stream);
}

static int hn_cu_count() {
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

static int hn_wgrad_batched_impl(int mode, const HnDwBatch* batches, int n_batches, const int32_t* order_dev,
                                 uint64_t* timeline_dev, uint32_t* ticket_dev, hnStream_t stream);

extern "C" int hn_mlp_wgrad_batched_t(int mode, const HnDwBatch* batches, int n_batches, const int32_t* order_dev,
                                      uint64_t* timeline_dev, hnStream_t stream) {
  return hn_wgrad_batched_impl(mode, batches, n_batches, order_dev, timeline_dev, nullptr, stream);
}

#ifdef HN_WGRAD_PERSIST
extern "C" int hn_mlp_wgrad_batched_p(int mode, const HnDwBatch* batches, int n_batches, const int32_t* order_dev,
                                      uint64_t* timeline_dev, uint32_t* ticket_dev, hnStream_t stream) {
  if (ticket_dev == nullptr) return -3;
  return hn_wgrad_batched_impl(mode, batches, n_batches, order_dev, timeline_dev, ticket_dev, stream);
}
#endif

static int hn_wgrad_batched_impl(int mode, const HnDwBatch* batches, int n_batches, const int32_t* order_dev,
                                 uint64_t* timeline_dev, uint32_t* ticket_dev, hnStream_t stream) {
  if (hn_wgrad_stage_check(mode) != 0) return -8;
  if (n_batches < 0 || n_batches > HN_MAX_WGRAD_BATCH) return -1;
  if (n_batches > 0 && batches == nullptr) return -3;
  HnDwBatchTable tab = {};
  long long total = 0;
  for (int i = 0; i < n_batches; ++i) {
    if (batches[i].n_jobs < 0 || batches[i].n_jobs > 0xffffff) return -1;
    if (batches[i].n_jobs == 0) {
      if (order_dev != nullptr) return -1;      // an order table indexes the batches as passed
      continue;
    }
    if (batches[i].jobs == nullptr || batches[i].stash == nullptr || batches[i].grads == nullptr) return -3;
    tab.b[tab.n++] = batches[i];
    total += batches[i].n_jobs;
  }
  tab.order = order_dev;
  tab.timeline = timeline_dev;
#if defined(HN_PROF) || defined(HN_WGRAD_JOBTIMES)
  if (hn_wgrad_prof != nullptr && tab.n < HN_MAX_WGRAD_BATCH) tab.b[HN_MAX_WGRAD_BATCH - 1].jobs = (const HnDwJob*)hn_wgrad_prof;
#endif
  if (total == 0) return 0;
  if (total > 0x7fffffffLL) return -2;
#ifdef HN_WGRAD_PERSIST
  // the persistent form exists for the bf16 stash on the 2-stage ring with block reads (the product's build)
  if (ticket_dev != nullptr && (mode & 255) == HN_MODE_BF16 && HN_WGRAD_STAGES == 2 && HN_WGRAD_BLOCK != 0 && !HN_WGRAD_BIAS_MFMA) {
    hn_allow_big_lds();
    const size_t lds = (size_t)2 * 8 * HN_WGRAD_MAXSLOT * 1024 + 64;
    if (lds > 160 * 1024) return -8;
    const int cus = hn_cu_count();
    const int grid = total < cus ? (int)total : cus;      // one workgroup per CU (128 KiB of LDS each)
    hipLaunchKernelGGL(hn_wgrad_persist_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, tab, ticket_dev, (int)total);
    HN_CHECK_LAUNCH();
    return 0;
  }
#endif
  (void)ticket_dev;
  return hn_launch_wgrad(mode, tab, (int)total, stream);
}
