// Shared device helpers for the gfx950 HyperNeRF kernels: MFMA fragment maps and mode traits.
//
// Layout conventions (wave64, one wave = one block of 32 points):
//   lane l : r = l & 31 (point inside the block / MFMA row or column), h = l >> 5 (k half).
//   32x32 accumulator tile (C/D map of every 32x32 MFMA on gfx950):
//       register i of lane (r,h) holds element [row = rho(i,h)][col = r],  rho(i,h) = (i&3) + 8*(i>>2) + 4*h
//   bf16 A/B operand of v_mfma_f32_32x32x16_bf16: element j of lane (r,h) is k = 8*h + j.
//   Activations are kept TRANSPOSED (features = rows, points = columns/lanes): H_next^T = W . H^T,
//   so W (nn.Linear (out,in) layout) is the A operand, the activation the B operand, and an
//   accumulator tile is already the next layer's B operand: registers 8u..8u+7 of a tile, converted
//   to bf16, are the B fragment of k-step u with the k order  pi16(h,j) = 8*(j>>2) + 4*h + (j&3);
//   the packed weights use the same k order, so no lane ever moves.
//   fp32 mode (v_mfma_f32_32x32x2_f32, A/B = one float per lane, k = h): accumulator register q is
//   directly the B operand of "step q", which contracts features rho(q,0) and rho(q,1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hn_kernels.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define HN_DEV __device__ __forceinline__

// Kernel timeline (HnMlpArgs.timeline, hn_mlp_wgrad_batched_t): a launch measures ITSELF — the first workgroup to
// arrive stores the 100 MHz wall clock, the last one to finish adds (now - start) to a running sum and counts the run —
// so that a step replayed as ONE HIP graph still yields per-kernel durations of exactly the replays that were timed
// (no events exist between the nodes of a graph).  Two device-scope atomics per workgroup; tickets re-arm themselves.
//   t[0] start of the current run   t[1] its end   t[2] arrive ticket   t[3] finish ticket
//   t[4] sum of (end - start) over runs, in 10 ns ticks   t[5] runs   t[6] first start ever   t[7] last end
HN_DEV void hn_timeline_begin(uint64_t* t) {
  if (t == nullptr || threadIdx.x != 0) return;
  const uint64_t now = wall_clock64();
  if (atomicAdd(reinterpret_cast<unsigned*>(t + 2), 1u) == 0u) {
    atomicExch(reinterpret_cast<unsigned long long*>(t), (unsigned long long)now);
    if (atomicAdd(reinterpret_cast<unsigned long long*>(t + 6), 0ull) == 0ull)
      atomicExch(reinterpret_cast<unsigned long long*>(t + 6), (unsigned long long)now);
  }
}
HN_DEV void hn_timeline_end(uint64_t* t) {
  if (t == nullptr) return;
  // every wave of the workgroup has finished its stores before thread 0 stamps the end (without the barrier waves 1-7
  // could still be writing their tiles when wave 0 of the LAST workgroup reads the clock: durations under-reported).
  // `t` is a kernel argument and every call site is workgroup-uniform, so the barrier is too.
  __syncthreads();
  if (threadIdx.x != 0) return;
  const uint64_t now = wall_clock64();
  if (atomicAdd(reinterpret_cast<unsigned*>(t + 3), 1u) == gridDim.x - 1) {
    const uint64_t start = atomicAdd(reinterpret_cast<unsigned long long*>(t), 0ull);
    atomicExch(reinterpret_cast<unsigned long long*>(t + 1), (unsigned long long)now);
    atomicExch(reinterpret_cast<unsigned long long*>(t + 7), (unsigned long long)now);
    atomicAdd(reinterpret_cast<unsigned long long*>(t + 4), (unsigned long long)(now - start));
    atomicAdd(reinterpret_cast<unsigned long long*>(t + 5), 1ull);
    atomicExch(reinterpret_cast<unsigned*>(t + 2), 0u);
    atomicExch(reinterpret_cast<unsigned*>(t + 3), 0u);
  }
}

// torch.optim.Adam's update of ONE element (utils.get_optimizer's default, reference utils/__init__.py:23-41), shared by
// hn_adam_kernel and the reduce launch that applies it on the fly (hn_mlp_wgrad_reduce_adam): the same operations in the
// same order, so both paths produce bit-identical parameters.
struct HnAdamConsts {
  float beta1, beta2, eps, weight_decay, gscale, step_size, inv_sqrt_bc2, t;
};
HN_DEV HnAdamConsts hn_adam_consts(const float* __restrict__ hyper, const float* step) {
  // hyper-parameters are read from device memory: a captured launch (HIP graph) follows later changes of lr etc.
  HnAdamConsts c;
  const float lr = hyper[0];
  c.beta1 = hyper[1]; c.beta2 = hyper[2]; c.eps = hyper[3]; c.weight_decay = hyper[4];
  c.gscale = hyper[5];      // 1 / world size after a SUM all-reduce (1 otherwise)
  c.t = step[0] + 1.0f;     // step[0] = number of updates done so far; this launch is update t
  const float bc1 = 1.0f - powf(c.beta1, c.t), bc2 = 1.0f - powf(c.beta2, c.t);
  c.step_size = lr / bc1;
  c.inv_sqrt_bc2 = rsqrtf(bc2);
  return c;
}
HN_DEV void hn_adam_update(const HnAdamConsts& c, float& p, float g, float& m, float& v) {
  const float gk = g * c.gscale + c.weight_decay * p;
  m = c.beta1 * m + (1.0f - c.beta1) * gk;
  v = c.beta2 * v + (1.0f - c.beta2) * gk * gk;
  p -= c.step_size * m / (sqrtf(v) * c.inv_sqrt_bc2 + c.eps);
}
// the block that finishes LAST (ticket counter in step[1]) stores t and re-arms the ticket — by then every block has read
// the old value, so no second launch is needed to advance the counter.  Call from ALL threads of the block.
HN_DEV void hn_adam_ticket(float* step, float t) {
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* ticket = reinterpret_cast<unsigned*>(step) + 1;
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
      step[0] = t;
      *ticket = 0u;
    }
  }
}

HN_DEV int hn_rho(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }
HN_DEV int hn_pi16(int h, int j) { return 8 * (j >> 2) + 4 * h + (j & 3); }

HN_DEV f32x16 hn_mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
HN_DEV f32x16 hn_mfma_f32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

#ifndef HN_BF16_WAVES
#define HN_BF16_WAVES 8
#endif

template <bool BF16>
struct ModeT;

template <>
struct ModeT<true> {
  static constexpr int WAVES = HN_BF16_WAVES;   // 8: 512 threads, 2 waves/SIMD (256 regs) ; 4: 1 wave/SIMD (512 regs)
  static constexpr int STEPS32 = 2;    // operand fragments per 32 features
  static constexpr int UNITS32 = 2;    // 1-KiB weight units per (32 out x 32 in) block
  static constexpr int TILE_UNITS = 2; // 1-KiB units per stashed 32x32 tile
  using Frag = bf16x8;
};
template <>
struct ModeT<false> {
  static constexpr int WAVES = 4;      // 256 threads, 1 wave per SIMD (512 registers)
  static constexpr int STEPS32 = 16;
  static constexpr int UNITS32 = 4;
  static constexpr int TILE_UNITS = 4;
  using Frag = float;
};

// acc += W[32 rows][32 in-features] . in   — one (out tile x 32-feature block)
// lds: UNITS32 consecutive 1-KiB units of the packed stream.
HN_DEV void hn_mma_block32(f32x16& acc, const char* lds, const bf16x8* in, int lane) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    bf16x8 a = *reinterpret_cast<const bf16x8*>(lds + u * 1024 + lane * 16);
    acc = hn_mfma_bf16(a, in[u], acc);
  }
}
HN_DEV void hn_mma_block32(f32x16& acc, const char* lds, const float* in, int lane) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    f32x4 a = *reinterpret_cast<const f32x4*>(lds + u * 1024 + lane * 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = hn_mfma_f32(a[e], in[4 * u + e], acc);
  }
}

// fp32 mode only (the bf16 stash keeps the operand layout and is transposed by the weight-gradient kernel's LDS read):
// Z = X^T through the matrix core: fr = the 16 steps of one 32-feature tile in the points-on-lanes layout (they are,
// unchanged, the A operand of X^T); B = a permuted identity.  Result: lane (c,h) holds feature c of the tile at the 16
// points rho(q,h) — the operand layout of the dW product.
HN_DEV f32x16 hn_transpose_tile(const float* fr, int lane) {
  const int c = lane & 31, h = lane >> 5;
  f32x16 z = {0};
#pragma unroll
  for (int q = 0; q < 16; ++q) z = hn_mfma_f32(fr[q], (c == hn_rho(q, h)) ? 1.0f : 0.0f, z);
  return z;
}
// store a transposed fp32 tile (4 KiB at dst): [g][lane][4]
HN_DEV void hn_store_tile(const f32x16& z, char* dst, int lane, float*) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 o = {z[4 * g], z[4 * g + 1], z[4 * g + 2], z[4 * g + 3]};
    __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dst + g * 1024 + lane * 16));
  }
}

// accumulator tile -> operand fragments of the next product (no lane movement)
HN_DEV void hn_acc_to_frags(const f32x16& a, bf16x8* out) {
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int j = 0; j < 8; ++j) out[u][j] = (__bf16)a[8 * u + j];
}
HN_DEV void hn_acc_to_frags(const f32x16& a, float* out) {
#pragma unroll
  for (int q = 0; q < 16; ++q) out[q] = a[q];
}

HN_DEV void hn_zero_frags(bf16x8* f, int n) {
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) f[i][j] = (__bf16)0.0f;
}
HN_DEV void hn_zero_frags(float* f, int n) {
  for (int i = 0; i < n; ++i) f[i] = 0.0f;
}

#define HN_CHECK_LAUNCH()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)
