// Stand-alone micro-benchmark (diagnostic, not part of the product): LAYER-STATIONARY backward + weight gradient for a
// chain of L plain 256 -> 256 ReLU layers (the template trunk of the render levels), VERDICT r03 item 1, step A.
//
//   The 256 CUs are split into L stages of G workgroups (one 512-thread workgroup per CU, persistent).  A stage-s
//   workgroup keeps, for the whole launch,
//     * its layer's W^T in REGISTERS (each of the 8 waves one 32-row slab = 64 VGPRs: no weight ring, no ring barrier),
//     * its layer's dW in REGISTERS (256 KiB fp32 = 128 accumulator VGPRs x 8 waves; flushed once, by atomics, at the end),
//   and per 32-point block
//     * takes dZ_l  (16 KiB, operand layout of the bf16 stash) from the upstream stage through a ring in global memory
//       (write-through `sc1` stores, `sc1` LDS-DMA loads, one published-count word and one consumed-count word per
//       producer/consumer pair; stage 0 reads the top gradient from HBM),
//     * takes X_{l-1} (16 KiB) from the forward stash in HBM,
//     * computes dX = W^T . dZ  AND  dW += dZ . X^T  from the SAME dZ bytes in LDS,
//     * masks dX with (X_{l-1} > 0) — the ReLU mask is read off the stashed activation itself, no mask stream —
//       and hands dZ_{l-1} downstream.
//   dZ never goes to HBM between the layers (the last stage writes its output, as the real pipeline would hand it to
//   the encoder-gradient ops).
//
// Modes (argv[4]):  0 = dX product only, no hand-off (every stage reads HBM)      [forward-like ceiling]
//                   1 = both products, no hand-off (every stage reads HBM, writes HBM)
//                   2 = both products, hand-off through the rings                   [the pipeline]
//                   3 = dX product only, hand-off through the rings
//                   4 / 5 = as 2 / 3 with XCD-LOCAL pipelines: roles from the hardware XCC id + a per-XCD ticket, so the L
//                           stages of a pipeline share one L2; plain stores, the hand-off never leaves that L2
// Every mode is checked against plain reference kernels (fp32 accumulation of bf16 operands, same rounding points).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ls_bench tools/ls_bench.hip
// Run:   tools/ls_bench [points=131072] [layers=3] [reps=10] [mode=2] [G=0 (=256/L)] [ring slots=16]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <utility>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#define DEV __device__ __forceinline__
template <int... I, class F>
DEV void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
DEV void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

DEV int rho(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }
DEV int pi16(int h, int j) { return 8 * (j >> 2) + 4 * h + (j & 3); }
DEV int stash_slot(int r, int h, int u) { return 32 * h + (r ^ (4 * h + 8 * u)); }     // hn_mlp.hip hn_stash_slot
DEV unsigned pack2(float a, float b) {
  const bf16x2 v = __builtin_convertvector((f32x2){a, b}, bf16x2);
  return __builtin_bit_cast(unsigned, v);
}
DEV f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

constexpr int BLK_BYTES = 16384;        // one 32-point block of a 256-feature activation in the stash layout
#ifndef LS_SPREAD
#define LS_SPREAD 0      /* 1: the DMA pieces of a block are issued one at a time between the dW products */
#endif
#ifndef LS_STAGGER
#define LS_STAGGER 0     /* 1: waves 4-7 (the SIMD partners of 0-3) run dW before dX: matrix work beside the partner's issue / wait phases */
#endif
#ifndef LS_BBUF
#define LS_BBUF 2        /* B-fragment buffers of the dX product (2: the next four are read under the current products) */
#endif
#ifndef LS_NST
#define LS_NST 4
#endif
constexpr int NST = LS_NST;             // LDS stages of 32 KiB (dZ block | X block); -DLS_NST=3: two blocks in flight
constexpr int DPF = NST - 1;            // blocks whose DMA is in flight ahead of the one being computed
constexpr int FLAG_STRIDE = 32;         // unsigned words between two flags (128 B: one line each)

struct LsArgs {
  const char* wt;        // [L][8 slabs][16 frags][64 lanes][16 B]   W^T as A operands
  const char* x;         // [L][nblk][16 KiB]   forward stash of the layer inputs
  const char* dz_top;    // [nblk][16 KiB]      gradient at the top of the chain
  char* dz_out;          // [L][nblk][16 KiB]   per-stage output (hand-off modes: only the last stage writes, into slab L-1)
  char* ring;            // [L-1][G][R][16 KiB]
  unsigned* pub;         // [L-1][G] x FLAG_STRIDE   blocks published by the producer of the pair
  unsigned* freed;       // [L-1][G] x FLAG_STRIDE   blocks the consumer has landed in its LDS
  unsigned* tmo;         // [0] != 0: a spin gave up
  int* tickets;          // [8] x FLAG_STRIDE: roles handed out per XCD (LOCAL)
  float* dw;             // [L][256][256]
  float* db;             // [L][256]
  int nblk, L, G, R;
  int dbg;               // timing-only ablations (results wrong): 1 no block barrier, 2 no DMA, 4 no stores, 8 no dX LDS reads,
                         // 32 no vmcnt wait, 128 no polling in the block loop
  unsigned long long* prof;  // LS_PROF build: [2 waves per workgroup][8]: cycles per phase summed over the blocks
  unsigned long long* clk;   // [4 per workgroup]: s_memtime / s_memrealtime at start and end (wave 0)
};

DEV void wait_vmcnt(int n) {
  switch (n) {
#define C(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19)
    C(20) C(21) C(22) C(23) C(24)
#undef C
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

typedef __attribute__((address_space(1))) unsigned gu32;
DEV unsigned ld_flag(const unsigned* p) {
  return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DEV void st_flag(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// wave-uniform spin: until *p >= want; returns the value seen; gives up after ~2^21 polls and records it.  No load is
// left pending on any exit (a pending load's destination register would make hipcc guard every later write of that
// register — the asm LDS reads of every wave — with s_waitcnt vmcnt(0), draining the DMA ring each block)
DEV unsigned spin_ge(const unsigned* p, unsigned want, unsigned* tmo, int code, bool& dead) {
  int spins = 0;
  for (;;) {
    const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)ld_flag(p));
    if ((int)(v - want) >= 0) return v;
    if (dead || ++spins > (1 << 20)) { st_flag(tmo, (unsigned)code); dead = true; return 0x7fffffffu; }   // never spins again
    __builtin_amdgcn_s_sleep(2);
  }
}

// transposed reads of the weight-gradient operands: hn_mlp.hip hn_dw_tr_offsets / hn_tr_tile
DEV void dw_tr_offsets(int lane, int& o0, int& o1) {
  const int G = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int u = G & 1, hh = G >> 1, h = pp & 1, g = pp >> 1;
  o0 = u * 1024 + stash_slot(8 * hh + q, h, u) * 16 + 8 * g;
  o1 = u * 1024 + stash_slot(8 * hh + 4 + q, h, u) * 16 + 8 * g;
}
// two tiles (2 KiB apart... any OFFA / OFFB) = 8 transposed reads under one wait
template <int OFFA, int OFFB>
DEV void tr_tiles2(bf16x8* va, bf16x8* vb, unsigned a0, unsigned a1) {
  u32x2 r0, r1, r2, r3, r4, r5, r6, r7;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%10\n\t"
      "ds_read_b64_tr_b16 %1, %9 offset:%10\n\t"
      "ds_read_b64_tr_b16 %2, %8 offset:%11\n\t"
      "ds_read_b64_tr_b16 %3, %9 offset:%11\n\t"
      "ds_read_b64_tr_b16 %4, %8 offset:%12\n\t"
      "ds_read_b64_tr_b16 %5, %9 offset:%12\n\t"
      "ds_read_b64_tr_b16 %6, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %7, %9 offset:%13\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
      : "v"(a0), "v"(a1), "n"(OFFA), "n"(OFFA + 256), "n"(OFFB), "n"(OFFB + 256)
      : "memory");
  va[0] = __builtin_bit_cast(bf16x8, (u32x4){r0[0], r0[1], r1[0], r1[1]});
  va[1] = __builtin_bit_cast(bf16x8, (u32x4){r2[0], r2[1], r3[0], r3[1]});
  vb[0] = __builtin_bit_cast(bf16x8, (u32x4){r4[0], r4[1], r5[0], r5[1]});
  vb[1] = __builtin_bit_cast(bf16x8, (u32x4){r6[0], r6[1], r7[0], r7[1]});
}
// four B fragments (ds_read_b128) issued together; the wait is the caller's
template <int O0, int O1, int O2, int O3>
DEV void read4(u32x4& a, u32x4& b, u32x4& c, u32x4& d, unsigned addr0, unsigned addr1) {
  // units alternate u = 0 / 1: the lane's slot differs between them (stash_slot), hence two base addresses
  asm volatile(
      "ds_read_b128 %0, %4 offset:%6\n\t"
      "ds_read_b128 %1, %5 offset:%7\n\t"
      "ds_read_b128 %2, %4 offset:%8\n\t"
      "ds_read_b128 %3, %5 offset:%9"
      : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
      : "v"(addr0), "v"(addr1), "n"(O0), "n"(O1), "n"(O2), "n"(O3)
      : "memory");
}
#define LGKM_WAIT4(a, b, c, d, n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory")

#ifdef LS_PROF
#define TS(x) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory"); } while (0)
#define ACC(i, a_, b_) prof[i] += (b_) - (a_)
#else
#define TS(x) do {} while (0)
#define ACC(i, a_, b_) do {} while (0)
#endif
template <bool DW, bool HANDOFF, bool LOCAL = false>
__global__ __launch_bounds__(512, 2) void ls_stage_kernel(const LsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int G = a.G, L = a.L, R = a.R, nblk = a.nblk;
  const int Rm = R - 1;                        // R is a power of two
  int stage = blockIdx.x / G, g = blockIdx.x % G;
  if constexpr (LOCAL) {
    // XCD-local pipelines: a workgroup reads the XCD it really runs on (hardware register, not an assumption about
    // dispatch) and takes the next role of THAT XCD from a ticket counter: the L stages of a pipeline always share one
    // L2, so the hand-off can stay in it (plain stores, L1-bypassing loads) — correctness by construction, not by
    // placement luck; an XCD that received more workgroups than it has roles leaves the surplus idle (and the run fails
    // its check if another XCD then lacks one)
    int* role = reinterpret_cast<int*>(smem);
    if (threadIdx.x == 0) {
      const int xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 15;      // HW_REG_XCC_ID[3:0]
      const int t = atomicAdd(a.tickets + xcc * FLAG_STRIDE, 1);
      const int gx = G / 8;                              // pipelines per XCD
      role[0] = (xcc < 8 && t < L * gx) ? ((t / gx) << 16 | (xcc * gx + t % gx)) : -1;
    }
    __syncthreads();
    const int rl = role[0];
    __syncthreads();
    if (rl < 0) return;
    stage = __builtin_amdgcn_readfirstlane(rl >> 16);
    g = __builtin_amdgcn_readfirstlane(rl & 0xffff);
  }
  if (stage >= L) return;
  const int nq = (nblk - g + G - 1) / G;                 // my blocks: g, g + G, ...
  const bool from_ring = HANDOFF && stage > 0, to_ring = HANDOFF && stage < L - 1;
  const char* xin = a.x + (size_t)stage * nblk * BLK_BYTES;
  const char* zin = from_ring ? a.ring + ((size_t)(stage - 1) * G + g) * R * BLK_BYTES
                              : (HANDOFF ? a.dz_top : (stage == 0 ? a.dz_top : a.dz_out + (size_t)(stage - 1) * nblk * BLK_BYTES));
  char* zout = to_ring ? a.ring + ((size_t)stage * G + g) * R * BLK_BYTES : a.dz_out + (size_t)stage * nblk * BLK_BYTES;
  // (mode without hand-off: stage s > 0 reads what a PREVIOUS launch's stage s-1 wrote: same bytes moved, no dependence)
  unsigned* pub_in = a.pub + ((size_t)(stage - 1) * G + g) * FLAG_STRIDE;       // valid when from_ring
  unsigned* freed_in = a.freed + ((size_t)(stage - 1) * G + g) * FLAG_STRIDE;
  unsigned* pub_out = a.pub + ((size_t)stage * G + g) * FLAG_STRIDE;            // valid when to_ring
  unsigned* freed_out = a.freed + ((size_t)stage * G + g) * FLAG_STRIDE;

  unsigned long long t0c = 0, t0r = 0;
  if (wave == 0) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
  const int dbg = a.dbg;
  // W^T slab of this wave: 16 A fragments (rows 32 wave .. +31 of dX, all 256 contraction features)
  bf16x8 wt[16];
  {
    const char* wp = a.wt + ((size_t)(stage * 8 + wave) * 16) * 1024 + lane * 16;
#pragma unroll
    for (int f = 0; f < 16; ++f) wt[f] = *reinterpret_cast<const bf16x8*>(wp + f * 1024);
    // make hipcc wait for these loads HERE: left pending, it guards the first MFMA of every block with s_waitcnt
    // vmcnt(0) (it cannot see the asm waits of the loop), which drains the DMA ring
#pragma unroll
    for (int f = 0; f < 16; ++f) asm volatile("" : "+v"(wt[f]));
  }
  // dW rectangle of this wave: n tiles 2 wn, 2 wn + 1 ; k tiles 4 wk .. 4 wk + 3
  const int wn = wave >> 1, wk = wave & 1;
  f32x16 acc[2][4];
  float bsum = 0.0f;       // bias gradient of n tile 2 wn + wk, feature (lane & 31), this lane's 8 + 8 points per block
  if constexpr (DW) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  }
  int tro0, tro1;
  dw_tr_offsets(lane, tro0, tro1);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // this lane's B-fragment read addresses inside a tile (unit 0 / unit 1)
  const unsigned bo0 = stash_slot(r, h, 0) * 16, bo1 = 1024 + stash_slot(r, h, 1) * 16;

  int ops = 0;                       // vector-memory instructions issued by this wave
  unsigned long long hist = 0;       // `ops` right after the DMA of the most recent blocks (16 bits each, latest lowest)
  auto blk_of = [&](int q) { return g + q * G; };
  auto piece = [&](int q, int i) __attribute__((always_inline)) {     // one of this wave's 4 DMA pieces of block q (0, 1: dZ; 2, 3: X)
    if (q < nq && !(dbg & 2)) {
      char* dst = smem + (q % NST) * (2 * BLK_BYTES);
      const int unit = wave + 8 * (i & 1);
      if (i < 2) {
        const char* zs = from_ring ? zin + (size_t)(q & Rm) * BLK_BYTES : zin + (size_t)blk_of(q) * BLK_BYTES;
        if (from_ring)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(zs + unit * 1024 + lane * 16),
                                           (__attribute__((address_space(3))) void*)(dst + unit * 1024), 16, 0, 16);   // sc1
        else
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(zs + unit * 1024 + lane * 16),
                                           (__attribute__((address_space(3))) void*)(dst + unit * 1024), 16, 0, 2);    // nt
      } else {
        const char* xs = xin + (size_t)blk_of(q) * BLK_BYTES;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xs + unit * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + BLK_BYTES + unit * 1024), 16, 0, 2);
      }
      ops += 1;
    }
  };
  auto issue_mark = [&]() __attribute__((always_inline)) { hist = (hist << 16) | (unsigned)(ops & 0xffff); };
  auto issue = [&](int q) { for (int i = 0; i < 4; ++i) piece(q, i); issue_mark(); };

  // buffer descriptor of the output (write-through 16-byte stores need the raw-buffer form for the sc1 bit)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(zout, 0, -1, 0x00020000);

  // prologue: DMA of the first DPF blocks
  bool dead = false;
  unsigned pub_seen = 0, freed_seen = 0;       // wave 0: the counts as last read (a poll is skipped while they suffice)
  for (int q = 0; q < DPF; ++q) {
    if (from_ring && q < nq && wave == 0 && pub_seen < (unsigned)(q + 1)) pub_seen = spin_ge(pub_in, (unsigned)(q + 1), a.tmo, 1, dead);
    if (from_ring) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    issue(q);
  }

#ifdef LS_PROF
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
#endif
  // the block loop, instantiated once per phase order (a run-time choice of the order inside ONE loop makes hipcc merge
  // the register state of both orders at every iteration: 480-544 B/lane of scratch)
  auto block_loop = [&](auto hi_) __attribute__((always_inline)) {
  constexpr bool HI = decltype(hi_)::value;
  for (int q = 0; q < nq; ++q) {
    TS(s0);
    // ---- top: block q has landed; flags; barrier ------------------------------------------------------------
    if (HANDOFF && wave == 0 && !(dbg & 128)) {
      // the block whose DMA this iteration issues must have been published ...
      if (from_ring && q + DPF < nq && pub_seen < (unsigned)(q + DPF + 1))
        pub_seen = spin_ge(pub_in, (unsigned)(q + DPF + 1), a.tmo, 2, dead);
      // ... and the ring slot this iteration's stores go to must have been consumed
      if (to_ring && q >= R && freed_seen < (unsigned)(q - R + 1))
        freed_seen = spin_ge(freed_out, (unsigned)(q - R + 1), a.tmo, 3, dead);
    }
    {
      const int mark = (int)((hist >> (16 * (DPF - 1))) & 0xffff);      // ops right after DMA(q)
      // steady state: (DPF - 1) x (4 DMA + 2 stores) younger operations may stay in flight; first / last blocks: drain
      if (dbg & 32) {}
      else if (((ops - mark) & 0xffff) >= 6 * (DPF - 1)) wait_vmcnt(6 * (DPF - 1));
      else wait_vmcnt(0);
    }
    TS(s1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(dbg & 1)) __builtin_amdgcn_s_barrier();
    TS(s2);
    __builtin_amdgcn_sched_barrier(0);
    if (HANDOFF && wave == 0 && lane == 0) {
      // every wave's DMA of block q has landed: ring slots up to q are free again
      if (from_ring) st_flag(freed_in, (unsigned)(q + 1));
      // every wave's stores of blocks <= q - DPF are complete (they were issued before DMA(q))
      // (staggered waves store AFTER their DMA issue: their stores of block q - DPF are only known complete one block later)
      if (to_ring && q >= DPF + LS_STAGGER) st_flag(pub_out, (unsigned)(q - DPF + 1 - LS_STAGGER));
    }
    const unsigned sb = lds0 + (unsigned)((q % NST) * (2 * BLK_BYTES));

    auto phase_dx = [&]() __attribute__((always_inline)) {
    // ---- dX tile `wave` = W^T slab . dZ block --------------------------------------------------------------------
    f32x16 dx;
#pragma unroll
    for (int e = 0; e < 16; ++e) dx[e] = 0.0f;
    {
      const unsigned z0 = sb + bo0, z1 = sb + bo1;
#if LS_BBUF == 1
      // one buffer of four fragments: 16 registers fewer (what the counter-phase build needs to stay out of scratch);
      // every group's LDS latency is exposed to this wave (its SIMD partner covers it)
      u32x4 b0[4];
      static_for<4>([&](auto G_) __attribute__((always_inline)) {
        constexpr int g4 = decltype(G_)::value;
        read4<4096 * g4, 4096 * g4, 4096 * g4 + 2048, 4096 * g4 + 2048>(b0[0], b0[1], b0[2], b0[3], z0, z1);
        LGKM_WAIT4(b0[0], b0[1], b0[2], b0[3], 0);
#pragma unroll
        for (int f = 0; f < 4; ++f) dx = mfma(wt[4 * g4 + f], __builtin_bit_cast(bf16x8, b0[f]), dx);
        __builtin_amdgcn_sched_barrier(0);
      });
#else
      u32x4 b0[4], b1[4];
      read4<0, 0, 2048, 2048>(b0[0], b0[1], b0[2], b0[3], z0, z1);
      read4<4096, 4096, 6144, 6144>(b1[0], b1[1], b1[2], b1[3], z0, z1);
      LGKM_WAIT4(b0[0], b0[1], b0[2], b0[3], 4);
#pragma unroll
      for (int f = 0; f < 4; ++f) dx = mfma(wt[f], __builtin_bit_cast(bf16x8, b0[f]), dx);
      __builtin_amdgcn_sched_barrier(0);
      read4<8192, 8192, 10240, 10240>(b0[0], b0[1], b0[2], b0[3], z0, z1);
      LGKM_WAIT4(b1[0], b1[1], b1[2], b1[3], 4);
#pragma unroll
      for (int f = 0; f < 4; ++f) dx = mfma(wt[4 + f], __builtin_bit_cast(bf16x8, b1[f]), dx);
      __builtin_amdgcn_sched_barrier(0);
      read4<12288, 12288, 14336, 14336>(b1[0], b1[1], b1[2], b1[3], z0, z1);
      LGKM_WAIT4(b0[0], b0[1], b0[2], b0[3], 4);
#pragma unroll
      for (int f = 0; f < 4; ++f) dx = mfma(wt[8 + f], __builtin_bit_cast(bf16x8, b0[f]), dx);
      __builtin_amdgcn_sched_barrier(0);
      LGKM_WAIT4(b1[0], b1[1], b1[2], b1[3], 0);
#pragma unroll
      for (int f = 0; f < 4; ++f) dx = mfma(wt[12 + f], __builtin_bit_cast(bf16x8, b1[f]), dx);
#endif
    }
    TS(s3);
    // ---- ReLU mask off the stashed X tile `wave`, conversion, write-through stores ---------------------------------
    {
      u32x4 m0, m1;
      const unsigned xa = sb + BLK_BYTES + wave * 2048;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(m0), "=&v"(m1) : "v"(xa + bo0), "v"(xa + bo1) : "memory");
      u32x4 o0, o1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned w0 = m0[k], w1 = m1[k];
        const float a0 = (w0 & 0xffffu) ? dx[2 * k] : 0.0f, a1 = (w0 >> 16) ? dx[2 * k + 1] : 0.0f;
        const float c0 = (w1 & 0xffffu) ? dx[8 + 2 * k] : 0.0f, c1 = (w1 >> 16) ? dx[8 + 2 * k + 1] : 0.0f;
        o0[k] = pack2(a0, a1);
        o1[k] = pack2(c0, c1);
      }
      const unsigned off = (unsigned)((to_ring ? (q & Rm) : blk_of(q)) * BLK_BYTES + wave * 2048);
      if (dbg & 4) {
        asm volatile("" :: "v"(o0), "v"(o1));
      } else if (LOCAL && to_ring) {
        __builtin_amdgcn_raw_buffer_store_b128(o0, rsrc, off + bo0, 0, 0);          // plain: stays in the XCD's L2
        __builtin_amdgcn_raw_buffer_store_b128(o1, rsrc, off + bo1, 0, 0);
      } else if (HANDOFF) {
        __builtin_amdgcn_raw_buffer_store_b128(o0, rsrc, off + bo0, 0, 16);         // sc1: write-through
        __builtin_amdgcn_raw_buffer_store_b128(o1, rsrc, off + bo1, 0, 16);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(o0, rsrc, off + bo0, 0, 2);          // nt
        __builtin_amdgcn_raw_buffer_store_b128(o1, rsrc, off + bo1, 0, 2);
      }
      ops += 2;
    }
    };
    auto phase_dw = [&]() __attribute__((always_inline)) {
    TS(s4);
    // ---- DMA of block q + DPF (its LDS stage was read last in iteration q - 1) -------------------------------------
    if (!(LS_SPREAD && DW)) { for (int i = 0; i < 4; ++i) piece(q + DPF, i); }
    TS(s5);
    // ---- dW rectangle += dZ tiles (A) . X tiles (B), both transposed on the LDS read -----------------------------------
    if constexpr (DW) {
      bf16x8 za[2][2];
      const unsigned az = sb + (unsigned)(2 * wn * 2048);
      tr_tiles2<0, 2048>(za[0], za[1], az + tro0, az + tro1);
      if (LS_SPREAD) piece(q + DPF, 0);
      // bias gradient: sum over the points of dZ (feature on the lane, 8 + 8 points in the registers); tile 2 wn + wk
      // (a scalar branch: indexing za[] with the run-time wk makes hipcc select every register through v_cndmask chains)
      {
        auto add16 = [&](const bf16x8& p0, const bf16x8& p1) {
          const u32x4 v0 = __builtin_bit_cast(u32x4, p0), v1 = __builtin_bit_cast(u32x4, p1);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            bsum += __uint_as_float(v0[k] << 16) + __uint_as_float(v0[k] & 0xffff0000u);
            bsum += __uint_as_float(v1[k] << 16) + __uint_as_float(v1[k] & 0xffff0000u);
          }
        };
        if (wk) add16(za[1][0], za[1][1]);
        else add16(za[0][0], za[0][1]);
      }
      const unsigned ax = sb + BLK_BYTES + (unsigned)(4 * wk * 2048);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        bf16x8 xb[2][2];
        if (half == 0) tr_tiles2<0, 2048>(xb[0], xb[1], ax + tro0, ax + tro1);
        else tr_tiles2<4096, 6144>(xb[0], xb[1], ax + tro0, ax + tro1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][2 * half + j] = mfma(za[i][0], xb[j][0], acc[i][2 * half + j]);
            acc[i][2 * half + j] = mfma(za[i][1], xb[j][1], acc[i][2 * half + j]);
          }
          // one DMA piece behind every four products: its issue time overlaps the matrix pipe's backlog
          if (LS_SPREAD && half == 0 && i == 0) piece(q + DPF, 1);
          if (LS_SPREAD && half == 0 && i == 1) piece(q + DPF, 2);
          if (LS_SPREAD && half == 1 && i == 0) piece(q + DPF, 3);
        }
      }
    }
    issue_mark();
    };
    if constexpr (HI) { phase_dw(); phase_dx(); }
    else { phase_dx(); phase_dw(); }
    TS(s6);
    ACC(0, s0, s1); ACC(1, s1, s2); ACC(2, s2, s3); ACC(3, s3, s4); ACC(4, s4, s5); ACC(5, s5, s6);
  }
  };
  if (LS_STAGGER && wave >= 4) block_loop(std::true_type{});
  else block_loop(std::false_type{});
#ifdef LS_PROF
  if (lane == 0 && a.prof != nullptr && (wave == 0 || wave == 5)) {
    unsigned long long* o = a.prof + ((size_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 8;
    for (int i = 0; i < 6; ++i) o[i] = prof[i];
    o[6] = (unsigned long long)nq; o[7] = (unsigned long long)stage;
  }
#endif
  if (wave == 0 && lane == 0 && a.clk != nullptr) {
    a.clk[4 * blockIdx.x + 0] = t0c; a.clk[4 * blockIdx.x + 1] = t0r;
    a.clk[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime(); a.clk[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
  }
  // ---- end: publish the last blocks; flush dW ----------------------------------------------------------------------
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (to_ring && wave == 0 && lane == 0) st_flag(pub_out, (unsigned)nq);
  if constexpr (DW) {
    float* Gd = a.dw + (size_t)stage * 65536;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = 32 * (2 * wn + i) + rho(e, h), col = 32 * (4 * wk + j) + r;
          atomicAdd(Gd + row * 256 + col, acc[i][j][e]);
        }
    // lanes c and c + 32 hold the two point halves of feature c
    bsum += __shfl_xor(bsum, 32, 64);
    if (h == 0) atomicAdd(a.db + stage * 256 + 32 * (2 * wn + wk) + r, bsum);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// plain reference kernels on natural [point][256] arrays of bf16 bits
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float((unsigned)b << 16); }
__global__ void ref_dx_kernel(const uint16_t* dz, const uint16_t* x, const uint16_t* w /*[n][k]*/, uint16_t* out, int P) {
  // out[p][k] = x[p][k] != 0 ? bf16( sum_n dz[p][n] w[n][k] ) : 0 ; one block = 64 points x 256 k
  __shared__ float zs[64][17];
  const int k = threadIdx.x;          // 256 threads
  const int p0 = blockIdx.x * 64;
  float acc[64];
  for (int i = 0; i < 64; ++i) acc[i] = 0.0f;
  for (int n0 = 0; n0 < 256; n0 += 16) {
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
      const int pp = i / 16, nn = i % 16;
      zs[pp][nn] = (p0 + pp < P) ? bf2f(dz[(size_t)(p0 + pp) * 256 + n0 + nn]) : 0.0f;
    }
    __syncthreads();
    for (int nn = 0; nn < 16; ++nn) {
      const float wv = bf2f(w[(size_t)(n0 + nn) * 256 + k]);
      for (int i = 0; i < 64; ++i) acc[i] = fmaf(zs[i][nn], wv, acc[i]);
    }
  }
  for (int i = 0; i < 64; ++i)
    if (p0 + i < P) {
      const size_t o = (size_t)(p0 + i) * 256 + k;
      const unsigned u = pack2(acc[i], 0.0f) & 0xffffu;
      out[o] = x[o] != 0 ? (uint16_t)u : (uint16_t)0;
    }
}
__global__ void ref_dw_kernel(const uint16_t* dz, const uint16_t* x, float* dw, float* db, int P, int chunk) {
  // dw[n][k] += sum_{p in chunk} dz[p][n] x[p][k] ; block = (n tile of 16) x (all k), grid.y = chunks
  const int k = threadIdx.x, n0 = blockIdx.x * 16;
  const int p0 = blockIdx.y * chunk, p1 = min(P, p0 + chunk);
  float acc[16], bs = 0.0f;
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  for (int p = p0; p < p1; ++p) {
    const float xv = bf2f(x[(size_t)p * 256 + k]);
    for (int i = 0; i < 16; ++i) acc[i] = fmaf(bf2f(dz[(size_t)p * 256 + n0 + i]), xv, acc[i]);
    if (k < 16) bs += bf2f(dz[(size_t)p * 256 + n0 + k]);
  }
  for (int i = 0; i < 16; ++i) atomicAdd(dw + (size_t)(n0 + i) * 256 + k, acc[i]);
  if (k < 16) atomicAdd(db + n0 + k, bs);
}

// natural [p][256] <-> stash layout [block][tile][unit][slot][8]
__global__ void to_stash_kernel(const uint16_t* nat, uint16_t* st, int P) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one per (point, feature)
  if (idx >= (size_t)P * 256) return;
  const int p = (int)(idx / 256), f = (int)(idx % 256);
  const int blk = p / 32, r = p % 32, t = f / 32, ft = f % 32;
  const int u = ft / 16, w = ft % 16, h = (w >> 2) & 1, j = (w & 3) + 4 * (w >> 3);
  st[(((size_t)blk * 8 + t) * 2 + u) * 512 + (32 * h + (r ^ (4 * h + 8 * u))) * 8 + j] = nat[idx];
}
__global__ void from_stash_kernel(const uint16_t* st, uint16_t* nat, int P) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)P * 256) return;
  const int p = (int)(idx / 256), f = (int)(idx % 256);
  const int blk = p / 32, r = p % 32, t = f / 32, ft = f % 32;
  const int u = ft / 16, w = ft % 16, h = (w >> 2) & 1, j = (w & 3) + 4 * (w >> 3);
  nat[idx] = st[(((size_t)blk * 8 + t) * 2 + u) * 512 + (32 * h + (r ^ (4 * h + 8 * u))) * 8 + j];
}
__global__ void fill_kernel(uint16_t* a, size_t n, unsigned seed, int relu_like, float scale) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned v = (unsigned)i * 2654435761u ^ (seed * 0x9E3779B9u + (unsigned)(i >> 32));
  v ^= v >> 15; v *= 0x2c1b3c6du; v ^= v >> 12; v *= 0x297a2d39u; v ^= v >> 15;
  float f = ((float)(v & 0xffffff) / 8388608.0f - 1.0f) * scale;
  if (relu_like && f < 0.0f) f = 0.0f;
  a[i] = (uint16_t)(pack2(f, 0.0f) & 0xffffu);
}

static float bf16_round(float x) {
  uint32_t u; memcpy(&u, &x, 4);
  u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
  float y; memcpy(&y, &u, 4); return y;
}
static uint16_t bf16_bits(float x) { float y = bf16_round(x); uint32_t u; memcpy(&u, &y, 4); return (uint16_t)(u >> 16); }
static float bf16_to_f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float y; memcpy(&y, &u, 4); return y; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 131072;
  const int L = argc > 2 ? atoi(argv[2]) : 3;
  const int reps = argc > 3 ? atoi(argv[3]) : 10;
  const int mode = argc > 4 ? atoi(argv[4]) : 2;
  int G = argc > 5 ? atoi(argv[5]) : 0;
  const int R = argc > 6 ? atoi(argv[6]) : 16;
  const int check = argc > 7 ? atoi(argv[7]) : 1;
  const int dbg = argc > 8 ? atoi(argv[8]) : 0;
  if (G <= 0) G = 256 / L;
  if (P % 32 != 0 || L * G > 256 || (R & (R - 1)) != 0 || R < 2 * DPF + 2 + LS_STAGGER /* fewer slots: producer and consumer wait for each other */) { printf("bad arguments\n"); return 2; }
  const int nblk = P / 32;
  const bool dwm = mode == 1 || mode == 2 || mode == 4, handoff = mode >= 2;
  if (mode >= 4 && (G % 8 != 0 || L * (G / 8) > 32)) { printf("XCD-local modes need G = 8 x pipelines per XCD, L * G / 8 <= 32\n"); return 2; }
  const size_t act = (size_t)P * 256;         // elements of one activation

  // weights: W[l][n][k] bf16 ; W^T slabs as A operands: slab w, fragment f = 2 t + u, lane (m, h), element j:
  //   W[n = 32 t + 16 u + pi16(h, j)][k = 32 w + m]
  std::vector<float> W((size_t)L * 65536);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  const float lim = sqrtf(6.0f / 256.0f);
  for (auto& w : W) w = bf16_round(rnd() * lim);
  std::vector<uint16_t> wt((size_t)L * 8 * 16 * 512), wnat((size_t)L * 65536);
  for (size_t i = 0; i < W.size(); ++i) wnat[i] = bf16_bits(W[i]);
  for (int l = 0; l < L; ++l)
    for (int w = 0; w < 8; ++w)
      for (int f = 0; f < 16; ++f)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int m = lane & 31, h = lane >> 5, t = f >> 1, u = f & 1;
            const int n = 32 * t + 16 * u + 8 * (j >> 2) + 4 * h + (j & 3), k = 32 * w + m;
            wt[((((size_t)l * 8 + w) * 16 + f) * 64 + lane) * 8 + j] = wnat[((size_t)l * 256 + n) * 256 + k];
          }
  char *d_wt; uint16_t *d_wnat, *d_xnat, *d_xst, *d_ztop_nat, *d_ztop_st, *d_zout_st, *d_tmp_nat, *d_ref_a, *d_ref_b;
  CK(hipMalloc(&d_wt, wt.size() * 2)); CK(hipMemcpy(d_wt, wt.data(), wt.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_wnat, wnat.size() * 2)); CK(hipMemcpy(d_wnat, wnat.data(), wnat.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_xnat, (size_t)L * act * 2));
  CK(hipMalloc(&d_xst, (size_t)L * act * 2));
  CK(hipMalloc(&d_ztop_nat, act * 2)); CK(hipMalloc(&d_ztop_st, act * 2));
  CK(hipMalloc(&d_zout_st, (size_t)L * act * 2));
  CK(hipMalloc(&d_tmp_nat, act * 2)); CK(hipMalloc(&d_ref_a, act * 2)); CK(hipMalloc(&d_ref_b, act * 2));
  const int TB = 256;
  const unsigned gb = (unsigned)((act + TB - 1) / TB);
  for (int l = 0; l < L; ++l) {
    fill_kernel<<<gb, TB>>>(d_xnat + (size_t)l * act, act, 100 + l, 1, 1.0f);
    to_stash_kernel<<<gb, TB>>>(d_xnat + (size_t)l * act, d_xst + (size_t)l * act, P);
  }
  fill_kernel<<<gb, TB>>>(d_ztop_nat, act, 7, 0, 1.0f);
  to_stash_kernel<<<gb, TB>>>(d_ztop_nat, d_ztop_st, P);
  CK(hipMemset(d_zout_st, 0, (size_t)L * act * 2));
  CK(hipDeviceSynchronize());

  char* d_ring; unsigned *d_flags; float *d_dw, *d_db, *d_dw_ref, *d_db_ref;
  const size_t ring_bytes = (size_t)std::max(1, L - 1) * G * R * BLK_BYTES;
  const size_t flag_words = (size_t)std::max(1, L - 1) * G * FLAG_STRIDE;
  CK(hipMalloc(&d_ring, ring_bytes)); CK(hipMemset(d_ring, 0xff, ring_bytes));
  CK(hipMalloc(&d_flags, (2 * flag_words + 64 + 8 * FLAG_STRIDE) * 4));
  CK(hipMalloc(&d_dw, (size_t)L * 65536 * 4)); CK(hipMalloc(&d_db, (size_t)L * 256 * 4));
  CK(hipMalloc(&d_dw_ref, (size_t)L * 65536 * 4)); CK(hipMalloc(&d_db_ref, (size_t)L * 256 * 4));
  LsArgs a{};
  a.wt = d_wt; a.x = (const char*)d_xst; a.dz_top = (const char*)d_ztop_st; a.dz_out = (char*)d_zout_st; a.ring = d_ring;
  a.pub = d_flags; a.freed = d_flags + flag_words; a.tmo = d_flags + 2 * flag_words; a.tickets = (int*)(d_flags + 2 * flag_words + 64);
  a.dw = d_dw; a.db = d_db; a.nblk = nblk; a.L = L; a.G = G; a.R = R;
  unsigned long long* d_clk; CK(hipMalloc(&d_clk, 256 * 4 * 8)); CK(hipMemset(d_clk, 0, 256 * 4 * 8));
  a.clk = d_clk; a.dbg = 0;
  unsigned long long* d_prof; CK(hipMalloc(&d_prof, 256 * 2 * 8 * 8)); CK(hipMemset(d_prof, 0, 256 * 2 * 8 * 8));
  a.prof = d_prof;

  const int lds = NST * 2 * BLK_BYTES;
  auto kern = mode == 0 ? ls_stage_kernel<false, false> : mode == 1 ? ls_stage_kernel<true, false>
            : mode == 2 ? ls_stage_kernel<true, true> : mode == 3 ? ls_stage_kernel<false, true>
            : mode == 4 ? ls_stage_kernel<true, true, true> : ls_stage_kernel<false, true, true>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    CK(hipMemsetAsync(d_flags, 0, (2 * flag_words + 64 + 8 * FLAG_STRIDE) * 4, st));
    hipLaunchKernelGGL(kern, dim3(L * G), dim3(512), lds, st, a);
  };
  // without hand-off stage s reads the previous launch's output of stage s-1: run the chain L times so that every
  // stage has seen its true input before the check (results of the LAST launch are checked)
  CK(hipMemsetAsync(d_dw, 0, (size_t)L * 65536 * 4, st)); CK(hipMemsetAsync(d_db, 0, (size_t)L * 256 * 4, st));
  for (int i = 0; i < (handoff ? 1 : L); ++i) launch();
  CK(hipStreamSynchronize(st));
  unsigned tmo = 0; CK(hipMemcpy(&tmo, a.tmo, 4, hipMemcpyDeviceToHost));
  if (tmo) { printf("FAIL: a spin gave up (code %u)\n", tmo); return 3; }

  int bad = 0;
  if (check) {
    // results of one clean launch
    CK(hipMemsetAsync(d_dw, 0, (size_t)L * 65536 * 4, st)); CK(hipMemsetAsync(d_db, 0, (size_t)L * 256 * 4, st));
    launch();
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(&tmo, a.tmo, 4, hipMemcpyDeviceToHost));
    if (tmo) { printf("FAIL: a spin gave up (code %u)\n", tmo); return 3; }
    // reference chain
    CK(hipMemset(d_dw_ref, 0, (size_t)L * 65536 * 4)); CK(hipMemset(d_db_ref, 0, (size_t)L * 256 * 4));
    const uint16_t* zin = d_ztop_nat;
    uint16_t* bufs[2] = {d_ref_a, d_ref_b};
    const int chunk = 2048;
    for (int l = 0; l < L; ++l) {
      if (dwm) ref_dw_kernel<<<dim3(16, (P + chunk - 1) / chunk), 256>>>(zin, d_xnat + (size_t)l * act, d_dw_ref + (size_t)l * 65536, d_db_ref + l * 256, P, chunk);
      ref_dx_kernel<<<(P + 63) / 64, 256>>>(zin, d_xnat + (size_t)l * act, d_wnat + (size_t)l * 65536, bufs[l & 1], P);
      zin = bufs[l & 1];
    }
    from_stash_kernel<<<gb, TB>>>(d_zout_st + (size_t)(L - 1) * act, d_tmp_nat, P);
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> got(act), want(act);
    CK(hipMemcpy(got.data(), d_tmp_nat, act * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(want.data(), zin, act * 2, hipMemcpyDeviceToHost));
    double num = 0, den = 0; size_t nbad = 0; float worst = 0;
    for (size_t i = 0; i < act; ++i) {
      const float x = bf16_to_f(got[i]), y = bf16_to_f(want[i]);
      num += (double)(x - y) * (x - y); den += (double)y * y;
      const float tol = 0.02f * fabsf(y) + 0.02f;
      if (!(fabsf(x - y) <= tol)) { ++nbad; worst = std::max(worst, fabsf(x - y)); }
    }
    printf("check dZ_out: rel L2 %.3e, %zu of %zu outside tolerance (worst %.3g)\n", sqrt(num / std::max(den, 1e-30)), nbad, act, worst);
    if (nbad > act / 100000 || !(sqrt(num / std::max(den, 1e-30)) < 2e-2)) bad = 1;
    if (dwm) {
      std::vector<float> gw((size_t)L * 65536), rw((size_t)L * 65536), gbv((size_t)L * 256), rbv((size_t)L * 256);
      CK(hipMemcpy(gw.data(), d_dw, gw.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(rw.data(), d_dw_ref, rw.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(gbv.data(), d_db, gbv.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(rbv.data(), d_db_ref, rbv.size() * 4, hipMemcpyDeviceToHost));
      for (int l = 0; l < L; ++l) {
        double n1 = 0, d1 = 0, n2 = 0, d2 = 0;
        for (int i = 0; i < 65536; ++i) { const double x = gw[(size_t)l * 65536 + i], y = rw[(size_t)l * 65536 + i]; n1 += (x - y) * (x - y); d1 += y * y; }
        for (int i = 0; i < 256; ++i) { const double x = gbv[l * 256 + i], y = rbv[l * 256 + i]; n2 += (x - y) * (x - y); d2 += y * y; }
        const double e1 = sqrt(n1 / std::max(d1, 1e-30)), e2 = sqrt(n2 / std::max(d2, 1e-30));
        printf("check layer %d: dW rel L2 %.3e, db rel L2 %.3e\n", l, e1, e2);
        if (!(e1 < 1e-2) || !(e2 < 1e-2)) bad = 1;
      }
    }
    printf(bad ? "CHECK FAILED\n" : "check ok\n");
  }

  // timing: back-to-back warm-up launches first (the clock the chip holds under sustained load, not the boost of the
  // first launches after an idle period), then `reps` timed launches
  a.dbg = dbg;
  for (int i = 0; i < 30; ++i) launch();
  CK(hipStreamSynchronize(st));
  std::vector<float> ms;
  for (int i = 0; i < reps; ++i) {
    CK(hipMemsetAsync(d_flags, 0, (2 * flag_words + 64 + 8 * FLAG_STRIDE) * 4, st));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(kern, dim3(L * G), dim3(512), lds, st, a);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
  }
  CK(hipMemcpy(&tmo, a.tmo, 4, hipMemcpyDeviceToHost));
  {
    std::vector<unsigned long long> clk(256 * 4);
    CK(hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz, us;
    for (int w = 0; w < L * G; ++w)
      if (clk[4 * w + 3] > clk[4 * w + 1]) {
        ghz.push_back((double)(clk[4 * w + 2] - clk[4 * w]) / (double)(clk[4 * w + 3] - clk[4 * w + 1]) * 0.1);
        us.push_back((double)(clk[4 * w + 3] - clk[4 * w + 1]) * 0.01);
      }
    if (!ghz.empty()) {
      std::sort(ghz.begin(), ghz.end()); std::sort(us.begin(), us.end());
      printf("in-kernel (last launch): shader clock median %.3f GHz, workgroup lifetime median %.1f us (min %.1f, max %.1f)\n",
             ghz[ghz.size() / 2], us[us.size() / 2], us[0], us.back());
    }
  }
#ifdef LS_PROF
  {
    std::vector<unsigned long long> pr(256 * 2 * 8);
    CK(hipMemcpy(pr.data(), d_prof, pr.size() * 8, hipMemcpyDeviceToHost));
    const char* names[6] = {"poll+vmcnt wait", "barrier", "dX product", "mask+cvt+stores", "DMA issue", "dW product"};
    for (int wv = 0; wv < 2; ++wv) {
      double sum[6] = {0, 0, 0, 0, 0, 0}; double nb = 0;
      for (int w = 0; w < L * G; ++w) {
        const unsigned long long* o = &pr[((size_t)w * 2 + wv) * 8];
        if (o[6] == 0) continue;
        for (int i = 0; i < 6; ++i) sum[i] += (double)o[i];
        nb += (double)o[6];
      }
      double tot = 0; for (int i = 0; i < 6; ++i) tot += sum[i];
      printf("prof wave %d: cycles per block:", wv ? 5 : 0);
      for (int i = 0; i < 6; ++i) printf("  %s %.0f", names[i], sum[i] / nb);
      printf("  | total %.0f\n", tot / nb);
    }
    // per stage (wave 0)
    for (int st_ = 0; st_ < L; ++st_) {
      double sum[6] = {0, 0, 0, 0, 0, 0}; double nb = 0;
      for (int w = 0; w < L * G; ++w) {
        const unsigned long long* o = &pr[((size_t)w * 2) * 8];
        if (o[6] == 0 || (int)o[7] != st_) continue;
        for (int i = 0; i < 6; ++i) sum[i] += (double)o[i];
        nb += (double)o[6];
      }
      if (nb > 0) { printf("prof stage %d wave 0:", st_); for (int i = 0; i < 6; ++i) printf(" %.0f", sum[i] / nb); printf("\n"); }
    }
  }
#endif
  printf("reps (ms):");
  for (float t : ms) printf(" %.3f", t);
  printf("\n");
  std::sort(ms.begin(), ms.end());
  const double t = ms[ms.size() / 2] * 1e-3;
  const double flops = (dwm ? 2.0 : 1.0) * 2.0 * 65536.0 * (double)P * L;
  const double cus = (double)L * G;
  printf("mode %d dbg %d  P %d  L %d  G %d  R %d : median %.3f ms (min %.3f)  %.1f TFLOP/s = %.3f of 2.5 PF (%.3f of the %d CUs used)  "
         "blocks/CU %d  %.2f us/block%s\n",
         mode, dbg, P, L, G, R, t * 1e3, ms[0], flops / t / 1e12, flops / t / 2.5e15, flops / t / (2.5e15 * cus / 256.0), (int)cus,
         (nblk + G - 1) / G, t * 1e6 / ((nblk + G - 1) / G), tmo ? "  [SPIN TIMEOUT]" : "");
  return bad ? 1 : 0;
}
