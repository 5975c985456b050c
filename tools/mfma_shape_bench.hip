// Stand-alone micro-benchmark (diagnostic, not part of the product; round 5, verdict item 6): the forward machine's
// chain of plain 256 -> 256 ReLU layers — register-resident transposed activations, 8 waves x 32 points per workgroup,
// the weight stream through a 2 x 32 KiB LDS ring filled by LDS-DMA, one barrier per chunk — once with the product's
// accumulator shape, v_mfma_f32_32x32x16_bf16, and once with v_mfma_f32_16x16x32_bf16, the shape for which
// MI355X_MICROARCH.md reports a higher sustained clock on data-like operands.  Same epilogue WORK in both: bias,
// ReLU as one integer max, conversion to the next layer's operand fragments and — training — one mask bit per
// activation and two 16-byte non-temporal stash stores per 32 features and wave.
//
// 16x16x32 fragment algebra (the part a port of the machine kernels would have to adopt): a wave's 32 points are two
// column groups of 16; an output tile is 16 features x 16 points in 4 accumulator registers (lane (c, g): point c,
// features 4g + i); two vertically adjacent tiles are exactly one B operand of the next layer (K = 32: lane (c, g)
// element j <-> feature 32 s + (j < 4 ? 4 g + j : 16 + 4 g + j - 4)), the weights are packed in that k order, and one
// A fragment (16 rows x 32 k, 1 KiB per wave) feeds the MFMAs of BOTH column groups.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/mfma_shape_bench tools/mfma_shape_bench.hip
// Run:   tools/mfma_shape_bench [points] [layers] [reps]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#define DEV __device__ __forceinline__

DEV unsigned pack2(float a, float b) {
  const bf16x2 v = __builtin_convertvector((f32x2){a, b}, bf16x2);
  return __builtin_bit_cast(unsigned, v);
}
DEV float relu(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
DEV bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
DEV int pi16(int h, int j) { return 8 * (j >> 2) + 4 * h + (j & 3); }                   // 32x32x16: k of element j, half h
DEV int pi32(int g, int j) { return j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4); }         // 16x16x32: k of element j, group g

struct Args {
  const char* wstream;   // L * 128 units of 1 KiB in the shape's fragment order
  const float* bias;     // L * 256
  char* stash;           // [L][block][8][2 KiB]
  uint32_t* masks;       // [L][block][4][64]
  __bf16* out;           // [points][256]
  int n_points, n_layers;
};

constexpr int CHUNK = 32;   // units of 1 KiB per ring chunk
struct Ring {               // the product's weight ring: 2 chunks, vmcnt(0) + barrier at a chunk boundary
  const char* g;
  char* lds;
  int nchunks, wave, lane, ctr;
  DEV void issue(int c) {
    const char* src = g + (size_t)c * (CHUNK * 1024);
    char* dst = lds + (c & 1) * (CHUNK * 1024);
#pragma unroll
    for (int i = 0; i < CHUNK / 8; ++i) {
      const int unit = wave + 8 * i;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + unit * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + unit * 1024), 16, 0, 0);
    }
  }
  DEV void start() {
    ctr = 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(0);
  }
  DEV const char* take(int n) {       // n consecutive units (never straddles a chunk)
    if ((ctr & (CHUNK - 1)) == 0) {
      const int c = ctr / CHUNK;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < nchunks) issue(c + 1);
    }
    const char* p = lds + (((ctr / CHUNK) & 1) * CHUNK + (ctr & (CHUNK - 1))) * 1024;
    ctr += n;
    return p;
  }
};

// SHAPE 32: v_mfma_f32_32x32x16_bf16 (the product).  SHAPE 16: v_mfma_f32_16x16x32_bf16.  TRAIN: masks + stash stores.
template <int SHAPE, bool TRAIN>
__global__ __launch_bounds__(512, 2) void chain(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Ring ws;
  ws.g = a.wstream; ws.lds = smem; ws.nchunks = a.n_layers * 128 / CHUNK; ws.wave = wave; ws.lane = lane;
  float* bias_lds = reinterpret_cast<float*>(smem + 2 * CHUNK * 1024);
  for (int i = threadIdx.x; i < a.n_layers * 256; i += blockDim.x) bias_lds[i] = a.bias[i];
  __syncthreads();
  const int nblk_total = (a.n_points + 31) / 32;
  const int blk = blockIdx.x * 8 + wave;
  // cur[u]: operand fragment u of the 256-feature activation of this wave's 32 points.
  //   SHAPE 32: u = k-step of 16 features (16 fragments), lane (r = point, h), element j <-> feature 16 u + pi16(h, j)
  //   SHAPE 16: u = 2 s + cg: k-step s of 32 features, column group cg; lane (c, g), element j <-> feature 32 s + pi32(g, j)
  u32x4 cur[16], nxt[16];
#pragma unroll
  for (int u = 0; u < 16; ++u)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      unsigned p, f;
      if (SHAPE == 32) { p = blk * 32 + (lane & 31); f = 16 * u + pi16(lane >> 5, 2 * jj); }
      else { p = blk * 32 + 16 * (u & 1) + (lane & 15); f = 32 * (u >> 1) + pi32(lane >> 4, 2 * jj); }
      // input: +-[0.5, 1) from a hash of (point, feature pair); features f, f + 1 are adjacent elements in both orders
      unsigned x = p * 2654435761u ^ (f * 40503u + 12345u);
      x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
      cur[u][jj] = (x & 0x807f807fu) | 0x3f003f00u;
    }
  ws.start();
  for (int l = 0; l < a.n_layers; ++l) {
    const float* bias = bias_lds + l * 256;
    char* out_base = a.stash + ((size_t)l * nblk_total + blk) * (8 * 2048) + lane * 16;
    uint32_t* mask_base = a.masks + ((size_t)l * nblk_total + blk) * 256 + lane;
    unsigned bits = 0;
    if constexpr (SHAPE == 32) {
      const int h = lane >> 5;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const char* wl = ws.take(16) + lane * 16;
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 32 * t + 8 * g + 4 * h);
          acc[4 * g] = b[0]; acc[4 * g + 1] = b[1]; acc[4 * g + 2] = b[2]; acc[4 * g + 3] = b[3];
        }
        bf16x8 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const bf16x8*>(wl + u * 1024);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q[u % 4], as_frag(cur[u]), acc, 0, 0, 0);
          if (u + 4 < 16) q[u % 4] = *reinterpret_cast<const bf16x8*>(wl + (u + 4) * 1024);
        }
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          if (TRAIN) {
            bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(acc[i]), 31);
            bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(acc[i + 1]), 31);
          }
          nxt[t * 2 + (i >> 3)][(i & 7) >> 1] = pack2(relu(acc[i]), relu(acc[i + 1]));
        }
        if (TRAIN) {
          if (t & 1) { mask_base[(t >> 1) * 64] = bits; bits = 0; }
          __builtin_nontemporal_store(nxt[2 * t], reinterpret_cast<u32x4*>(out_base + (size_t)t * 2048));
          __builtin_nontemporal_store(nxt[2 * t + 1], reinterpret_cast<u32x4*>(out_base + (size_t)t * 2048 + 1024));
        }
      }
    } else {
      const int g = lane >> 4;
      // 16 output tiles of 16 features; tiles 2 s and 2 s + 1 make operand fragment s of the next layer (both groups)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const char* wl = ws.take(16) + lane * 16;      // 8 k-steps of tile 2 s, then 8 of tile 2 s + 1
        f32x4 acc[2][2];                                // [tile of the pair][column group]
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 32 * s + 16 * tt + 4 * g);
          acc[tt][0] = b; acc[tt][1] = b;
        }
        bf16x8 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const bf16x8*>(wl + u * 1024);
#pragma unroll
        for (int u = 0; u < 16; ++u) {                  // u = 8 tt + k-step
          const int tt = u >> 3, ks = u & 7;
          acc[tt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q[u % 4], as_frag(cur[2 * ks]), acc[tt][0], 0, 0, 0);
          acc[tt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q[u % 4], as_frag(cur[2 * ks + 1]), acc[tt][1], 0, 0, 0);
          if (u + 4 < 16) q[u % 4] = *reinterpret_cast<const bf16x8*>(wl + (u + 4) * 1024);
        }
#pragma unroll
        for (int cg = 0; cg < 2; ++cg)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
              if (TRAIN) {
                bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(acc[tt][cg][i]), 31);
                bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(acc[tt][cg][i + 1]), 31);
              }
              nxt[2 * s + cg][2 * tt + (i >> 1)] = pack2(relu(acc[tt][cg][i]), relu(acc[tt][cg][i + 1]));
            }
        if (TRAIN) {
          if (s & 1) { mask_base[(s >> 1) * 64] = bits; bits = 0; }
          __builtin_nontemporal_store(nxt[2 * s], reinterpret_cast<u32x4*>(out_base + (size_t)s * 2048));
          __builtin_nontemporal_store(nxt[2 * s + 1], reinterpret_cast<u32x4*>(out_base + (size_t)s * 2048 + 1024));
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) cur[u] = nxt[u];
  }
#pragma unroll
  for (int u = 0; u < 16; ++u)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int p, f;
      if (SHAPE == 32) { p = blk * 32 + (lane & 31); f = 16 * u + pi16(lane >> 5, j); }
      else { p = blk * 32 + 16 * (u & 1) + (lane & 15); f = 32 * (u >> 1) + pi32(lane >> 4, j); }
      if (p < a.n_points) a.out[(size_t)p * 256 + f] = as_frag(cur[u])[j];
    }
}

static float bf16_round(float x) {
  uint32_t u; memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  float y; memcpy(&y, &u, 4); return y;
}
static uint16_t bf16_bits(float x) { float y = bf16_round(x); uint32_t u; memcpy(&u, &y, 4); return (uint16_t)(u >> 16); }
static float bf16_to_f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float y; memcpy(&y, &u, 4); return y; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int SHAPE, bool TRAIN>
static double run(const char* name, Args a, int reps, std::vector<uint16_t>* out_host) {
  auto k = chain<SHAPE, TRAIN>;
  const size_t lds = (size_t)2 * CHUNK * 1024 + a.n_layers * 256 * 4;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int grid = (a.n_points + 255) / 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double flops = 2.0 * a.n_points * 256.0 * 256.0 * a.n_layers;
  printf("%-28s train=%d  %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 2.5 PF)\n", name, (int)TRAIN, ms, flops / ms * 1e-9,
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
  std::vector<float> W((size_t)L * 256 * 256), B((size_t)L * 256);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  const float lim = sqrtf(6.0f / 256.0f);
  for (auto& w : W) w = bf16_round(rnd() * lim);
  for (auto& b : B) b = rnd() * 0.05f;
  // SHAPE 32: [layer][tile of 32 rows][k-step of 16][lane (r, h)][8]: W[32 t + r][16 u + pi16(h, j)]
  std::vector<uint16_t> p32((size_t)L * 128 * 512), p16((size_t)L * 128 * 512);
  for (int l = 0; l < L; ++l)
    for (int t = 0; t < 8; ++t)
      for (int u = 0; u < 16; ++u)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int r = lane & 31, h = lane >> 5;
            p32[((((size_t)l * 8 + t) * 16 + u) * 64 + lane) * 8 + j] =
                bf16_bits(W[((size_t)l * 256 + 32 * t + r) * 256 + 16 * u + 8 * (j >> 2) + 4 * h + (j & 3)]);
          }
  // SHAPE 16: [layer][pair s][tile tt of the pair][k-step ks of 32][lane (row, g)][8]: W[32 s + 16 tt + row][32 ks + pi32(g, j)]
  for (int l = 0; l < L; ++l)
    for (int sp = 0; sp < 8; ++sp)
      for (int tt = 0; tt < 2; ++tt)
        for (int ks = 0; ks < 8; ++ks)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int row = lane & 15, g = lane >> 4;
              const int k = 32 * ks + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
              p16[(((((size_t)l * 8 + sp) * 2 + tt) * 8 + ks) * 64 + lane) * 8 + j] =
                  bf16_bits(W[((size_t)l * 256 + 32 * sp + 16 * tt + row) * 256 + k]);
            }
  const int nblk = (P + 31) / 32;
  char *d_w32, *d_w16, *d_s; float* d_b; uint32_t* d_m; __bf16* d_o;
  CK(hipMalloc(&d_w32, p32.size() * 2)); CK(hipMemcpy(d_w32, p32.data(), p32.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_w16, p16.size() * 2)); CK(hipMemcpy(d_w16, p16.data(), p16.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_b, B.size() * 4)); CK(hipMemcpy(d_b, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_s, (size_t)L * (nblk + 8) * 8 * 2048));
  CK(hipMalloc(&d_m, (size_t)L * (nblk + 8) * 256 * 4));
  CK(hipMalloc(&d_o, (size_t)(P + 512) * 256 * 2));
  Args a32{d_w32, d_b, d_s, d_m, d_o, P, L}, a16{d_w16, d_b, d_s, d_m, d_o, P, L};
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
    printf("   check %-12s max|err| = %.4g (max|ref| = %.4g) %s\n", nm, maxerr, maxref, maxerr <= 0.03 * maxref ? "ok" : "MISMATCH");
    return maxerr <= 0.03 * maxref;
  };
  std::vector<uint16_t> o32, o16;
  bool ok = true;
  run<32, true>("32x32x16", a32, 2, &o32); ok &= check(o32, "32x32x16");
  run<16, true>("16x16x32", a16, 2, &o16); ok &= check(o16, "16x16x32");
  { double md = 0; for (size_t i = 0; i < o32.size(); ++i) md = fmax(md, fabs(bf16_to_f(o32[i]) - bf16_to_f(o16[i])));
    printf("   16x16x32 vs 32x32x16: max |diff| over all %zu outputs = %.4g\n", o32.size(), md); }
  for (int rep = 0; rep < 4; ++rep) {        // interleaved: the boxes drift, the first launches run at other clocks
    run<32, true>("32x32x16 training epilogue", a32, reps, nullptr);
    run<16, true>("16x16x32 training epilogue", a16, reps, nullptr);
    run<32, false>("32x32x16 inference", a32, reps, nullptr);
    run<16, false>("16x16x32 inference", a16, reps, nullptr);
  }
  printf(ok ? "check ok\n" : "CHECK FAILED\n");
  return ok ? 0 : 1;
}
