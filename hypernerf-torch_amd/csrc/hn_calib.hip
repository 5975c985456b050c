// Box calibration probes for bench.py (round 5): the boxes of a pool differ by several per cent in sustained clock and
// HBM rate, and a training step runs against the board's power cap — a throughput number without the box's own
// ceilings next to it cannot show a 3-5 % change.  Two short kernels, measured in the same process right before the
// timed region:
//   hn_calib_mfma    register-resident v_mfma_f32_32x32x16_bf16 on every SIMD (8 independent accumulators per wave, no
//                    memory traffic): sustained dense bf16 TFLOP/s of THIS box under power management, and the shader
//                    clock it sustains doing so (s_memtime ticks per s_memrealtime tick x 100 MHz).
//   hn_calib_stream  every CU streams a buffer once through LDS-DMA (global_load_lds_dwordx4, nt), the access pattern
//                    of hn_wgrad_kernel without its products: the HBM rate that kernel can reach at best on this box.
// Both time themselves with the 100 MHz wall clock like the machine kernels' timeline (hn_common.h).
#include "hn_common.h"

__global__ __launch_bounds__(512, 2) void hn_calib_mfma_kernel(int iters, float* sink, uint64_t* t) {
  hn_timeline_begin(t);
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  // operands with the toggle rate of real activations / weights (hashed mantissas and signs, magnitudes around 1e-2):
  // all-zero or constant operands draw visibly less power and would report a clock the step never sees
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned ha = (unsigned)(lane * 8 + j + 1) * 2654435761u, hb = (unsigned)(lane * 8 + j + 77) * 2246822519u;
    a[j] = (__bf16)(((float)(ha >> 16) - 32768.0f) * (1.0f / 2097152.0f));
    b[j] = (__bf16)(((float)(hb >> 16) - 32768.0f) * (1.0f / 2097152.0f));
  }
  f32x16 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[k][q] = 0.0f;
  const uint64_t c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = hn_mfma_bf16(a, b, acc[k]);
  }
  const uint64_t c1 = clock64(), w1 = wall_clock64();
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][15];
  if (s == 123.456f) sink[0] = s;                       // keeps the products alive
  if (blockIdx.x == 0 && threadIdx.x == 0) {            // shader-clock ticks and wall ticks of one wave's loop
    t[8] = c1 - c0;
    t[9] = w1 - w0;
  }
  hn_timeline_end(t);
}

template <int PATTERN>
__global__ __launch_bounds__(512, 2) void hn_calib_stream_kernel(const char* __restrict__ buf, long long n_kib,
                                                                 float* sink, uint64_t* t) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  hn_timeline_begin(t);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // a workgroup takes 64-KiB pieces round-robin; each wave moves 8 consecutive 1-KiB units of a piece per pass into
  // its own slice of a 2 x 64 KiB ring (nothing reads the LDS: the probe measures the fetch path only)
  // PATTERN 0: the workgroups share one moving window (piece = blockIdx + k * grid): every CU reads next to every other.
  // PATTERN 1: each workgroup streams a region of its own (1/grid of the buffer, front to back), as the jobs of
  //            hn_wgrad_kernel do: 256 far-apart sequential streams.
  const long long pieces = n_kib / 64;
  const long long per_wg = pieces / gridDim.x;
  int flip = 0;
  for (long long k = 0; k < (PATTERN == 0 ? (pieces - blockIdx.x + gridDim.x - 1) / gridDim.x : per_wg); ++k) {
    const long long pc = PATTERN == 0 ? blockIdx.x + k * gridDim.x : blockIdx.x * per_wg + k;
    const char* src = buf + pc * 65536 + (long long)wave * 8192;
    char* dst = smem + flip * 65536 + wave * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 2);
    flip ^= 1;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // one pass in flight behind the one being issued
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && reinterpret_cast<float*>(smem)[0] == 123.456f) sink[0] = 1.0f;
  hn_timeline_end(t);
}

// The DMA protocol of hn_wgrad_kernel without its products (PROTO 0): a ring of STAGES stages of 32 KiB, every wave issues
// its 4 pieces of a stage, waits (counted vmcnt) for the oldest stage, all waves meet at a barrier, the freed buffer is
// refilled.  PROTO 1: the same ring without the barrier (each wave waits for its own pieces only — what the stream costs
// when nothing is shared).  Each workgroup streams a contiguous region of its own.
template <int STAGES, int PROTO>
__global__ __launch_bounds__(512, 2) void hn_calib_ring_kernel(const char* __restrict__ buf, long long n_kib,
                                                               float* sink, uint64_t* t) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  hn_timeline_begin(t);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long nstage = (n_kib / 32) / gridDim.x;            // stages of 32 KiB per workgroup
  const char* base = buf + (long long)blockIdx.x * nstage * 32768;
  auto issue = [&](long long s) {
    const char* src = base + s * 32768;
    char* dst = smem + (s % STAGES) * 32768;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int unit = wave + 8 * i;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + unit * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + unit * 1024), 16, 0, 2);
    }
  };
  for (int s0 = 0; s0 < STAGES - 1 && s0 < nstage; ++s0) issue(s0);
  for (long long s = 0; s < nstage; ++s) {
    // stages issued after s that may stay in flight: STAGES - 2 (4 pieces each)
    if (STAGES == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (STAGES == 5) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (PROTO == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (s + STAGES - 1 < nstage) issue(s + STAGES - 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && reinterpret_cast<float*>(smem)[0] == 123.456f) sink[0] = 1.0f;
  hn_timeline_end(t);
}

extern "C" int hn_calib_ring(const void* buf_dev, long long n_bytes, int stages, int proto, float* sink_dev,
                             uint64_t* t_dev, hnStream_t stream) {
  if (n_bytes < 32768LL * 256 || (stages != 3 && stages != 4 && stages != 5) || (proto != 0 && proto != 1)) return -2;
  if (buf_dev == nullptr || sink_dev == nullptr || t_dev == nullptr) return -3;
#define HN_RING(S, P)                                                                                                    \
  {                                                                                                                      \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hn_calib_ring_kernel<S, P>),                                 \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                   \
    hipLaunchKernelGGL((hn_calib_ring_kernel<S, P>), dim3(256), dim3(512), S * 32768, (hipStream_t)stream,               \
                       (const char*)buf_dev, n_bytes / 1024, sink_dev, t_dev);                                           \
  }
  if (stages == 3 && proto == 0) HN_RING(3, 0)
  else if (stages == 3) HN_RING(3, 1)
  else if (stages == 4 && proto == 0) HN_RING(4, 0)
  else if (stages == 4) HN_RING(4, 1)
  else if (proto == 0) HN_RING(5, 0)
  else HN_RING(5, 1)
#undef HN_RING
  HN_CHECK_LAUNCH();
  return 0;
}

extern "C" int hn_calib_mfma(int iters, float* sink_dev, uint64_t* t_dev, hnStream_t stream) {
  if (iters <= 0) return -2;
  if (sink_dev == nullptr || t_dev == nullptr) return -3;
  // two 512-thread workgroups per CU: 4 waves per SIMD, the matrix pipe never waits for an issuer
  hipLaunchKernelGGL(hn_calib_mfma_kernel, dim3(512), dim3(512), 0, (hipStream_t)stream, iters, sink_dev, t_dev);
  HN_CHECK_LAUNCH();
  return 0;
}

extern "C" int hn_calib_stream(const void* buf_dev, long long n_bytes, float* sink_dev, uint64_t* t_dev,
                               hnStream_t stream) {
  return hn_calib_stream_pattern(buf_dev, n_bytes, 0, sink_dev, t_dev, stream);
}

extern "C" int hn_calib_stream_pattern(const void* buf_dev, long long n_bytes, int pattern, float* sink_dev,
                                       uint64_t* t_dev, hnStream_t stream) {
  if (n_bytes < 65536 * 256 || (pattern != 0 && pattern != 1)) return -2;
  if (buf_dev == nullptr || sink_dev == nullptr || t_dev == nullptr) return -3;
  static bool allowed = false;
  if (!allowed) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hn_calib_stream_kernel<0>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hn_calib_stream_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    allowed = true;
  }
  if (pattern == 0)
    hipLaunchKernelGGL(hn_calib_stream_kernel<0>, dim3(256), dim3(512), 128 * 1024, (hipStream_t)stream,
                       (const char*)buf_dev, n_bytes / 1024, sink_dev, t_dev);
  else
    hipLaunchKernelGGL(hn_calib_stream_kernel<1>, dim3(256), dim3(512), 128 * 1024, (hipStream_t)stream,
                       (const char*)buf_dev, n_bytes / 1024, sink_dev, t_dev);
  HN_CHECK_LAUNCH();
  return 0;
}
