// Does an LDS-DMA (global_load_lds_dwordx4) reach every KiB of a 160-KiB dynamic LDS allocation?  One workgroup, one wave:
// unit u of the source goes to LDS byte u * 1024 (wave-uniform destination -> M0), then LDS is read back with ds_read_b128.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/lds_dma_probe tools/lds_dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const char* src, char* out, int units, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  for (int u = 0; u < units; ++u) *reinterpret_cast<int4*>(smem + u * 1024 + lane * 16) = int4{-1, -1, -1, -1};
  __syncthreads();
  if (mode == 0) {
    for (int u = 0; u < units; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)u * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(smem + u * 1024), 16, 0, 0);
  } else {      // base + a large wave-uniform offset computed the way hn_wgrad_kernel does (stage base + (wave + 8 i) KiB)
    char* base = smem + (mode - 1) * 1024;
#pragma unroll
    for (int i = 0; i < 10; ++i)
      for (int w = 0; w < 8; ++w) {
        const int u = w + 8 * i;
        if (u + mode - 1 < units)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(u + mode - 1) * 1024 + lane * 16),
                                           (__attribute__((address_space(3))) void*)(base + u * 1024), 16, 0, 0);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int u = 0; u < units; ++u)
    *reinterpret_cast<int4*>(out + (size_t)u * 1024 + lane * 16) = *reinterpret_cast<const int4*>(smem + u * 1024 + lane * 16);
}
int main() {
  const int units = 160;
  std::vector<int> h(units * 256);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (int)i;
  char *src, *out;
  hipMalloc(&src, units * 1024); hipMalloc(&out, units * 1024);
  hipMemcpy(src, h.data(), units * 1024, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, units * 1024);
  for (int mode : {0, 1, 73, 81}) {
    hipMemset(out, 0, units * 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), units * 1024, 0, src, out, units, mode);
    hipError_t e = hipDeviceSynchronize();
    std::vector<int> r(units * 256);
    hipMemcpy(r.data(), out, units * 1024, hipMemcpyDeviceToHost);
    int first_bad = -1, bad = 0;
    for (int u = 0; u < units; ++u) {
      bool ok = true;
      for (int j = 0; j < 256; ++j) ok &= r[u * 256 + j] == h[u * 256 + j];
      const bool expected = mode == 0 || (u >= mode - 1 && u < mode - 1 + 80);
      if (expected && !ok) { ++bad; if (first_bad < 0) first_bad = u; }
    }
    printf("mode %d: %s, bad units %d, first bad KiB %d\n", mode, hipGetErrorString(e), bad, first_bad);
  }
  return 0;
}
