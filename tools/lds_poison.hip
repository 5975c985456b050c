// Test infrastructure (tools/poison_probe.py): fills the LDS of every CU with a bit pattern.  LDS is not cleared
// between kernels, so whatever the machine kernels read from LDS without having written it (a staging plane of a
// component nobody published yet, a ring stage that has not landed) then reads as NaN / 3.4e38 instead of the
// plausible leftovers of the previous launch.  Round 3 found such a read by accident (padding features multiplying a
// stale plane of an unpublished component: 0 * NaN); this makes the hunt systematic.
//   hipcc --offload-arch=gfx950 -O2 -fPIC -shared -o tools/liblds_poison.so tools/lds_poison.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(1024) void lds_poison_kernel(uint32_t pattern, int words, unsigned* sink) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = pattern;
  __syncthreads();
  // keep the stores alive and the workgroup resident for a moment so that every CU gets one
  unsigned acc = 0;
  for (int i = threadIdx.x; i < words; i += blockDim.x) acc ^= lds[i];
  for (int k = 0; k < 2000; ++k) acc = acc * 1664525u + 1013904223u;
  if (acc == 0x12345678u && sink != nullptr) sink[0] = acc;
}

extern "C" int lds_poison(uint32_t pattern, int rounds, void* stream) {
  const int bytes = 160 * 1024;      // the whole LDS of a CU: one workgroup per CU at a time
  (void)hipFuncSetAttribute((const void*)lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  for (int r = 0; r < rounds; ++r)
    hipLaunchKernelGGL(lds_poison_kernel, dim3(256 * 4), dim3(1024), bytes, (hipStream_t)stream, pattern, bytes / 4,
                       (unsigned*)nullptr);
  return (int)hipGetLastError();
}
