// Diagnostic: lane layout of ds_read_b64_tr_b8 on gfx950 (no local documentation).  Lane L supplies LDS address 8 L
// (64 lanes x 8 bytes = 512 contiguous bytes); every byte of the result is printed as the LDS byte offset it came from.
// Measured (gpurun_out/tr_b8_probe.txt, round 3): per group of 16 lanes the 16 x 8 supplied bytes form 8 rows of 16
// bytes, row k = the 8 bytes of lane 2k followed by the 8 bytes of lane 2k+1; lane i of the group receives column i:
// result byte b = byte (i & 7) of the piece supplied by lane 2b + (i >> 3).  I.e. an 8 x 16 byte transposition — one
// read hands a lane its MFMA K-run of 8 consecutive rows (points) for its column (feature), the 8-bit analogue of what
// DwFrag<true> does with four ds_read_b64_tr_b16.
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/tr_b8_probe tools/tr_b8_probe.hip ; run: tools/tr_b8_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__global__ void probe(uint8_t* out_lo, uint8_t* out_hi, int stride) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
  const int lane = threadIdx.x;
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = lane; i < 4096; i += 64) lds[i] = pass == 0 ? (uint8_t)(i & 255) : (uint8_t)(i >> 8);
    __syncthreads();
    u32x2 v;
    const unsigned addr = (unsigned)(uintptr_t)lds + lane * stride;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    uint8_t* o = pass == 0 ? out_lo : out_hi;
    for (int b = 0; b < 8; ++b) o[lane * 8 + b] = (uint8_t)(v[b >> 2] >> (8 * (b & 3)));
    __syncthreads();
  }
}

int main() {
  uint8_t *lo, *hi;
  hipMalloc(&lo, 512);
  hipMalloc(&hi, 512);
  for (int stride : {8, 16, 64}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, lo, hi, stride);
    uint8_t hl[512], hh[512];
    hipMemcpy(hl, lo, 512, hipMemcpyDeviceToHost);
    hipMemcpy(hh, hi, 512, hipMemcpyDeviceToHost);
    printf("== lane address = %d * lane; result byte b of lane L <- LDS byte offset\n", stride);
    for (int L = 0; L < 64; ++L) {
      printf("lane %2d:", L);
      for (int b = 0; b < 8; ++b) printf(" %4d", hl[L * 8 + b] | hh[L * 8 + b] << 8);
      printf("\n");
    }
  }
  return 0;
}
