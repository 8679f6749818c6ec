// Micro-benchmark: cost of LDS store flavours on gfx950 (cycles per wave-store).
// hipcc --offload-arch=gfx950 -O3 -o lds_store_bench lds_store_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct __attribute__((packed)) PU32 { uint32_t v; };
template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t *out, long long *cyc, int stride, int iters) {
  extern __shared__ uint8_t lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 131072 / 4; i += 1024) reinterpret_cast<uint32_t *>(lds)[i] = 0;
  __syncthreads();
  uint32_t op = (uint32_t)tid * (uint32_t)stride;
  uint32_t v = tid * 2654435761u;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    v = v * 1664525u + 1013904223u;
    const uint32_t a = (op + (v >> 28)) & 131071u;   // pseudo-random small advance
    if (MODE == 0) reinterpret_cast<uint32_t *>(lds)[(a & ~3u) >> 2] = v;            // aligned dword
    if (MODE == 1) reinterpret_cast<PU32 *>(lds + (a < 131068u ? a : 131068u))->v = v;  // unaligned dword
    if (MODE == 2) lds[a] = (uint8_t)v;                                               // one byte
    if (MODE == 3) { if (v & 0x100) lds[a] = (uint8_t)v; if (v & 0x200) lds[(a + 1) & 131071u] = (uint8_t)(v >> 8);
                     if (v & 0x400) lds[(a + 2) & 131071u] = (uint8_t)(v >> 16); if (v & 0x800) lds[(a + 3) & 131071u] = (uint8_t)(v >> 24); }
    op = a + 3;
  }
  const long long t1 = clock64();
  __syncthreads();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 1024 + tid] = reinterpret_cast<uint32_t *>(lds)[tid];
}
int main() {
  uint32_t *out; long long *cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 1000;
  const char *names[] = {"aligned b32", "unaligned b32", "b8", "4x cond b8"};
  for (int stride : {128, 131, 4}) for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) { hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); k<0><<<256, 1024, 131072>>>(out, cyc, stride, iters); }
      if (mode == 1) { hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); k<1><<<256, 1024, 131072>>>(out, cyc, stride, iters); }
      if (mode == 2) { hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); k<2><<<256, 1024, 131072>>>(out, cyc, stride, iters); }
      if (mode == 3) { hipFuncSetAttribute((const void *)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); k<3><<<256, 1024, 131072>>>(out, cyc, stride, iters); }
      hipDeviceSynchronize();
    }
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %4d  %-14s %8.1f cycles per iteration (16 waves/CU)\n", stride, names[mode], (double)h[0] / iters);
  }
  return 0;
}
