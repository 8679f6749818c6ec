// hbm_calib.hip -- two calibrations the bench line and the PMC summaries lean on
// (VERDICT r4 "weak" #4 / "do this" #3), both on a KNOWN byte count:
//
//  (1) the copy ceiling of this box with hand-written kernels instead of Tensor.copy_:
//      k_cal_read  (16 B per lane, read only, a checksum keeps the loads alive),
//      k_cal_write (16 B per lane, write only), k_cal_copy (1:1) -- each over a buffer
//      far larger than L2 + Infinity Cache, grid-stride with 8 loads in flight per lane;
//  (2) the FETCH_SIZE / WRITE_SIZE reading for the ROW KERNEL's access pattern:
//      k_cal_rowlike reads a "payload" the way k_dec_row_fused's bit reader does (every
//      lane a dword at a lane-strided offset of ~36 bytes, advancing a dword per step,
//      clamped to the row) and stores 2 x 16 bytes per lane and pixel row the way
//      transform_store_pair does (lane pair (l, l + 32) -> tile, 8 pixel rows 16 KiB
//      apart).  Its byte counts are printed; run it under rocprofv3 --pmc FETCH_SIZE /
//      WRITE_SIZE (separate passes) and divide: that ratio is the correction factor for
//      byte-granular kernels in tools/summarize_pmc.py (instead of the hand-kept list).
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/hbm_calib.hip -o tools/micro/hbm_calib
// Run:   tools/micro/hbm_calib [GiB per buffer = 4]      (prints one JSON line)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kUnroll = 8;

__global__ __launch_bounds__(256) void k_cal_read(const uint4 *__restrict__ src, size_t n16, uint32_t *sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (; i + (kUnroll - 1) * stride < n16; i += kUnroll * stride) {
    uint4 v[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) v[k] = src[i + k * stride];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  }
  for (; i < n16; i += stride) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;   // (never true for the test pattern: no store traffic)
}

__global__ __launch_bounds__(256) void k_cal_write(uint4 *__restrict__ dst, size_t n16, uint32_t seed) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i < n16; i += stride) {
    uint4 v;
    v.x = seed + (uint32_t)i; v.y = seed; v.z = ~seed; v.w = (uint32_t)(i >> 3);
    dst[i] = v;
  }
}

__global__ __launch_bounds__(256) void k_cal_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (kUnroll - 1) * stride < n16; i += kUnroll * stride) {
    uint4 v[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) v[k] = src[i + k * stride];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) dst[i + k * stride] = v[k];
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}

// One 1024-lane workgroup per "block row": payload bytes [row * pay_stride, + pay_len) read as
// the bit reader reads them, 8 pixel rows x 4096 px x 4 B written as the transform writes them.
__global__ __launch_bounds__(1024) void k_cal_rowlike(const uint8_t *__restrict__ pay, uint32_t pay_len, size_t pay_stride,
                                                      uint8_t *__restrict__ out, int width_px, uint32_t *sink) {
  const int tid = threadIdx.x;
  const uint32_t *w = reinterpret_cast<const uint32_t *>(pay + (size_t)blockIdx.x * pay_stride);
  const uint32_t nw = pay_len / 4u;
  // every lane owns pay_len / 1024 bytes (unaligned to dwords, like a sub-sequence) and reads
  // them a dword per step, plus the two dwords of window look-ahead the reader keeps
  const uint32_t b0 = (uint32_t)(((unsigned long long)pay_len * tid) / 1024u), b1 = (uint32_t)(((unsigned long long)pay_len * (tid + 1)) / 1024u);
  uint32_t acc = 0;
  for (uint32_t j = b0 / 4u; j <= b1 / 4u + 2u; ++j) acc ^= w[j < nw ? j : nw - 1u];
  // pixels: lane pair (l, l + 32) of a wave shares a tile; lane half s writes pixel rows 4s..4s+3
  const int cols = width_px / 8;
  const int it = tid, u = (it >> 6) * 32 + (it & 31), s = (it >> 5) & 1;
  uint8_t *img = out + (size_t)blockIdx.x * 8u * (size_t)width_px * 4u;
  if (u < cols) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      uint4 o0, o1;
      o0.x = acc + rr; o0.y = acc; o0.z = tid; o0.w = rr;
      o1 = o0; o1.x ^= 0x55u;
      uint8_t *dst = img + ((size_t)(4 * s + rr) * width_px + 8u * u) * 4u;
      reinterpret_cast<uint4 *>(dst)[0] = o0;
      reinterpret_cast<uint4 *>(dst)[1] = o1;
    }
  }
  if (acc == 0x12345678u) sink[1] = acc;
}

static float time_ms(hipStream_t s, int reps, const std::function<void()> &f) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  f();   // warm-up
  CHECK(hipStreamSynchronize(s));
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(a, s));
    f();
    CHECK(hipEventRecord(b, s));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char **argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 4.0;
  const size_t bytes = (size_t)(gib * 1024.0 * 1024.0 * 1024.0) & ~(size_t)4095;
  const size_t n16 = bytes / 16;
  uint4 *a, *b;
  uint32_t *sink;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(a, 0x5a, bytes)); CHECK(hipMemset(b, 0, bytes));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  double best_r = 0, best_w = 0, best_c = 0;
  int gr = 0, gw = 0, gc = 0;
  for (int per_cu : {4, 8, 16, 32}) {
    const int grid = cus * per_cu;
    const float tr = time_ms(s, 5, [&] { hipLaunchKernelGGL(k_cal_read, dim3(grid), dim3(256), 0, s, a, n16, sink); });
    const float tw = time_ms(s, 5, [&] { hipLaunchKernelGGL(k_cal_write, dim3(grid), dim3(256), 0, s, b, n16, 7u); });
    const float tc = time_ms(s, 5, [&] { hipLaunchKernelGGL(k_cal_copy, dim3(grid), dim3(256), 0, s, a, b, n16); });
    const double r = bytes / (tr * 1e6), w = bytes / (tw * 1e6), c = 2.0 * bytes / (tc * 1e6);
    if (r > best_r) { best_r = r; gr = per_cu; }
    if (w > best_w) { best_w = w; gw = per_cu; }
    if (c > best_c) { best_c = c; gc = per_cu; }
  }
  // hipMemcpyDtoD for comparison (what Tensor.copy_ is)
  const float tm = time_ms(s, 5, [&] { CHECK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, s)); });
  // row-like pattern: 4096 px rows, q50 payload (~33.6 KB per row), `rows` workgroups
  const int width = 4096;
  const uint32_t pay_len = 33600;
  const size_t pay_stride = 33664;
  const size_t row_bytes = (size_t)8 * width * 4;
  const int rows = (int)(bytes / row_bytes);
  const float trl = time_ms(s, 3, [&] {
    hipLaunchKernelGGL(k_cal_rowlike, dim3(rows), dim3(1024), 0, s, reinterpret_cast<const uint8_t *>(a), pay_len, pay_stride,
                       reinterpret_cast<uint8_t *>(b), width, sink);
  });
  printf("{\"buffer_GiB\": %.2f, \"cus\": %d, \"read_GBs\": %.0f, \"read_wg_per_cu\": %d, \"write_GBs\": %.0f, \"write_wg_per_cu\": %d, "
         "\"copy_GBs_rd_plus_wr\": %.0f, \"copy_wg_per_cu\": %d, \"memcpy_dtod_GBs_rd_plus_wr\": %.0f, "
         "\"rowlike\": {\"rows\": %d, \"payload_bytes_read\": %.0f, \"pixel_bytes_written\": %.0f, \"ms\": %.3f, \"GBs\": %.0f}}\n",
         gib, cus, best_r, gr, best_w, gw, best_c, gc, 2.0 * bytes / (tm * 1e6),
         rows, (double)rows * pay_len, (double)rows * row_bytes, trl, ((double)rows * (pay_len + row_bytes)) / (trl * 1e6));
  return 0;
}
