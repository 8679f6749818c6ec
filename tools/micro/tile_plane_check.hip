// Unit check of tile_plane (kernels_dec.hip) against a scalar host model.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ihimg_amd/csrc tools/micro/tile_plane_check.hip -o tools/micro/tile_plane_check
#include "../../himg_amd/csrc/kernels_dec.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace himg_dev;
namespace himg_dev {
void prof_begin(Profiler *, const char *, hipStream_t) {}
void prof_end(Profiler *, hipStream_t) {}
}

// VARIANT 0: the generic gather (run-time tile count); -1: the identity-range gather with
// the run-time count; 64: the same with the count at compile time.
template <int VARIANT>
__global__ void k_check(const uint8_t *codes, const int16_t *unmap, const uint8_t *shift, const uint32_t *lr,
                        uint32_t *out, int n, uint32_t hf0, uint32_t hf1) {
  __shared__ int16_t s_unmap[256];
  __shared__ uint8_t s_shift[64];
  __shared__ uint32_t s_shiftp[32];
  __shared__ uint32_t s_hfast[2];
  __shared__ uint8_t s_sym[64 * 64];
  const int t = threadIdx.x;
  for (int k = t; k < 256; k += 64) s_unmap[k] = unmap[k];
  s_shift[t] = shift[t];
  if (t < 32) {
    const int x = t >> 2, j = t & 3;
    s_shiftp[t] = (uint32_t)shift[(2 * j) * 8 + x] | ((uint32_t)shift[(2 * j + 1) * 8 + x] << 16);
  }
  for (int k = 0; k < 64; ++k) s_sym[k * 64 + t] = codes[((size_t)blockIdx.x * 64 + k) * 64 + t];
  __syncthreads();
  uint32_t O[16];
  const int id = blockIdx.x * 64 + t;
  if (t == 0) { s_hfast[0] = hf0; s_hfast[1] = hf1; }
  __syncthreads();
  tile_plane<VARIANT>(s_sym + t, 64, s_unmap, s_shift, s_shiftp, lr[2 * id], lr[2 * id + 1], O, VARIANT ? s_hfast : nullptr);
  for (int i = 0; i < 16; ++i) out[(size_t)id * 16 + i] = O[i];
}

__global__ void k_lq(const uint32_t *lr, uint32_t *out) {
  const int id = blockIdx.x * 64 + threadIdx.x;
  uint32_t LQ[2][8];
  lowres_quads(lr[2 * id], lr[2 * id + 1], LQ);
  for (int i = 0; i < 16; ++i) out[(size_t)id * 16 + i] = LQ[i >> 3][i & 7];
}

static void h_iwht8(int *x) {
  int a[8] = {x[0] + x[4], x[1] + x[5], x[2] + x[6], x[3] + x[7], x[0] - x[4], x[1] - x[5], x[2] - x[6], x[3] - x[7]};
  int b[8] = {a[0] + a[2], a[1] + a[3], a[0] - a[2], a[1] - a[3], a[4] + a[6], a[5] + a[7], a[4] - a[6], a[5] - a[7]};
  int o[8] = {b[0] + b[1], b[4] + b[5], b[6] + b[7], b[2] + b[3], b[2] - b[3], b[6] - b[7], b[4] - b[5], b[0] - b[1]};
  for (int i = 0; i < 8; ++i) x[i] = (int16_t)(o[i] >> 3);
}
static void h_interp(int *a) {
  a[4] = (a[0] + a[8] + 1) >> 1; a[2] = (a[0] + a[4] + 1) >> 1; a[6] = (a[4] + a[8] + 1) >> 1;
  a[1] = (a[0] + a[2] + 1) >> 1; a[3] = (a[2] + a[4] + 1) >> 1; a[5] = (a[4] + a[6] + 1) >> 1; a[7] = (a[6] + a[8] + 1) >> 1;
}

int main(int argc, char **argv) {
  const int blocks = 256, n = blocks * 64;
  const int amp = argc > 1 ? atoi(argv[1]) : 40;       // unmap amplitude scale
  const int zero_pct = argc > 2 ? atoi(argv[2]) : 60;
  srand(12345);
  std::vector<uint8_t> codes((size_t)n * 64), shift(64);
  std::vector<int16_t> unmap(256);
  std::vector<uint32_t> lr(2 * n), out((size_t)n * 16);
  const int mode = argc > 3 ? atoi(argv[3]) : 0;
  for (int k = 0; k < 256; ++k) { int sc = (int8_t)k; unmap[k] = (int16_t)(sc * amp); }
  for (int k = 0; k < 64; ++k) shift[k] = rand() % 5;
  for (auto &c : codes) c = (rand() % 100 < zero_pct) ? 0 : (uint8_t)(rand() & 255);
  if (mode == 2)   // small codes only (inside the identity range of an identity table): the fast gather is taken
    for (auto &c : codes) c = (rand() % 100 < zero_pct) ? 0 : (uint8_t)(int8_t)(rand() % 81 - 40);
  if (mode == 1) {
    // The edges of packed_wht_exact: every coefficient at the largest magnitude one of
    // its two conditions allows, or one beyond it -- |d| = 3071 / 3072 / 4095 / 4096
    // outside register 0 (positions (0,0) and (1,0)), 11263 / 11264 / 4095 / 4096 there --
    // with all signs equal (the largest butterfly sums), alternating, or random.
    const int16_t mags[10] = {0, 3071, 3072, 4095, 4096, 11263, 11264, 2047, 16383, 1};
    for (int k = 0; k < 256; ++k) { const int sc = (int8_t)k; const int m = abs(sc) < 10 ? mags[abs(sc)] : 7; unmap[k] = (int16_t)(sc < 0 ? -m : m); }
    for (int k = 0; k < 64; ++k) shift[k] = 0;
    for (int id = 0; id < n; ++id) {
      const int blk = id / 64, t = id % 64;
      const int other = 1 + rand() % 4, dc = (rand() % 3 == 0) ? 1 + rand() % 4 : 5 + rand() % 2, signs = rand() % 4;
      for (int k = 0; k < 64; ++k) {
        const int pos = kScanD[k];
        int code = (pos == 0 || pos == 8) ? dc : other;
        if (rand() % 16 == 0) code = (pos == 0 || pos == 8) ? 8 : 7;   // a few smaller / larger ones
        const bool neg = signs == 0 ? false : signs == 1 ? true : signs == 2 ? ((pos ^ (pos >> 3)) & 1) : (rand() & 1);
        codes[((size_t)blk * 64 + k) * 64 + t] = (uint8_t)(neg ? -code : code);
      }
    }
  }
  for (auto &v : lr) v = (uint32_t)(rand() & 0xffff);
  uint8_t *d_codes, *d_shift; int16_t *d_unmap; uint32_t *d_lr, *d_out;
  hipMalloc(&d_codes, codes.size()); hipMalloc(&d_shift, 64); hipMalloc(&d_unmap, 512);
  hipMalloc(&d_lr, lr.size() * 4); hipMalloc(&d_out, out.size() * 4);
  hipMemcpy(d_codes, codes.data(), codes.size(), hipMemcpyHostToDevice);
  hipMemcpy(d_shift, shift.data(), 64, hipMemcpyHostToDevice);
  hipMemcpy(d_unmap, unmap.data(), 512, hipMemcpyHostToDevice);
  hipMemcpy(d_lr, lr.data(), lr.size() * 4, hipMemcpyHostToDevice);
  // The identity-test words, by the rule of identity_test_words (kernels_dec.hip).
  const int variant = argc > 4 ? atoi(argv[4]) : 0;
  uint32_t hf0, hf1;
  {
    int nid = 127;
    for (int i = 0; i < 128; ++i) if (unmap[i] != i) { nid = i - 1; break; }
    int smax = 0;
    for (int y = 2; y < 8; ++y) for (int x = 1; x < 8; ++x) smax = shift[y * 8 + x] > smax ? shift[y * 8 + x] : smax;
    int B = 0;
    if (nid >= 1) { B = 1; while (2 * B <= nid) B *= 2; while (B && ((long long)B << smax) > 2048) B >>= 1; }
    const uint32_t b16 = B ? (uint32_t)B : 0x4000u, m16 = B ? (uint32_t)(0xffffu & ~(2u * B - 1u)) : 0xffffu;
    hf0 = b16 | (b16 << 16); hf1 = m16 | (m16 << 16);
    printf("variant %d, identity range %d, B %d\n", variant, nid, B);
  }
  if (variant == 0) hipLaunchKernelGGL(k_check<0>, dim3(blocks), dim3(64), 0, 0, d_codes, d_unmap, d_shift, d_lr, d_out, n, hf0, hf1);
  else if (variant < 0) hipLaunchKernelGGL(k_check<-1>, dim3(blocks), dim3(64), 0, 0, d_codes, d_unmap, d_shift, d_lr, d_out, n, hf0, hf1);
  else hipLaunchKernelGGL(k_check<64>, dim3(blocks), dim3(64), 0, 0, d_codes, d_unmap, d_shift, d_lr, d_out, n, hf0, hf1);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
  hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
  {
    std::vector<uint32_t> lq((size_t)n * 16);
    uint32_t *d_lq; hipMalloc(&d_lq, lq.size() * 4);
    hipLaunchKernelGGL(k_lq, dim3(blocks), dim3(64), 0, 0, d_lr, d_lq);
    hipMemcpy(lq.data(), d_lq, lq.size() * 4, hipMemcpyDeviceToHost);
    long lbad = 0; int lh[64] = {0};
    for (int id = 0; id < n; ++id) {
      int left[9], right[9];
      left[0] = lr[2 * id] & 255; right[0] = (lr[2 * id] >> 8) & 255;
      left[8] = lr[2 * id + 1] & 255; right[8] = (lr[2 * id + 1] >> 8) & 255;
      h_interp(left); h_interp(right);
      for (int y = 0; y < 8; ++y) {
        int a[9]; a[0] = left[y]; a[8] = right[y]; h_interp(a);
        for (int x = 0; x < 8; ++x) {
          const int got = (lq[(size_t)id * 16 + (y >> 2) * 8 + x] >> ((y & 3) * 8)) & 255;
          if (got != a[x]) { if (lbad < 4) printf("LQ id %d y %d x %d got %d want %d (l %d r %d)\n", id, y, x, got, a[x], left[y], right[y]); ++lbad; ++lh[y * 8 + x]; }
        }
      }
    }
    printf("lowres_quads mismatches %ld\n", lbad);
    if (lbad) for (int y = 0; y < 8; ++y) { for (int x = 0; x < 8; ++x) printf("%6d", lh[y * 8 + x]); printf("\n"); }
  }
  long bad = 0, big = 0; int hist[64] = {0};
  for (int id = 0; id < n; ++id) {
    const int blk = id / 64, t = id % 64;
    int b[64]; bool large = false;
    for (int k = 0; k < 64; ++k) {
      const int pos = kScanD[k];
      const int code = codes[((size_t)blk * 64 + k) * 64 + t];
      b[pos] = (int16_t)((int)unmap[code] * (1 << shift[pos]));
      if (b[pos] > 4095 || b[pos] < -4096) large = true;
    }
    big += large;
    for (int y = 0; y < 8; ++y) h_iwht8(b + 8 * y);
    for (int x = 0; x < 8; ++x) { int c[8]; for (int y = 0; y < 8; ++y) c[y] = b[8 * y + x]; h_iwht8(c); for (int y = 0; y < 8; ++y) b[8 * y + x] = c[y]; }
    int left[9], right[9];
    left[0] = lr[2 * id] & 255; right[0] = (lr[2 * id] >> 8) & 255;
    left[8] = lr[2 * id + 1] & 255; right[8] = (lr[2 * id + 1] >> 8) & 255;
    h_interp(left); h_interp(right);
    for (int y = 0; y < 8; ++y) {
      int a[9]; a[0] = left[y]; a[8] = right[y]; h_interp(a);
      for (int x = 0; x < 8; ++x) {
        int v = (int16_t)(b[8 * y + x] + a[x]); v = v < 0 ? 0 : (v > 255 ? 255 : v);
        const int got = (out[(size_t)id * 16 + y * 2 + x / 4] >> ((x & 3) * 8)) & 255;
        if (got != v) { if (bad < 5) printf("id %d y %d x %d got %d want %d large %d\n", id, y, x, got, v, (int)large); ++bad; ++hist[y * 8 + x]; }
      }
    }
  }
  printf("planes %d (with a large coefficient: %ld), mismatching pixels %ld\n", n, big, bad);
  if (bad) { for (int y = 0; y < 8; ++y) { for (int x = 0; x < 8; ++x) printf("%6d", hist[y * 8 + x]); printf("\n"); } }
  return bad ? 1 : 0;
}
