// kernels_enc.hip -- HIMG encode path as hand-written HIP for gfx950 (MI355X).
//
// Pipeline (all on device, batched over frames):
//   k_lowres_avg      colour lift + 8x8 box average          (ycbcr.cpp:24-52, downsampled.cpp:76-96)
//   k_lowres_blend    1/16-phase blend -> low-res plane      (downsampled.cpp:98-113)
//   k_lres_predict    predictor select + delta coding        (downsampled.cpp:177-316)
//   k_pix_fwd /       lift, low-res removal, WHT, quantize,  (encoder.cpp:258-327, hadamard.cpp:78-88,
//   k_tile_fwd        compand, coefficient-major scatter      quantize.cpp:127-151, mapper.cpp:159-182)
//                     (k_pix_fwd: full RGBA8 tiles, two channels per register in packed int16)
//   k_lres_summary    zero-run summaries of LRES spans
//   k_tok_hist        RLE tokenise + histogram per span      (huffman_enc.cpp:98-144)
//   k_tree            Huffman tree, codes, serialised tree   (huffman_enc.cpp:148-238)
//   k_sizes           span sizes -> offsets, container bytes (huffman_enc.cpp:342-352, encoder.cpp:111-256,337-353)
//   k_emit            RLE tokenise + bit pack                (huffman_enc.cpp:290-358)
//   k_padfix          stale pad bits of the reference's reused scratch buffer (huffman_enc.cpp:288,355; trap T1)
//
// Wavefront = 64 lanes everywhere; no MFMA (there is no dense contraction).
#include "himg_dev.h"
#include "loop_counts.h"

#include <cstdlib>
#include <utility>

namespace himg_dev {

// L-shell coefficient scan order (reference common.cpp:13-22; part of the format).
#define HIMG_SCAN_ORDER                                                  \
    0,  1,  9,  8,  16, 17, 18, 10, 2,  3,  11, 19, 27, 26, 25, 24,     \
    32, 33, 34, 35, 36, 28, 20, 12, 4,  5,  13, 21, 29, 37, 45, 44,     \
    43, 42, 41, 40, 48, 49, 50, 51, 52, 53, 54, 46, 38, 30, 22, 14,     \
    6,  7,  15, 23, 31, 39, 47, 55, 63, 62, 61, 60, 59, 58, 57, 56
__device__ static constexpr uint8_t kScan[64] = {HIMG_SCAN_ORDER};
static constexpr uint8_t kScanHost[64] = {HIMG_SCAN_ORDER};   // the same for host code

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ int clamp255(int x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }

// Colour lift of one pixel (ycbcr.cpp:32-37).
__device__ __forceinline__ void lift_fwd(int &c0, int &c1, int &c2) {
  const int y = (c0 + 2 * c1 + c2 + 2) >> 2;
  const int cb = (c2 - c1 + 256) >> 1;
  const int cr = (c0 - c1 + 256) >> 1;
  c0 = y; c1 = cb; c2 = cr;
}

// Load one pixel as up to four channel values, lifted when the frame is YCbCr coded.
__device__ __forceinline__ void load_pixel(const uint8_t *img, const Geom &g, int x, int y,
                                           int ch[4]) {
  const uint8_t *p = img + ((long long)y * g.W + x) * g.stride;
  if (g.stride == 4 && g.C == 4) {
    const uint32_t w = *reinterpret_cast<const uint32_t *>(p);
    ch[0] = w & 255; ch[1] = (w >> 8) & 255; ch[2] = (w >> 16) & 255; ch[3] = w >> 24;
  } else {
    ch[0] = p[0];
    ch[1] = g.C > 1 ? p[1] : 0;
    ch[2] = g.C > 2 ? p[2] : 0;
    ch[3] = g.C > 3 ? p[3] : 0;
  }
  if (g.ycbcr) lift_fwd(ch[0], ch[1], ch[2]);
}

// ---------------------------------------------------------------------------
// k_lowres_avg: one thread per tile window.  Window x in [8u-3, 8u+4], y in
// [8v-3, 8v+4], clipped to the image (downsampled.cpp:76-96).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lowres_avg(Geom g, const uint8_t *frames,
                                                    uint8_t *avg, size_t plane_stride, int v0) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.y + v0, f = blockIdx.z;
  if (u >= g.cols) return;
  const uint8_t *img = frames + (long long)f * g.frame_bytes;
  const int x0 = max(0, 8 * u - 3), x1 = min(g.W - 1, 8 * u + 4);
  const int y0 = max(0, 8 * v - 3), y1 = min(g.H - 1, 8 * v + 4);
  int sum[4] = {0, 0, 0, 0};
  if (g.stride == 4 && g.C == 4 && (g.W & 7) == 0) {
    // Packed RGBA8: the window spans pixels 8u-3 .. 8u+4 = the last three of the
    // 16-byte group before the tile, the tile's first group, and the first pixel
    // of its second group -> three 16-byte loads per pixel row instead of eight
    // 4-byte ones.
    const bool has_left = u > 0;
    for (int y = y0; y <= y1; ++y) {
      const uint4 *rp = reinterpret_cast<const uint4 *>(img + ((long long)y * g.W + 8 * u) * 4);
      const uint4 b = rp[0];
      const uint32_t c0 = rp[1].x;
      uint4 a;
      a.x = a.y = a.z = a.w = 0;
      if (has_left) a = rp[-1];
      uint32_t px[8] = {a.y, a.z, a.w, b.x, b.y, b.z, b.w, c0};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k < 3 && !has_left) continue;
        int ch[4] = {(int)(px[k] & 255), (int)((px[k] >> 8) & 255), (int)((px[k] >> 16) & 255),
                     (int)(px[k] >> 24)};
        if (g.ycbcr) lift_fwd(ch[0], ch[1], ch[2]);
        sum[0] += ch[0]; sum[1] += ch[1]; sum[2] += ch[2]; sum[3] += ch[3];
      }
    }
  } else {
    for (int y = y0; y <= y1; ++y)
      for (int x = x0; x <= x1; ++x) {
        int ch[4];
        load_pixel(img, g, x, y, ch);
        sum[0] += ch[0]; sum[1] += ch[1]; sum[2] += ch[2]; sum[3] += ch[3];
      }
  }
  const int cnt = (x1 - x0 + 1) * (y1 - y0 + 1);
  uint8_t *a = avg + (size_t)f * plane_stride;
  for (int c = 0; c < g.C; ++c)
    a[((size_t)c * g.rows + v) * g.cols + u] = (uint8_t)((sum[c] + (cnt >> 1)) / cnt);
}

// k_lowres_blend: m = blend of the averages at (v-1,v) x (u-1,u) (downsampled.cpp:98-113).
constexpr int kBlendRows = 8;    // block rows per k_lowres_blend workgroup (the kernel is launch-bound)
// QUAD: four columns per lane (cols a multiple of 4, planes dword aligned): one dword load
// per row and one byte for the column in front, a dword store -- the byte-per-lane form
// is bound by its instruction count, not by the 2 MiB it moves per frame.
template <bool QUAD>
__global__ __launch_bounds__(256) void k_lowres_blend(Geom g, const uint8_t *avg, uint8_t *low,
                                                      size_t plane_stride, int v0, int v1) {
  const int f = blockIdx.z / g.C, c = blockIdx.z % g.C;
  const uint8_t *a = avg + (size_t)f * plane_stride + (size_t)c * g.rows * g.cols;
  const int vb = v0 + (int)blockIdx.y * kBlendRows;
  if (QUAD) {
    const int u = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (u >= g.cols) return;
    uint8_t *o = low + (size_t)f * plane_stride + (size_t)c * g.rows * g.cols;
    auto row = [&](int r, uint32_t *w, uint32_t *left) {
      *w = *reinterpret_cast<const uint32_t *>(a + (size_t)r * g.cols + u);
      *left = u ? a[(size_t)r * g.cols + u - 1] : (*w & 255u);
    };
    for (int v = vb; v < min(vb + kBlendRows, v1); ++v) {
      uint32_t w1, l1, w2, l2;
      row(max(0, v - 1), &w1, &l1);
      row(v, &w2, &l2);
      uint32_t out = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int x12 = (w1 >> (8 * k)) & 255, x22 = (w2 >> (8 * k)) & 255;
        const int x11 = k ? (int)((w1 >> (8 * (k - 1))) & 255) : (int)l1, x21 = k ? (int)((w2 >> (8 * (k - 1))) & 255) : (int)l2;
        const int a1 = (x11 + 15 * x12 + 8) >> 4, a2 = (x21 + 15 * x22 + 8) >> 4;
        out |= (uint32_t)((a1 + 15 * a2 + 8) >> 4) << (8 * k);
      }
      *reinterpret_cast<uint32_t *>(o + (size_t)v * g.cols + u) = out;
    }
    return;
  }
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= g.cols) return;
  const int c1 = max(0, u - 1);
  for (int v = vb; v < min(vb + kBlendRows, v1); ++v) {
    const int r1 = max(0, v - 1);
    const int x11 = a[r1 * g.cols + c1], x12 = a[r1 * g.cols + u];
    const int x21 = a[v * g.cols + c1], x22 = a[v * g.cols + u];
    const int a1 = (x11 + 15 * x12 + 8) >> 4;
    const int a2 = (x21 + 15 * x22 + 8) >> 4;
    low[(size_t)f * plane_stride + ((size_t)c * g.rows + v) * g.cols + u] =
        (uint8_t)((a1 + 15 * a2 + 8) >> 4);
  }
}

// The value of the lane before inside a DPP row of 16 lanes (lane 0 of a row: its own).
__device__ __forceinline__ uint32_t dpp_row_shr1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
}

// Predictors of downsampled.cpp:41-60.
__device__ __forceinline__ int predict(int s1, int s2, int s3, int p) {
  switch (p) {
    default:
    case 0: return clamp255((3 * (s2 + s3) - 2 * s1 + 2) >> 2);
    case 1: return s2;
    case 2: return s3;
    case 3: return (s2 + s3 + 1) >> 1;
    case 4: return clamp255(s2 + s3 - s1);
  }
}

// ---------------------------------------------------------------------------
// k_lres_predict: one wavefront per 16x16 macro block.
//  1. predictor selection over ORIGINAL neighbours: 4 samples per lane, wave
//     reduction of the five squared-error sums (downsampled.cpp:182-253);
//  2. delta coding over RECONSTRUCTED neighbours (downsampled.cpp:255-315):
//     the 256-step serial chain is run as 31 anti-diagonal wavefronts, lane =
//     row of the macro block, reconstructed samples exchanged through LDS.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_lres_predict(Geom g, const uint8_t *low,
                                                     size_t plane_stride, uint8_t *lres_sym,
                                                     size_t lres_stride, LresTables lt) {
  // FOUR macro blocks per wavefront, 16 lanes each (lane & 15 = row of the block):
  // the delta chain only ever has 16 rows to work on, so one block per wave left
  // three quarters of it idle -- and a 4096x4096 frame is 4096 blocks per channel.
  __shared__ __attribute__((aligned(4))) uint8_t mb[4][16][20];   // (rows dword aligned: the fast path stores them as dwords)
  __shared__ uint8_t recp[4][17][18];   // the reconstructed block with a border row / column in front (see the chain below)
  // The companding tables in LDS: the delta chain below looks them up twice per
  // step, and out of the kernel-argument segment each lookup is a global load on
  // the critical path of 31 dependent steps.
  __shared__ int16_t s_tab[128];
  __shared__ uint8_t s_code[512];
  const int lane = threadIdx.x, b = lane >> 4, dv = lane & 15;
  const int mu = blockIdx.x * 4 + b, mv = blockIdx.y;
  const int f = blockIdx.z / g.C, c = blockIdx.z % g.C;
  for (int k = lane; k < 128; k += 64) s_tab[k] = lt.tab[k];
  for (int k = lane; k < 512; k += 64) s_code[k] = lt.code[k];
  const uint8_t *m = low + (size_t)f * plane_stride + (size_t)c * g.rows * g.cols;
  const bool live = mu < g.mcols;
  const int u0 = mu * 16, v0 = mv * 16;
  const int bw = live ? min(16, g.cols - u0) : 0, bh = min(16, g.rows - v0);

  int err[5] = {0, 0, 0, 0, 0};
  // Four full blocks in rows of sixteen-byte-aligned samples (every block of the BASELINE frames
  // but those at the right / bottom edge of odd sizes): the lane's row is ONE 16-byte load and stays
  // in registers, the row above comes from the lane before by DPP (a block is a DPP row of 16
  // lanes), and the predictors' squared errors are accumulated without a branch or an LDS read.
  const bool fast = __all(live && bw == 16 && bh == 16) && (g.cols & 15) == 0;
  if (fast) {
    const uint4 q = *reinterpret_cast<const uint4 *>(m + (size_t)(v0 + dv) * g.cols + u0);
    const uint32_t R[4] = {q.x, q.y, q.z, q.w};
    uint32_t U[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      U[k] = dpp_row_shr1(R[k]);
      *reinterpret_cast<uint32_t *>(&mb[b][dv][4 * k]) = R[k];
    }
    const bool up_ok1 = dv > 0;
#pragma unroll
    for (int du = 0; du < 16; ++du) {
      const int actual = (int)((R[du >> 2] >> (8 * (du & 3))) & 255u);
      const int up = (int)((U[du >> 2] >> (8 * (du & 3))) & 255u);
      int s1, s2, s3;
      if (du > 0) {
        const int left = (int)((R[(du - 1) >> 2] >> (8 * ((du - 1) & 3))) & 255u);
        const int ul = (int)((U[(du - 1) >> 2] >> (8 * ((du - 1) & 3))) & 255u);
        s3 = left; s2 = up_ok1 ? up : left; s1 = up_ok1 ? ul : left;
      } else {
        s1 = s2 = s3 = up_ok1 ? up : 128;
      }
      const int t = s2 + s3;
      const int pr[5] = {clamp255((3 * t - 2 * s1 + 2) >> 2), s2, s3, (t + 1) >> 1, clamp255(t - s1)};
#pragma unroll
      for (int p = 0; p < 5; ++p) {
        const int dd = actual - pr[p];
        err[p] += dd * dd;
      }
    }
    __syncthreads();
  } else {
#pragma unroll
  for (int du = 0; du < 16; ++du)
    mb[b][dv][du] = (dv < bh && du < bw) ? m[(size_t)(v0 + dv) * g.cols + u0 + du] : 0;
  __syncthreads();

  if (dv < bh) {
    for (int du = 0; du < bw; ++du) {
      int s1, s2, s3;
      if (du > 0 && dv > 0) { s1 = mb[b][dv - 1][du - 1]; s2 = mb[b][dv - 1][du]; s3 = mb[b][dv][du - 1]; }
      else if (du > 0) { s1 = s2 = s3 = mb[b][dv][du - 1]; }
      else if (dv > 0) { s1 = s2 = s3 = mb[b][dv - 1][du]; }
      else { s1 = s2 = s3 = 128; }
      const int actual = mb[b][dv][du];
#pragma unroll
      for (int p = 0; p < 5; ++p) {
        const int d = actual - predict(s1, s2, s3, p);
        err[p] += d * d;
      }
    }
  }
  }
#pragma unroll
  for (int p = 0; p < 5; ++p)
    for (int d = 8; d >= 1; d >>= 1) err[p] += __shfl_xor(err[p], d);   // over the block's 16 lanes
  int best = 0, best_err = err[0];
#pragma unroll
  for (int p = 1; p < 5; ++p)
    if (err[p] < best_err) { best = p; best_err = err[p]; }

  uint8_t *out = lres_sym + (size_t)f * lres_stride + (size_t)c * g.chan_size;
  if (live && dv == 0) out[mv * g.mcols + mu] = (uint8_t)(best - 2);  // downsampled.cpp:33-35
  // The stored byte is read back as (uint8 + 2) in int arithmetic
  // (downsampled.cpp:37-39), so selections 0 and 1 both CODE with predictor 0.
  const int pc = best <= 1 ? 0 : best;

  uint8_t *dst = out + g.mrows * g.mcols + (size_t)v0 * g.cols + (size_t)bh * u0;
  // The delta chain, branch free: the three reconstructed neighbours are read from a copy of the
  // block with a border (index + 1: the reads of row / column -1 land on it, their values are
  // not used), the cases of downsampled.cpp:263-281 are four selects (f = the neighbour that
  // stands for all three at an edge), all five predictors are computed and the block's is
  // selected -- the four blocks of a wavefront code with different predictors, and a switch ran
  // every case taken by any of them.  (~100 -> ~45 instructions per anti-diagonal step.)
  const bool row_live = dv < bh;
  const bool up_ok = dv > 0;
  uint8_t *rrow = &recp[b][dv + 1][1];          // rrow[du] = reconstructed sample (dv, du)
  const uint8_t *urow = &recp[b][dv][1];        // the row above
  for (int d = 0; d < 31; ++d) {
    const int du = d - dv;
    const bool active = row_live && du >= 0 && du < bw;
    const int duc = active ? du : 0;
    const int left = rrow[duc - 1], up = urow[duc], ul = urow[duc - 1];
    const bool left_ok = duc > 0;
    const int f = up_ok ? up : (left_ok ? left : 128);
    const int s3 = left_ok ? left : f, s2 = f, s1 = (up_ok && left_ok) ? ul : f;
    const int t = s2 + s3;
    const int p0 = clamp255((3 * t - 2 * s1 + 2) >> 2), p3 = (t + 1) >> 1, p4 = clamp255(t - s1);
    const int predicted = pc == 2 ? s3 : pc == 3 ? p3 : pc == 4 ? p4 : p0;   // (pc is 0, 2, 3 or 4: selections 0 and 1 both code with 0)
    const int delta = (int)mb[b][dv][duc] - predicted;
    const uint8_t code = s_code[delta + 255];
    const int sc = (int8_t)code;
    const int mag = s_tab[sc < 0 ? -sc : sc];
    const int un = sc < 0 ? -mag : mag;
    if (active) {
      rrow[du] = (uint8_t)clamp255(predicted + un);
      dst[dv * bw + du] = code;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Tile helpers.
// ---------------------------------------------------------------------------

// Recursive rounded midpoints at positions 4,2,6,1,3,5,7 (downsampled.cpp:131-147).
__device__ __forceinline__ void interp9(int a[9]) {
  a[4] = (a[0] + a[8] + 1) >> 1;
  a[2] = (a[0] + a[4] + 1) >> 1;
  a[6] = (a[4] + a[8] + 1) >> 1;
  a[1] = (a[0] + a[2] + 1) >> 1;
  a[3] = (a[2] + a[4] + 1) >> 1;
  a[5] = (a[4] + a[6] + 1) >> 1;
  a[7] = (a[6] + a[8] + 1) >> 1;
}

// 8-point sequency-ordered WHT butterfly (hadamard.cpp:18-44); in-place on
// eight ints.  Forward results are taken mod 2^16 by the caller (the reference
// computes in wrapping int16; add/sub commute with the wrap).
__device__ __forceinline__ void wht8(int &x0, int &x1, int &x2, int &x3, int &x4, int &x5,
                                     int &x6, int &x7) {
  const int a0 = x0 + x4, a1 = x1 + x5, a2 = x2 + x6, a3 = x3 + x7;
  const int a4 = x0 - x4, a5 = x1 - x5, a6 = x2 - x6, a7 = x3 - x7;
  const int b0 = a0 + a2, b1 = a1 + a3, b2 = a0 - a2, b3 = a1 - a3;
  const int b4 = a4 + a6, b5 = a5 + a7, b6 = a4 - a6, b7 = a5 - a7;
  x0 = b0 + b1; x1 = b4 + b5; x2 = b6 + b7; x3 = b2 + b3;
  x4 = b2 - b3; x5 = b6 - b7; x6 = b4 - b5; x7 = b0 - b1;
}

// ---------------------------------------------------------------------------
// k_tile_fwd: one lane per 8x8 tile, a wavefront covers 64 horizontally
// adjacent tiles of one block row, so for every coefficient the 64 lanes write
// 64 consecutive symbol bytes (encoder.cpp:320-323 layout).
//
// This is the generic kernel (ragged edges, 1-3 channels, pixel stride != 4);
// full RGBA8 tiles take k_pix_fwd below.  Channels are processed one after
// the other and the tile's pixels are re-read for each instead of being held in
// registers, which keeps the register count down.
// ---------------------------------------------------------------------------
enum { kChanRaw = 0, kChanY = 1, kChanCb = 2, kChanCr = 3 };

// One channel value of a packed RGBA pixel, with the colour lift of
// ycbcr.cpp:32-37 applied on the fly.
template <int MODE>
__device__ __forceinline__ int channel_value(uint32_t px, int sh) {
  if (MODE == kChanRaw) return (int)((px >> sh) & 255u);
  const int c0 = px & 255, c1 = (px >> 8) & 255, c2 = (px >> 16) & 255;
  if (MODE == kChanY) return (c0 + 2 * c1 + c2 + 2) >> 2;
  if (MODE == kChanCb) return (c2 - c1 + 256) >> 1;
  return (c0 - c1 + 256) >> 1;
}

// Residual of one full tile of one channel: pixel - bilinear low-res block
// (downsampled.cpp:116-169, encoder.cpp:304-309).  `row0` points at the tile's
// first pixel (RGBA8, 16-byte aligned), `pitch` is the image row pitch in bytes.
// The tile is re-read for every channel (L2 / Infinity Cache hits after the
// first pass); staging it in lane-private LDS instead measured 25 % slower.
constexpr int kTileThreads = 256;

template <int MODE>
__device__ __forceinline__ void residual_full_tile(const uint8_t *row0, size_t pitch, int sh,
                                                   const int left[9], const int right[9],
                                                   int b[64]) {
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    const uint4 *rp = reinterpret_cast<const uint4 *>(row0 + (size_t)y * pitch);
    const uint4 q0 = rp[0], q1 = rp[1];
    int a[9];
    a[0] = left[y]; a[8] = right[y];
    interp9(a);
    b[y * 8 + 0] = channel_value<MODE>(q0.x, sh) - a[0];
    b[y * 8 + 1] = channel_value<MODE>(q0.y, sh) - a[1];
    b[y * 8 + 2] = channel_value<MODE>(q0.z, sh) - a[2];
    b[y * 8 + 3] = channel_value<MODE>(q0.w, sh) - a[3];
    b[y * 8 + 4] = channel_value<MODE>(q1.x, sh) - a[4];
    b[y * 8 + 5] = channel_value<MODE>(q1.y, sh) - a[5];
    b[y * 8 + 6] = channel_value<MODE>(q1.z, sh) - a[6];
    b[y * 8 + 7] = channel_value<MODE>(q1.w, sh) - a[7];
  }
}

// FAST: every tile is full and pixels are packed RGBA8 (the BASELINE configs);
// otherwise the generic path handles ragged edges, other channel counts and
// pixel strides.
template <bool FAST, int COLS>
__global__ __launch_bounds__(kTileThreads) void k_tile_fwd(Geom g, const uint8_t *frames,
                                                  const uint8_t *low, size_t plane_stride,
                                                  uint8_t *fres_sym, size_t fres_stride,
                                                  const uint8_t *__restrict__ fmap_lut,
                                                  ShiftTables st, int v0) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.y + v0, f = blockIdx.z;
  if (u >= g.cols) return;
  const uint8_t *img = frames + (long long)f * g.frame_bytes;
  const int bw = min(8, g.W - 8 * u), bh = min(8, g.H - 8 * v);
  const int u2 = min(u + 1, g.cols - 1), v2 = min(v + 1, g.rows - 1);
  uint8_t *dst_row = fres_sym + (size_t)f * fres_stride + (size_t)v * g.row_block + u;
  const uint8_t *row0 = img + ((long long)(8 * v) * g.W + 8 * u) * 4;
  const size_t pitch = (size_t)g.W * 4;

#pragma unroll 1
  for (int c = 0; c < g.C; ++c) {
    const uint8_t *m = low + (size_t)f * plane_stride + (size_t)c * g.rows * g.cols;
    // Bilinear low-res block from the four corners (downsampled.cpp:116-169).
    int left[9], right[9];
    left[0] = m[(size_t)v * g.cols + u];   left[8] = m[(size_t)v2 * g.cols + u];
    right[0] = m[(size_t)v * g.cols + u2]; right[8] = m[(size_t)v2 * g.cols + u2];
    interp9(left);
    interp9(right);

    int b[64];
    if (FAST) {
      const int mode = (g.ycbcr && c < 3) ? (c + 1) : kChanRaw;  // wave-uniform
      if (mode == kChanRaw) residual_full_tile<kChanRaw>(row0, pitch, 8 * c, left, right, b);
      else if (mode == kChanY) residual_full_tile<kChanY>(row0, pitch, 0, left, right, b);
      else if (mode == kChanCb) residual_full_tile<kChanCb>(row0, pitch, 0, left, right, b);
      else residual_full_tile<kChanCr>(row0, pitch, 0, left, right, b);
    } else {
      // Partial tiles replicate the last valid pixel of the row, rows below the
      // image repeat the bottom-right valid pixel (encoder.cpp:26-52).
#pragma unroll
      for (int y = 0; y < 8; ++y) {
        int a[9];
        a[0] = left[y]; a[8] = right[y];
        interp9(a);
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          const int yy = y < bh ? y : bh - 1;
          const int xx = y < bh ? min(x, bw - 1) : bw - 1;
          int ch[4];
          load_pixel(img, g, 8 * u + xx, 8 * v + yy, ch);
          b[y * 8 + x] = ch[c] - a[x];
        }
      }
    }
    // Forward 2-D WHT: rows, then columns (hadamard.cpp:78-88).
#pragma unroll
    for (int y = 0; y < 8; ++y)
      wht8(b[y * 8 + 0], b[y * 8 + 1], b[y * 8 + 2], b[y * 8 + 3], b[y * 8 + 4], b[y * 8 + 5],
           b[y * 8 + 6], b[y * 8 + 7]);
#pragma unroll
    for (int x = 0; x < 8; ++x)
      wht8(b[x], b[8 + x], b[16 + x], b[24 + x], b[32 + x], b[40 + x], b[48 + x], b[56 + x]);

    const bool chroma = g.ycbcr && (c == 1 || c == 2);  // encoder.cpp:284
    const uint8_t *shift = st.s[chroma ? 1 : 0];
    const int cols = COLS ? COLS : g.cols;  // compile-time stride -> no 64 live store addresses
    uint8_t *dst = dst_row + (size_t)c * 64 * cols;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      const int pos = kScan[i];
      const int s = shift[pos];
      const int x = (int)(int16_t)b[pos];  // the reference's int16 wrap
      // Sign-magnitude rounding shift (quantize.cpp:135-148).
      const int r = s ? (1 << (s - 1)) : 0;
      const int mag = x < 0 ? ((-x + r) >> s) : ((x + r) >> s);
      // Companding (mapper.cpp:159-182): the full-res table is the identity up to
      // 50; larger magnitudes go through the LUT of the restated search.
      uint32_t code = (uint32_t)mag;
      if (mag > 50) code = fmap_lut[mag];
      dst[(size_t)i * cols] = (x < 0) ? (uint8_t)(0u - code) : (uint8_t)code;
    }
  }
}

// ---------------------------------------------------------------------------
// Packed int16 helpers of the pixel stage for full RGBA8 tiles (k_pix_fwd below):
// two channels per register (v_pk_*_i16).  With 8-bit input the forward path never
// leaves int16 (|coefficient| <= 255 * 64), so the packed arithmetic is exact.
// ---------------------------------------------------------------------------
typedef short pk16 __attribute__((ext_vector_type(2)));
typedef unsigned short upk16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wht8_pk(pk16 &x0, pk16 &x1, pk16 &x2, pk16 &x3, pk16 &x4, pk16 &x5,
                                        pk16 &x6, pk16 &x7) {
  const pk16 a0 = x0 + x4, a1 = x1 + x5, a2 = x2 + x6, a3 = x3 + x7;
  const pk16 a4 = x0 - x4, a5 = x1 - x5, a6 = x2 - x6, a7 = x3 - x7;
  const pk16 b0 = a0 + a2, b1 = a1 + a3, b2 = a0 - a2, b3 = a1 - a3;
  const pk16 b4 = a4 + a6, b5 = a5 + a7, b6 = a4 - a6, b7 = a5 - a7;
  x0 = b0 + b1; x1 = b4 + b5; x2 = b6 + b7; x3 = b2 + b3;
  x4 = b2 - b3; x5 = b6 - b7; x6 = b4 - b5; x7 = b0 - b1;
}

// ---------------------------------------------------------------------------
// k_pix_fwd: the pixel stage of the BASELINE shapes (full tiles, packed RGBA8,
// cols a multiple of 16) in its round-2 form.  Lane = tile, a wavefront = 64
// adjacent tiles of one block row, like k_tile_fwd, but
//   * the tile's 64 pixels are loaded ONCE (sixteen 16-byte loads) and stay in
//     registers for both channel pairs -- the kernel runs at two waves per SIMD,
//     which costs this ALU-dense code a few per cent and saves a second pass
//     over the pixels;
//   * the pairs are chosen so that the colour lift is packed arithmetic:
//     (R,B) = px & 0x00ff00ff gives (Cr,Cb) = ((R,B) + 256 - (G,G)) >> 1 in three
//     packed ops, Y = v_dot4_u32_u8(px, {1,2,1,0}) + 2 >> 2; so pair 0 = (Cr, Cb)
//     (both use the chroma shifts) and pair 1 = (Y, A) (both luma) -- one shift
//     amount per coefficient for both halves (ycbcr.cpp:32-37, encoder.cpp:284);
//   * the bilinear low-res block (downsampled.cpp:116-169) is built four rows per
//     register with v_lerp_u8 ((a + b + 1) >> 1 per byte), 29 instructions per
//     channel instead of 8 x 7 x 3;
//   * the sign-magnitude rounding shift (quantize.cpp:135-148) is branch free:
//     -((-x + r) >> s) == (x + r - 1) >> s (arithmetic) for x < 0, s >= 1, so
//     q = (x + r + (sign & [s > 0])) >> s for either sign; the companding map
//     (mapper.cpp:159-182) is the identity up to 50, so the byte is the low byte
//     of q unless some |q| > 50 -- tracked with one packed max per coefficient
//     and redone with the LUT for the (rare) tile that needs it;
//   * the symbol bytes leave through buffer_store_byte with the lane offset in a
//     VGPR and the coefficient row's offset in an SGPR: no 64-bit address adds.
// Tried and measured, not kept: having this kernel also leave the non-zero masks
// of the symbol stream (one ballot per channel and coefficient, lane i keeping the
// word of coefficient i through v_writelane) for the entropy kernels to tokenise
// from: +0.05 ms here per 8 frames, and the mask-driven histogram kernel was no
// faster than k_tok_hist (0.305 vs 0.295 ms) -- the per-token walk, not the search
// for the zeros, is what those kernels spend their time on.
// ---------------------------------------------------------------------------
template <class F, int... I>
__device__ __forceinline__ void for_seq_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void for_seq(F &&f) { for_seq_impl(f, std::make_integer_sequence<int, N>{}); }

// lerp tree of downsampled.cpp:116-169 on four packed bytes.
__device__ __forceinline__ void interp9_u8x4e(uint32_t a[9]) {
  const uint32_t rnd = 0x01010101u;
  a[4] = __builtin_amdgcn_lerp(a[0], a[8], rnd);
  a[2] = __builtin_amdgcn_lerp(a[0], a[4], rnd);
  a[6] = __builtin_amdgcn_lerp(a[4], a[8], rnd);
  a[1] = __builtin_amdgcn_lerp(a[0], a[2], rnd);
  a[3] = __builtin_amdgcn_lerp(a[2], a[4], rnd);
  a[5] = __builtin_amdgcn_lerp(a[4], a[6], rnd);
  a[7] = __builtin_amdgcn_lerp(a[6], a[8], rnd);
}

// Low-res block of one channel: LQ[q][x] = bytes of rows 4q..4q+3 at column x.
// lr0 / lr8: (left, right) low-res samples of this and the next block row in bytes 0, 1.
__device__ __forceinline__ void lowres_quads_e(uint32_t lr0, uint32_t lr8, uint32_t LQ[2][8]) {
  uint32_t lr[9];
  lr[0] = lr0; lr[8] = lr8;
  interp9_u8x4e(lr);   // bytes 0,1 = left,right of rows 0..7
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint32_t p01 = __builtin_amdgcn_perm(lr[4 * q + 1], lr[4 * q], 0x05010400u);      // L0 L1 R0 R1
    const uint32_t p23 = __builtin_amdgcn_perm(lr[4 * q + 3], lr[4 * q + 2], 0x05010400u);  // L2 L3 R2 R3
    uint32_t a[9];
    a[0] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
    a[8] = __builtin_amdgcn_perm(p23, p01, 0x07060302u);
    interp9_u8x4e(a);
#pragma unroll
    for (int x = 0; x < 8; ++x) LQ[q][x] = a[x];
  }
}

// The two channel values of pair PAIR of one pixel as packed int16.
//   YCBCR: pair 0 = (Cr, Cb) = channels (2, 1), pair 1 = (Y, A) = channels (0, 3)
//   else : pair 0 = channels (0, 2),            pair 1 = channels (1, 3)
template <bool YCBCR, int PAIR>
__device__ __forceinline__ pk16 pix_pair(uint32_t px) {
  if (!YCBCR) {
    const uint32_t v = PAIR == 0 ? (px & 0x00ff00ffu) : ((px >> 8) & 0x00ff00ffu);
    return __builtin_bit_cast(pk16, v);
  }
  if (PAIR == 0) {
    const uint32_t rb = (px & 0x00ff00ffu) | 0x01000100u;                 // (R + 256, B + 256)
    const uint32_t gg = __builtin_amdgcn_perm(px, px, 0x0c010c01u);       // (G, G)
    const upk16 d = __builtin_bit_cast(upk16, (pk16)(__builtin_bit_cast(pk16, rb) - __builtin_bit_cast(pk16, gg)));
    const upk16 one = {1, 1};
    return __builtin_bit_cast(pk16, (upk16)(d >> one));                   // (Cr, Cb)
  }
  const uint32_t y = __builtin_amdgcn_udot4(px, 0x00010201u, 2u, false) >> 2;   // (R + 2G + B + 2) >> 2
  return __builtin_bit_cast(pk16, __builtin_amdgcn_perm(px, y, 0x0c070c00u));    // (Y, A)
}

constexpr int kPixThreads = 256;
constexpr int kPixLut = 8192;

// The quantiser's per-coefficient constants as packed pairs in SCAN order, [0] luma /
// [1] chroma (quantize.cpp:127-151): rounding term r = s ? 1 << (s - 1) : 0, sign
// factor k = [s > 0], shift s -- each in both halves of a word.  Kernel arguments:
// scalar loads, nothing derived per coefficient in the kernel.
struct PixQuant {
  uint32_t rr[2][64], kk[2][64], ss[2][64];
};
static PixQuant make_pix_quant(const ShiftTables &st) {
  PixQuant pq;
  for (int t = 0; t < 2; ++t)
    for (int i = 0; i < 64; ++i) {
      const uint32_t sft = st.s[t][kScanHost[i]], r = sft ? 1u << (sft - 1) : 0u, k = sft ? 1u : 0u;
      pq.rr[t][i] = r | (r << 16);
      pq.kk[t][i] = k | (k << 16);
      pq.ss[t][i] = sft | (sft << 16);
    }
  return pq;
}

template <bool YCBCR, int COLS, bool FULL>
__global__ __launch_bounds__(kPixThreads, 2) void k_pix_fwd(Geom g, const uint8_t *frames,
                                                            const uint8_t *low, size_t plane_stride,
                                                            uint8_t *fres_sym, size_t fres_stride,
                                                            const uint8_t *__restrict__ fmap_lut,
                                                            PixQuant pq, int v0) {
  // Companding LUT for magnitudes below kPixLut (every larger one maps to 127:
  // the full-res table tops out at 8039, mapper.cpp:54-71,159-182).
  __shared__ __attribute__((aligned(16))) uint8_t s_lut[kPixLut];
  for (int k = threadIdx.x; k < kPixLut / 16; k += kPixThreads)
    reinterpret_cast<uint4 *>(s_lut)[k] = reinterpret_cast<const uint4 *>(fmap_lut)[k];
  __syncthreads();
  const int cols = COLS ? COLS : g.cols;
  const int u = blockIdx.x * kPixThreads + threadIdx.x;
  const int v = blockIdx.y + v0, f = blockIdx.z;
  if ((int)(blockIdx.x * kPixThreads + (threadIdx.x & ~63)) >= cols) return;   // the whole wave is beyond the row
  // FULL: cols is a multiple of 64, every lane of a live wave owns a tile.
  const bool valid = FULL || u < cols;
  const int uc = valid ? u : cols - 1;
  const uint8_t *img = frames + (long long)f * g.frame_bytes;
  const int u2 = min(uc + 1, cols - 1), v2 = min(v + 1, g.rows - 1);
  // Wave-uniform base + 32-bit lane offset: the stores take the scalar-base form and
  // the per-coefficient stride is scalar arithmetic, not 64-bit adds per lane.
  // The symbol stores go through a buffer descriptor of the frame's symbol plane:
  // buffer_store_byte takes the lane offset in a VGPR and the (wave-uniform) offset of
  // the coefficient row in an SGPR -- no address arithmetic on the vector unit.
  const __amdgpu_buffer_rsrc_t sym_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      fres_sym + (size_t)f * fres_stride, 0, (int)g.fres_size, 0x00020000);
  const uint32_t row_off = (uint32_t)v * (uint32_t)g.row_block;
  const uint32_t lane_off = (uint32_t)uc;
  const uint8_t *row0 = img + ((long long)(8 * v) * g.W + 8 * uc) * 4;
  const size_t pitch = (size_t)g.W * 4;

  // The tile: 8 rows x 8 pixels, one pass over HBM.
  uint32_t px[64];
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    const uint4 *rp = reinterpret_cast<const uint4 *>(row0 + (size_t)y * pitch);
    const uint4 q0 = rp[0], q1 = rp[1];
    px[y * 8 + 0] = q0.x; px[y * 8 + 1] = q0.y; px[y * 8 + 2] = q0.z; px[y * 8 + 3] = q0.w;
    px[y * 8 + 4] = q1.x; px[y * 8 + 5] = q1.y; px[y * 8 + 6] = q1.z; px[y * 8 + 7] = q1.w;
  }
  // Low-res corners of the four channels: (left, right) of block rows v and v + 1.
  uint32_t lr0[4], lr8[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint8_t *m = low + (size_t)f * plane_stride + (size_t)c * g.rows * cols;
    lr0[c] = (uint32_t)m[(size_t)v * cols + uc] | ((uint32_t)m[(size_t)v * cols + u2] << 8);
    lr8[c] = (uint32_t)m[(size_t)v2 * cols + uc] | ((uint32_t)m[(size_t)v2 * cols + u2] << 8);
  }

#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    // Channels in the low / high half of this pair, and their shift table.
    const int cA = YCBCR ? (pr == 0 ? 2 : 0) : (pr == 0 ? 0 : 1);
    const int cB = YCBCR ? (pr == 0 ? 1 : 3) : (pr == 0 ? 2 : 3);
    pk16 b[64];
    {
      uint32_t LA[2][8], LB[2][8];
      lowres_quads_e(lr0[cA], lr8[cA], LA);
      lowres_quads_e(lr0[cB], lr8[cB], LB);
#pragma unroll
      for (int y = 0; y < 8; ++y)
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          // (low A, low B) of this pixel, zero-extended to the two halves.
          const uint32_t sel = 0x0c000c00u | (uint32_t)(y & 3) | ((uint32_t)(4 + (y & 3)) << 16);
          const pk16 lo = __builtin_bit_cast(pk16, __builtin_amdgcn_perm(LB[y >> 2][x], LA[y >> 2][x], sel));
          const pk16 pv = pr == 0 ? pix_pair<YCBCR, 0>(px[y * 8 + x]) : pix_pair<YCBCR, 1>(px[y * 8 + x]);
          b[y * 8 + x] = pv - lo;
        }
    }
    // Forward 2-D WHT: rows, then columns (hadamard.cpp:78-88).
#pragma unroll
    for (int y = 0; y < 8; ++y)
      wht8_pk(b[y * 8 + 0], b[y * 8 + 1], b[y * 8 + 2], b[y * 8 + 3], b[y * 8 + 4], b[y * 8 + 5],
              b[y * 8 + 6], b[y * 8 + 7]);
#pragma unroll
    for (int x = 0; x < 8; ++x)
      wht8_pk(b[x], b[8 + x], b[16 + x], b[24 + x], b[32 + x], b[40 + x], b[48 + x], b[56 + x]);

    const uint32_t offA = row_off + (uint32_t)(cA * 64 * cols), offB = row_off + (uint32_t)(cB * 64 * cols);
    // Quantise (quantize.cpp:127-151), compand (mapper.cpp:159-182) and store, in
    // groups of coefficients in scan order.  sign * ((|x| + r) >> s) is branch free:
    // (x + r + sign * k) >> s with sign = x >> 15 and k = [s > 0]; r, k and s come
    // as packed pairs from the kernel arguments (pq, scan order).  Companding is the
    // identity while |q| <= 50; the group keeps the largest q + 50 (as unsigned: <=
    // 100 exactly then) and ONE wave-uniform branch per group sends the group
    // through the LUT -- the first sixteen coefficients (DC and first order, the
    // ones high-contrast tiles push beyond 50) in groups of four, the rest in
    // sixteens.  (A test per coefficient was 128 compare + branch pairs per tile,
    // each behind hazard no-ops, and 128 basic blocks the scheduler could not
    // interleave across.)
    const int qt = (YCBCR && pr == 0) ? 1 : 0;
    auto group = [&](auto i0c, auto nc) {
      constexpr int I0 = decltype(i0c)::value, N = decltype(nc)::value;
      pk16 q[N];
      upk16 top = {0, 0};
      for_seq<N>([&](auto kc) {
        constexpr int k = decltype(kc)::value, i = I0 + k, pos = kScan[i];
        const pk16 x = b[pos];
        const pk16 fifteen = {15, 15};
        const pk16 sign = x >> fifteen;                       // 0 or -1 per half
        const pk16 rr = __builtin_bit_cast(pk16, pq.rr[qt][i]), kk = __builtin_bit_cast(pk16, pq.kk[qt][i]);
        const pk16 ss = __builtin_bit_cast(pk16, pq.ss[qt][i]);
        q[k] = (sign * kk + x + rr) >> ss;
        const upk16 fifty = {50, 50};
        top = __builtin_elementwise_max(top, (upk16)(__builtin_bit_cast(upk16, q[k]) + fifty));
      });
      const upk16 hundred = {100, 100};
      const uint32_t over = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(top, hundred));
      if (__builtin_expect(__any(over != 0u), 0)) {
        for_seq<N>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          const pk16 fifteen = {15, 15};
          const pk16 sign = q[k] >> fifteen;
          const pk16 mag = (q[k] ^ sign) - sign;
          const uint32_t ma = min((uint32_t)(uint16_t)mag.x, (uint32_t)(kPixLut - 1));
          const uint32_t mb = min((uint32_t)(uint16_t)mag.y, (uint32_t)(kPixLut - 1));
          pk16 code;                                          // the LUT is the identity below 51
          code.x = (short)s_lut[ma];
          code.y = (short)s_lut[mb];
          q[k] = (code ^ sign) - sign;
        });
      }
      if (valid) {
        for_seq<N>([&](auto kc) {
          constexpr int k = decltype(kc)::value, i = I0 + k;
          __builtin_amdgcn_raw_buffer_store_b8((uint8_t)q[k].x, sym_rsrc, lane_off, offA + (uint32_t)(i * cols), 0);
          __builtin_amdgcn_raw_buffer_store_b8((uint8_t)q[k].y, sym_rsrc, lane_off, offB + (uint32_t)(i * cols), 0);
        });
      }
    };
    using std::integral_constant;
    group(integral_constant<int, 0>{}, integral_constant<int, 4>{});
    group(integral_constant<int, 4>{}, integral_constant<int, 4>{});
    group(integral_constant<int, 8>{}, integral_constant<int, 4>{});
    group(integral_constant<int, 12>{}, integral_constant<int, 4>{});
    group(integral_constant<int, 16>{}, integral_constant<int, 16>{});
    group(integral_constant<int, 32>{}, integral_constant<int, 16>{});
    group(integral_constant<int, 48>{}, integral_constant<int, 16>{});
  }
}

// ---------------------------------------------------------------------------
// k_front: box averages, low-res plane AND pixel stage in one pass over the pixels (batches of
// full RGBA8 frames, rows of at most 512 tiles) -- k_lowres_avg + k_lowres_blend + k_pix_fwd
// read every pixel twice; here a workgroup walks DOWN a chunk of block rows, one lane per tile of
// the row's width, and a pixel is read from HBM once:
//   step t   the tile row t arrives in registers (requested during the step before);
//            its 5x5 / 3x5 / 5x3 / 3x3 corner sums give, with the row above's, the 8x8 windows at
//            offset -3 (downsampled.cpp:76-96): box averages of row t, exchanged through LDS;
//            low-res samples of row t (downsampled.cpp:98-113) from the averages of rows t - 1, t;
//            THEN tile row t - 1 is transformed (it needed the low-res rows t - 1 and t), its pixels
//            read from the wavefront's 16 KiB parking slot in LDS; once the second channel pair has
//            read them, row t is parked in the same slot and row t + 1 is requested: the loads fly
//            under the second pair's WHT / quantise / stores.
// A chunk starts two tile rows above its first (their sums only) -- 3 % more pixel reads at 64 rows
// per chunk.  Same arithmetic as k_pix_fwd (pix_pair, lowres_quads_e, packed WHT, group-tested
// companding), same symbols; avg / low planes are written for the LRES branch and the tests.
// ---------------------------------------------------------------------------
constexpr int kFrontExch = 5;   // dwords per tile in the exchange area: right-column sums of both pairs (top, bottom), averages

// Corner sums of one tile for one channel pair: TL = rows 0..4 x columns 0..4, TR = rows 0..4 x
// columns 5..7, BL / BR = rows 5..7 (the window of tile (u, v) is x in [8u-3, 8u+4], y alike).
template <bool YCBCR, int PAIR>
__device__ __forceinline__ void front_sums(const uint32_t (&px)[64], pk16 &tl, pk16 &tr, pk16 &bl, pk16 &br) {
  const pk16 z = {0, 0};
  tl = tr = bl = br = z;
#pragma unroll
  for (int y = 0; y < 8; ++y)
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const pk16 v = pix_pair<YCBCR, PAIR>(px[y * 8 + x]);
      if (y < 5) { if (x < 5) tl += v; else tr += v; }
      else { if (x < 5) bl += v; else br += v; }
    }
}

// Channel bytes (c0..c3 in a dword) <-> the two packed pairs of pix_pair.
template <bool YCBCR>
__device__ __forceinline__ uint32_t front_pack_channels(uint32_t p0x, uint32_t p0y, uint32_t p1x, uint32_t p1y) {
  // YCBCR: pair 0 = (Cr, Cb) = channels (2, 1), pair 1 = (Y, A) = channels (0, 3); else (0, 2), (1, 3)
  return YCBCR ? (p1x | (p0y << 8) | (p0x << 16) | (p1y << 24)) : (p0x | (p1x << 8) | (p0y << 16) | (p1y << 24));
}

// low = blend of the averages at (t-1, t) x (u-1, u) (downsampled.cpp:98-113), four channel bytes at once.
__device__ __forceinline__ uint32_t front_blend(uint32_t a11, uint32_t a12, uint32_t a21, uint32_t a22) {
  uint32_t out = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t x11 = (a11 >> (8 * c)) & 255u, x12 = (a12 >> (8 * c)) & 255u;
    const uint32_t x21 = (a21 >> (8 * c)) & 255u, x22 = (a22 >> (8 * c)) & 255u;
    const uint32_t b1 = (x11 + 15u * x12 + 8u) >> 4, b2 = (x21 + 15u * x22 + 8u) >> 4;
    out |= ((b1 + 15u * b2 + 8u) >> 4) << (8 * c);
  }
  return out;
}

// (One argument block: the quantiser's tables are read through the kernel-argument segment at
// offsetof(FrontArgs, pq) -- see the walk below.)
struct FrontArgs {
  Geom g;
  const uint8_t *frames;
  uint8_t *avg, *low;
  size_t plane_stride;
  uint8_t *fres_sym;
  size_t fres_stride;
  const uint8_t *fmap_lut;
  int chunk_rows;
  PixQuant pq;
};
typedef const __attribute__((address_space(4))) uint32_t *KargWords;

template <bool YCBCR, int COLS>
__global__ __launch_bounds__(512) void k_front(FrontArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const Geom &g = a.g;
  const uint8_t *frames = a.frames, *fmap_lut = a.fmap_lut;
  uint8_t *avg = a.avg, *low = a.low, *fres_sym = a.fres_sym;
  const size_t plane_stride = a.plane_stride, fres_stride = a.fres_stride;
  const int chunk_rows = a.chunk_rows;
  const int cols = COLS ? COLS : g.cols;
  const int nt = (int)blockDim.x;                       // 64 x wavefronts per row
  uint4 *park = reinterpret_cast<uint4 *>(smem);        // [wave][16][64]: the tile row in waiting
  uint8_t *s_lut = smem + (size_t)nt * 256;             // (nt / 64 waves x 16 KiB)
  uint32_t *s_ex = reinterpret_cast<uint32_t *>(s_lut + kPixLut);   // [kFrontExch][nt]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int f = blockIdx.y;
  const int v0 = (int)blockIdx.x * chunk_rows, v1 = min(v0 + chunk_rows, g.rows);
  for (int k = tid; k < kPixLut / 16; k += nt)
    reinterpret_cast<uint4 *>(s_lut)[k] = reinterpret_cast<const uint4 *>(fmap_lut)[k];
  const int u = tid;
  const bool valid = u < cols;
  const int uc = valid ? u : cols - 1;
  const uint8_t *img = frames + (long long)f * g.frame_bytes;
  const size_t pitch = (size_t)g.W * 4;
  uint4 *slot = park + (size_t)wv * 16 * 64 + lane;     // + k * 64: piece k (pixel row k >> 1, half k & 1)
  const __amdgpu_buffer_rsrc_t sym_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      fres_sym + (size_t)f * fres_stride, 0, (int)g.fres_size, 0x00020000);
  uint8_t *avg_f = avg + (size_t)f * plane_stride, *low_f = low + (size_t)f * plane_stride;
  const size_t chan = (size_t)g.rows * cols;

  uint32_t px[64];
  auto load_row = [&](int t) {
    const uint8_t *row0 = img + ((long long)(8 * t) * g.W + 8 * uc) * 4;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      const uint4 *rp = reinterpret_cast<const uint4 *>(row0 + (size_t)y * pitch);
      const uint4 q0 = rp[0], q1 = rp[1];
      px[y * 8 + 0] = q0.x; px[y * 8 + 1] = q0.y; px[y * 8 + 2] = q0.z; px[y * 8 + 3] = q0.w;
      px[y * 8 + 4] = q1.x; px[y * 8 + 5] = q1.y; px[y * 8 + 6] = q1.z; px[y * 8 + 7] = q1.w;
    }
  };
  auto park_row = [&]() {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      uint4 q;
      q.x = px[k * 4 + 0]; q.y = px[k * 4 + 1]; q.z = px[k * 4 + 2]; q.w = px[k * 4 + 3];
      slot[k * 64] = q;
    }
  };

  const int t_begin = max(v0 - 2, 0), t_end = v1;   // rows whose sums this chunk needs (t_end == rows: nothing to load)
  load_row(t_begin);
  const pk16 zero2 = {0, 0};
  pk16 bl_prev[2] = {zero2, zero2}, brl_prev[2] = {zero2, zero2};   // BL(u, t-1), BR(u-1, t-1)
  uint32_t avg_prev[3] = {0, 0, 0};    // averages of row t - 1 at u - 1, u, u + 1 (clipped)
  uint32_t low_prev[2] = {0, 0};       // low-res row t - 1 at u, u2
  __syncthreads();                     // (the LUT)

  for (int t = t_begin; t <= t_end; ++t) {
    // The quantiser's 384 words are scalar loads from the kernel arguments AT THEIR USES, as in
    // k_pix_fwd: through a pointer the compiler cannot see through, or it hoists all of them out
    // of this loop (370 scalar registers spilled).
    KargWords qw = (KargWords)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() +
                               offsetof(FrontArgs, pq));
    asm volatile("" : "+s"(qw));
    const bool have = t < g.rows;
    uint32_t low_cur[2] = {low_prev[0], low_prev[1]};   // (t == rows: the row below the last one is the last one)
    if (have) {
      // ---- corner sums of tile row t, windows, box averages ----
      pk16 tl[2], tr[2], bl[2], br[2];
      front_sums<YCBCR, 0>(px, tl[0], tr[0], bl[0], br[0]);
      front_sums<YCBCR, 1>(px, tl[1], tr[1], bl[1], br[1]);
      s_ex[0 * nt + tid] = __builtin_bit_cast(uint32_t, tr[0]);
      s_ex[1 * nt + tid] = __builtin_bit_cast(uint32_t, tr[1]);
      s_ex[2 * nt + tid] = __builtin_bit_cast(uint32_t, br[0]);
      s_ex[3 * nt + tid] = __builtin_bit_cast(uint32_t, br[1]);
      __syncthreads();
      pk16 trl[2] = {zero2, zero2}, brl[2] = {zero2, zero2};   // of the tile to the left (none at u = 0)
      if (u > 0) {
        trl[0] = __builtin_bit_cast(pk16, s_ex[0 * nt + tid - 1]); trl[1] = __builtin_bit_cast(pk16, s_ex[1 * nt + tid - 1]);
        brl[0] = __builtin_bit_cast(pk16, s_ex[2 * nt + tid - 1]); brl[1] = __builtin_bit_cast(pk16, s_ex[3 * nt + tid - 1]);
      }
      const upk16 w0 = __builtin_bit_cast(upk16, (pk16)(tl[0] + trl[0] + bl_prev[0] + brl_prev[0]));
      const upk16 w1 = __builtin_bit_cast(upk16, (pk16)(tl[1] + trl[1] + bl_prev[1] + brl_prev[1]));
      // (sum + cnt / 2) / cnt with cnt = (u ? 8 : 5) * (t ? 8 : 5): a 24-bit multiply by 2^22 / cnt
      // rounded up, exact for sums up to 255 * 64.
      const uint32_t cnt = (u ? 8u : 5u) * (t ? 8u : 5u);
      const uint32_t half = cnt >> 1, mul = cnt == 64u ? 65536u : cnt == 40u ? 104858u : 167773u;
      auto mean = [&](uint32_t sum) { return __umul24(sum + half, mul) >> 22; };
      const uint32_t a_cur = front_pack_channels<YCBCR>(mean(w0.x), mean(w0.y), mean(w1.x), mean(w1.y));
      s_ex[4 * nt + tid] = a_cur;
      bl_prev[0] = bl[0]; bl_prev[1] = bl[1];
      brl_prev[0] = brl[0]; brl_prev[1] = brl[1];
      __syncthreads();
      const uint32_t a_l = s_ex[4 * nt + (u > 0 ? tid - 1 : tid)];
      const uint32_t a_r = s_ex[4 * nt + min(u + 1, cols - 1)];
      if (t == 0) { avg_prev[0] = a_l; avg_prev[1] = a_cur; avg_prev[2] = a_r; }   // (row -1 reads as row 0)
      low_cur[0] = front_blend(avg_prev[0], avg_prev[1], a_l, a_cur);
      low_cur[1] = front_blend(avg_prev[1], avg_prev[2], a_cur, a_r);
      if (u == cols - 1) low_cur[1] = low_cur[0];
      avg_prev[0] = a_l; avg_prev[1] = a_cur; avg_prev[2] = a_r;
      if (valid && t >= v0 && t < v1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          avg_f[(size_t)c * chan + (size_t)t * cols + u] = (uint8_t)(a_cur >> (8 * c));
          low_f[(size_t)c * chan + (size_t)t * cols + u] = (uint8_t)(low_cur[0] >> (8 * c));
        }
      }
      // (No third barrier.  The sums' words are next written in front of the next step's first
      // barrier: by then every wavefront has passed this step's SECOND barrier, which it reached with
      // its reads of the sums done.  The averages' word is next written behind the next step's first
      // barrier, which nobody passes before everybody has read this step's averages.)
    }
    const int v = t - 1;
    bool requested = false;   // has tile row t + 1 been requested (and row t parked) inside the transform?
    if (v >= v0 && v < v1) {
      // ---- transform of tile row v: pixels from the parking slot, low-res rows v (low_prev) and v + 1 (low_cur) ----
      uint32_t lr0[4], lr8[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        lr0[c] = ((low_prev[0] >> (8 * c)) & 255u) | (((low_prev[1] >> (8 * c)) & 255u) << 8);
        lr8[c] = ((low_cur[0] >> (8 * c)) & 255u) | (((low_cur[1] >> (8 * c)) & 255u) << 8);
      }
      const uint32_t row_off = (uint32_t)v * (uint32_t)g.row_block;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int cA = YCBCR ? (pr == 0 ? 2 : 0) : (pr == 0 ? 0 : 1);
        const int cB = YCBCR ? (pr == 0 ? 1 : 3) : (pr == 0 ? 2 : 3);
        pk16 b[64];
        {
          uint32_t LA[2][8], LB[2][8];
          lowres_quads_e(lr0[cA], lr8[cA], LA);
          lowres_quads_e(lr0[cB], lr8[cB], LB);
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const uint4 q = slot[k * 64];
            const uint32_t p4[4] = {q.x, q.y, q.z, q.w};
            const int y = k >> 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int x = (k & 1) * 4 + j;
              const uint32_t sel = 0x0c000c00u | (uint32_t)(y & 3) | ((uint32_t)(4 + (y & 3)) << 16);
              const pk16 lo = __builtin_bit_cast(pk16, __builtin_amdgcn_perm(LB[y >> 2][x], LA[y >> 2][x], sel));
              const pk16 pv = pr == 0 ? pix_pair<YCBCR, 0>(p4[j]) : pix_pair<YCBCR, 1>(p4[j]);
              b[y * 8 + x] = pv - lo;
            }
          }
        }
        if (pr == 1 && have) {
          // The slot has been read for the last time: tile row t takes it, and row t + 1 is requested --
          // the loads fly under this pair's WHT, quantiser and stores.
          park_row();
          if (t + 1 <= t_end && t + 1 < g.rows) load_row(t + 1);
          requested = true;
        }
#pragma unroll
        for (int y = 0; y < 8; ++y)
          wht8_pk(b[y * 8 + 0], b[y * 8 + 1], b[y * 8 + 2], b[y * 8 + 3], b[y * 8 + 4], b[y * 8 + 5],
                  b[y * 8 + 6], b[y * 8 + 7]);
#pragma unroll
        for (int x = 0; x < 8; ++x)
          wht8_pk(b[x], b[8 + x], b[16 + x], b[24 + x], b[32 + x], b[40 + x], b[48 + x], b[56 + x]);
        const uint32_t offA = row_off + (uint32_t)(cA * 64 * cols), offB = row_off + (uint32_t)(cB * 64 * cols);
        const int qt = (YCBCR && pr == 0) ? 1 : 0;
        // (quantise / compand / store in groups of coefficients: see k_pix_fwd)
        auto group = [&](auto i0c, auto nc) {
          constexpr int I0 = decltype(i0c)::value, N = decltype(nc)::value;
          pk16 q[N];
          upk16 top = {0, 0};
          for_seq<N>([&](auto kc) {
            constexpr int k = decltype(kc)::value, i = I0 + k, pos = kScan[i];
            const pk16 x = b[pos];
            const pk16 fifteen = {15, 15};
            const pk16 sign = x >> fifteen;
            const pk16 rr = __builtin_bit_cast(pk16, qw[(0 * 2 + qt) * 64 + i]), kk = __builtin_bit_cast(pk16, qw[(1 * 2 + qt) * 64 + i]);
            const pk16 ss = __builtin_bit_cast(pk16, qw[(2 * 2 + qt) * 64 + i]);
            q[k] = (sign * kk + x + rr) >> ss;
            const upk16 fifty = {50, 50};
            top = __builtin_elementwise_max(top, (upk16)(__builtin_bit_cast(upk16, q[k]) + fifty));
          });
          const upk16 hundred = {100, 100};
          const uint32_t over = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(top, hundred));
          if (__builtin_expect(__any(over != 0u), 0)) {
            for_seq<N>([&](auto kc) {
              constexpr int k = decltype(kc)::value;
              const pk16 fifteen = {15, 15};
              const pk16 sign = q[k] >> fifteen;
              const pk16 mag = (q[k] ^ sign) - sign;
              const uint32_t ma = min((uint32_t)(uint16_t)mag.x, (uint32_t)(kPixLut - 1));
              const uint32_t mb = min((uint32_t)(uint16_t)mag.y, (uint32_t)(kPixLut - 1));
              pk16 code;
              code.x = (short)s_lut[ma];
              code.y = (short)s_lut[mb];
              q[k] = (code ^ sign) - sign;
            });
          }
          if (valid) {
            for_seq<N>([&](auto kc) {
              constexpr int k = decltype(kc)::value, i = I0 + k;
              __builtin_amdgcn_raw_buffer_store_b8((uint8_t)q[k].x, sym_rsrc, (uint32_t)uc, offA + (uint32_t)(i * cols), 0);
              __builtin_amdgcn_raw_buffer_store_b8((uint8_t)q[k].y, sym_rsrc, (uint32_t)uc, offB + (uint32_t)(i * cols), 0);
            });
          }
        };
        using std::integral_constant;
        group(integral_constant<int, 0>{}, integral_constant<int, 4>{});
        group(integral_constant<int, 4>{}, integral_constant<int, 4>{});
        group(integral_constant<int, 8>{}, integral_constant<int, 4>{});
        group(integral_constant<int, 12>{}, integral_constant<int, 4>{});
        group(integral_constant<int, 16>{}, integral_constant<int, 16>{});
        group(integral_constant<int, 32>{}, integral_constant<int, 16>{});
        group(integral_constant<int, 48>{}, integral_constant<int, 16>{});
      }
    }
    if (!requested && have) {
      park_row();
      if (t + 1 <= t_end && t + 1 < g.rows) load_row(t + 1);
    }
    low_prev[0] = low_cur[0]; low_prev[1] = low_cur[1];
  }
}

// ---------------------------------------------------------------------------
// Entropy coder helpers.
// ---------------------------------------------------------------------------

// LDS stores of this wavefront visible to its own later loads (no workgroup barrier).
__device__ __forceinline__ void wave_lds_sync_e() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct ZR {   // zero-run summary of a symbol range
  int tz;     // trailing zeros
  int az;     // 1 when the range is all zeros
};
__device__ __forceinline__ ZR zr_combine(ZR l, ZR r) {
  ZR o;
  o.tz = r.az ? l.tz + r.tz : r.tz;
  o.az = l.az & r.az;
  return o;
}

// Wavefront scans on the DPP path (one VALU operation per step where __shfl_up is a
// ds_bpermute with its address arithmetic and a select): four row_shr steps scan the
// rows of 16 lanes, row_bcast:15 / row_bcast:31 carry the row totals across.  A lane
// without a source keeps `identity` (bound_ctrl off / the row mask).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_from(uint32_t identity, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xf, false);
}
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppBcast15 = 0x142, kDppBcast31 = 0x143, kDppWaveShr1 = 0x138;

// Inclusive add scan over the wavefront; lane 63 ends up with the total.
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
  v += dpp_from<kDppRowShr1, 0xf>(0, v);
  v += dpp_from<kDppRowShr2, 0xf>(0, v);
  v += dpp_from<kDppRowShr4, 0xf>(0, v);
  v += dpp_from<kDppRowShr8, 0xf>(0, v);
  v += dpp_from<kDppBcast15, 0xa>(0, v);
  v += dpp_from<kDppBcast31, 0xc>(0, v);
  return v;
}

// A zero-run summary in one word: trailing zeros | all-zero flag << 31; combining is
// r + ((l + 2^31) & sign(r)): r on its own unless r is all zeros, else both counts and
// l's flag.
constexpr uint32_t kZrIdentity = 0x80000000u;
__device__ __forceinline__ uint32_t zr_pack(ZR z) { return (uint32_t)z.tz | ((uint32_t)z.az << 31); }
__device__ __forceinline__ ZR zr_unpack(uint32_t p) {
  ZR z;
  z.tz = (int)(p & 0x7fffffffu);
  z.az = (int)(p >> 31);
  return z;
}
__device__ __forceinline__ uint32_t zrp_combine(uint32_t l, uint32_t r) {
  return r + ((l + 0x80000000u) & (uint32_t)((int)r >> 31));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t zrp_step(uint32_t v) {
  return zrp_combine(dpp_from<CTRL, ROW_MASK>(kZrIdentity, v), v);
}
// Inclusive scan of packed summaries over the wavefront.
__device__ __forceinline__ uint32_t wave_scan_zrp(uint32_t v) {
  v = zrp_step<kDppRowShr1, 0xf>(v);
  v = zrp_step<kDppRowShr2, 0xf>(v);
  v = zrp_step<kDppRowShr4, 0xf>(v);
  v = zrp_step<kDppRowShr8, 0xf>(v);
  v = zrp_step<kDppBcast15, 0xa>(v);
  v = zrp_step<kDppBcast31, 0xc>(v);
  return v;
}
// The value of the lane before (lane 0: identity).
__device__ __forceinline__ uint32_t wave_prev(uint32_t identity, uint32_t v) {
  return dpp_from<kDppWaveShr1, 0xf>(identity, v);
}

// Exclusive block scan (256 threads) of zero-run summaries; `carry` is the state
// before this block of symbols.  Returns the exclusive prefix; *total receives
// the state after the block.  `sm` needs NW entries (NW wavefronts in the workgroup).
// ALT: the caller alternates between two `sm` buffers from call to call, so the barrier
// that protects sm against the next call's stores is not needed (a wavefront can only be
// two calls ahead of another once that one has passed the barrier of the call between).
template <int NW = 4, bool ALT = false>
__device__ __forceinline__ ZR block_scan_zr(ZR mine, ZR carry, ZR *sm, ZR *total) {
  const int lane = lane_id(), wave = wave_id();
  const uint32_t inclp = wave_scan_zrp(zr_pack(mine));
  if (lane == 63) sm[wave] = zr_unpack(inclp);
  const ZR ex = zr_unpack(wave_prev(kZrIdentity, inclp));
  __syncthreads();
  ZR pre = carry, tot = carry;
  for (int w = 0; w < NW; ++w) {
    if (w < wave) pre = zr_combine(pre, sm[w]);
    tot = zr_combine(tot, sm[w]);
  }
  if (!ALT) __syncthreads();
  *total = tot;
  return zr_combine(pre, ex);
}

// Exclusive block scan (256 threads) of a bit count; *total = block sum.
__device__ __forceinline__ uint32_t block_scan_u32(uint32_t v, uint32_t *sm, uint32_t *total) {
  const int lane = lane_id(), wave = wave_id();
  const uint32_t incl = wave_scan_add(v);
  if (lane == 63) sm[wave] = incl;
  __syncthreads();
  uint32_t pre = 0, tot = 0;
  for (int w = 0; w < 4; ++w) {
    if (w < wave) pre += sm[w];
    tot += sm[w];
  }
  __syncthreads();
  *total = tot;
  return pre + incl - v;
}

// Greedy zero-run tokens: from the run start in steps of 16662, remainder last
// (huffman_enc.cpp:111-141; trap T6).  f(symbol, extra_bits, extra_value).
template <class F>
__device__ __forceinline__ void emit_run(int len, F &&f) {
  while (len >= 16662) { f(260, 14, 16662 - 279); len -= 16662; }
  if (len == 0) return;
  if (len == 1) f(0, 0, 0);
  else if (len == 2) f(256, 0, 0);
  else if (len <= 6) f(257, 2, len - 3);
  else if (len <= 22) f(258, 4, len - 7);
  else if (len <= 278) f(259, 8, len - 23);
  else f(260, 14, len - 279);
}

// Bit k of the result is set when symbol k (byte k of the 16-byte chunk) is
// non-zero; symbols at or beyond nvalid read as zero.
__device__ __forceinline__ uint32_t nonzero_mask16(const uint32_t w[4], int nvalid) {
  // Byte k of f = 0x80 where byte k != 0; eight flags become eight mask bits with two dot
  // products (weights 1, 2, 4, 8 and 16, 32, 64, 128, chained through the accumulator): 128 x the
  // byte of the mask, shifted down once for all sixteen.
  uint32_t f[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) f[q] = (((w[q] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w[q]) & 0x80808080u;
  const uint32_t lo = __builtin_amdgcn_udot4(f[1], 0x80402010u, __builtin_amdgcn_udot4(f[0], 0x08040201u, 0u, false), false);
  const uint32_t hi = __builtin_amdgcn_udot4(f[3], 0x80402010u, __builtin_amdgcn_udot4(f[2], 0x08040201u, 0u, false), false);
  const uint32_t m = (lo | (hi << 8)) >> 7;
  return nvalid >= 16 ? m : (m & ((1u << nvalid) - 1u));
}

__device__ __forceinline__ int symbol_at(const uint32_t w[4], int k) {
  const unsigned long long lo = ((unsigned long long)w[1] << 32) | w[0];
  const unsigned long long hi = ((unsigned long long)w[3] << 32) | w[2];
  return (int)(((k < 8 ? lo : hi) >> ((k & 7) * 8)) & 255ull);
}

// Walk one thread's 16 symbols: iterate over the NON-ZERO symbols only (set bits
// of the mask); the zeros in front of each are one run.  A zero run is tokenised
// by the thread that holds the run's LAST zero (the run start comes in through
// run_in), which keeps token order == thread order with a forward scan only.
// f(symbol, extra_bits, extra_value).
template <class F>
__device__ __forceinline__ void walk16(const uint32_t w[4], uint32_t mask, int nvalid, int run_in,
                                       bool flush, F &&f) {
  int prev = -1 - run_in;   // the zeros in front of the chunk count as positions before it
  uint32_t m = mask;
  LoopCount lc;
  while (m) {
    HIMG_REGION_BEGIN("enc.walk");
    lc.step();
    const int k = __ffs(m) - 1;
    m &= m - 1;
    const int run = k - prev - 1;
    if (run) emit_run(run, f);
    f(symbol_at(w, k), 0, 0);
    prev = k;
  }
  if (flush) {
    const int run = nvalid - 1 - prev;
    if (run) emit_run(run, f);
  }
}

// The same walk with the common case folded: a non-zero symbol preceded by at
// most kPairRuns - 1 zeros is handed to fp as ONE table entry (run token and
// literal merged, see k_emit / k_tok_hist); everything else goes token by token
// to f.  pair_tab[r * 256 + sym]; an entry of 0 means "not merged".
constexpr int kPairRuns = 7;
constexpr int kRunTab = 279;   // run_tab[r]: the token of a run of r zeros (r < 279: symbols 0, 256..259)
template <class FP, class F>
__device__ __forceinline__ void walk16_pairs(const uint32_t w[4], uint32_t mask, int nvalid,
                                             int run_in, bool flush, const uint32_t *pair_tab,
                                             const uint32_t *run_tab, FP &&fp, F &&f) {
  int prev = -1 - run_in;   // the zeros in front of the chunk count as positions before it
  uint32_t m = mask;
  LoopCount lc;
  while (m) {
    HIMG_REGION_BEGIN("enc.walk");
    lc.step();
    const int k = __ffs(m) - 1;
    m &= m - 1;
    const int run = k - prev - 1;
    const int sym = symbol_at(w, k);
    prev = k;
    const uint32_t pair = run < kPairRuns ? pair_tab[run * 256 + sym] : 0u;
    if (__builtin_expect(pair != 0, 1)) {
      fp(pair);
    } else {
      // Medium path: the run token (code + extra bits) from a table indexed by the
      // run length, then the literal -- no classification branches.
      const uint32_t rt = (run_tab && run < kRunTab) ? run_tab[run] : 0u;
      if (rt) fp(rt);
      else if (run) emit_run(run, f);
      f(sym, 0, 0);
    }
  }
  if (flush) {
    const int run = nvalid - 1 - prev;
    if (run) emit_run(run, f);
  }
}

// The walk of k_emit's fast path: the same tokens as walk16_pairs, with as few
// branches as the format allows (every divergent branch here is paid by the whole
// wavefront on almost every token: some lane always takes it).  pair_tab has one row
// more, all zero, for runs of kPairRuns zeros or more; run_tab one entry more (0).
// put(bits, n) appends n <= 32 bits (n == 0 is a no-op); f as in walk16.
template <class PUT, class F>
__device__ __forceinline__ void walk16_emit(const uint32_t w[4], uint32_t mask, int nvalid, int run_in,
                                            bool flush, const uint32_t *pair_tab, const uint32_t *run_tab,
                                            const unsigned long long *code_len, PUT &&put, F &&f) {
  int prev = -1 - run_in;   // the zeros in front of the chunk count as positions before it
  uint32_t m = mask;
  LoopCount lc;
  while (m) {
    HIMG_REGION_BEGIN("enc.walk");
    lc.step();
    const int k = __ffs(m) - 1;
    m &= m - 1;
    const int run = k - prev - 1;
    const int sym = symbol_at(w, k);
    prev = k;
    const uint32_t pair = pair_tab[min(run, kPairRuns) * 256 + sym];
    if (__builtin_expect(pair != 0, 1)) {
      put(pair & 0xffffffu, (int)(pair >> 24));
    } else {
      const uint32_t rt = run_tab[min(run, kRunTab)];
      if (__builtin_expect(rt == 0 && run != 0, 0)) emit_run(run, f);   // 279 zeros or more, or a token beyond 24 bits
      // The run token and the literal as ONE put where they fit 32 bits together -- they do
      // whenever the literal is short, and a literal behind seven zeros or more is (|s| <= 4 in
      // every such token of the benchmark frames; 86 % of a wavefront's 1024-symbol iterations
      // hold such a token: the sparse high-frequency rows).  -3 % VALU instructions, -6 % LDS
      // (PMC); the kernel's time does not move (5.63 ms per 128 frames either way).
      const unsigned long long cl = code_len[sym];
      const int nr = (int)(rt >> 24), nl = (int)(cl >> 32);
      if (__builtin_expect(nr + nl <= 32, 1)) {
        put((rt & 0xffffffu) | ((uint32_t)cl << nr), nr + nl);
      } else {
        put(rt & 0xffffffu, nr);
        put((uint32_t)cl, nl);
      }
    }
    HIMG_REGION_END("enc.walk");
  }
  lc.done(1);
  if (flush) {
    const int run = nvalid - 1 - prev;
    if (run) emit_run(run, f);
  }
}

// Zero-run summary of one thread's chunk.  Chunks with fewer than 16 valid
// symbols only occur at the very end of a span; an empty chunk is the identity.
__device__ __forceinline__ ZR summarize16(uint32_t mask, int nvalid) {
  ZR z;
  z.tz = mask ? nvalid - (32 - __clz(mask)) : nvalid;
  z.az = mask ? 0 : 1;
  return z;
}

// Load 16 symbols of a span (zero padded beyond `remaining`).
__device__ __forceinline__ void load16(const uint8_t *p, long long remaining, uint32_t w[4]) {
  if (remaining >= 16 && ((uintptr_t)p & 15) == 0) {
    const uint4 q = *reinterpret_cast<const uint4 *>(p);
    w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
  } else {
    w[0] = w[1] = w[2] = w[3] = 0;
    for (int k = 0; k < 16; ++k)
      if (k < remaining) w[k >> 2] |= (uint32_t)p[k] << ((k & 3) * 8);
  }
}

// Span descriptor shared by k_tok_hist / k_emit.
struct Span {
  const uint8_t *sym;   // first symbol of the span
  int len;              // symbols in the span
  bool last_of_block;   // the span holds the block's last symbol
  bool is_lres;
  int index;            // index into the per-frame span arrays
};

__device__ __forceinline__ Span get_span(const Geom &g, const EncWs &ws, int sp, int f) {
  Span s;
  s.index = sp;
  if (sp < g.lres_spans) {
    const int start = sp * kLresSpan;
    s.sym = ws.lres_sym + (size_t)f * ws.lres_stride + start;
    s.len = min(kLresSpan, g.lres_size - start);
    s.last_of_block = (sp == g.lres_spans - 1);
    s.is_lres = true;
  } else {
    const int r = sp - g.lres_spans;
    s.sym = ws.fres_sym + (size_t)f * ws.fres_stride + (size_t)r * g.row_block;
    s.len = g.row_block;
    s.last_of_block = true;
    s.is_lres = false;
  }
  return s;
}

// Zeros immediately before LRES span `sp` (0 for FRES rows: runs never cross
// a block row, huffman_enc.cpp:105-106).
__device__ __forceinline__ int span_carry_in(const Geom &g, const EncWs &ws, int sp, int f) {
  if (sp >= g.lres_spans) return 0;
  const uint32_t *tr = ws.lres_trail + (size_t)f * g.lres_spans;
  int t = 0;
  for (int k = sp - 1; k >= 0; --k) {
    const uint32_t x = tr[k];
    t += (int)(x & 0x7fffffffu);
    if (!(x >> 31)) break;
  }
  return t;
}

// k_lres_summary: trailing-zero summary of every LRES span.
__global__ __launch_bounds__(256) void k_lres_summary(Geom g, EncWs ws) {
  __shared__ ZR sm[4];
  const int sp = blockIdx.x, f = blockIdx.y;
  const Span s = get_span(g, ws, sp, f);
  ZR carry = {0, 1}, total = carry;
  for (int base = 0; base < s.len; base += kIterSyms) {
    const int off = base + threadIdx.x * 16;
    const int nvalid = max(0, min(16, s.len - off));
    uint32_t w[4];
    load16(s.sym + off, s.len - off, w);
    const ZR mine = summarize16(nonzero_mask16(w, nvalid), nvalid);
    block_scan_zr(mine, carry, sm, &total);
    carry = total;
  }
  if (threadIdx.x == 0)
    ws.lres_trail[(size_t)f * g.lres_spans + sp] =
        (uint32_t)total.tz | (total.az ? 0x80000000u : 0u);
}

// k_tok_hist: token histogram of one span (huffman_enc.cpp:98-144).
// A non-zero symbol preceded by fewer than kPairRuns zeros -- almost every token
// pair -- is counted with ONE LDS atomic in a 2-D histogram [run][symbol] (which
// also spreads the hot symbols over more addresses); the pairs are folded into
// the 261 token bins at the end.  Longer runs are counted token by token.
// NT lanes per span: 256 (batches: eight workgroups per CU), or 1024 for a SINGLE frame --
// its 512 rows are 512 workgroups, a quarter of the chip's slots at 256 lanes, and a row is
// 16 iterations of three barriers each; with 1024 lanes it is four iterations and the chip
// is full (latency: 84 -> 3x us).
template <int NT>
__global__ __launch_bounds__(NT) void k_tok_hist(Geom g, EncWs ws, int sp0) {
  __shared__ uint32_t hist[kHistStride];
  // [zeros in front, kPairRuns = that many or more][literal]; rows 257 words apart: the
  // hot literals (+-1, +-2) of the eight rows then lie in different banks.
  __shared__ uint32_t hist2[kPairRuns + 1][257];
  __shared__ uint32_t hrun[kRunTab + 1];   // runs of kPairRuns..278 zeros, by exact length
  __shared__ uint32_t s_sym[8 * NT];      // [word][lane]: the lane's 32 symbols of this iteration
  __shared__ ZR sm[2][NT / 64];           // (alternating: one barrier per iteration, see block_scan_zr)
  const int sp = blockIdx.x + sp0, f = blockIdx.y;
  const Span s = get_span(g, ws, sp, f);
  for (int k = threadIdx.x; k < kHistStride; k += NT) hist[k] = 0;
  for (int k = threadIdx.x; k < (kPairRuns + 1) * 257; k += NT) (&hist2[0][0])[k] = 0;
  for (int k = threadIdx.x; k < kRunTab + 1; k += NT) hrun[k] = 0;
  ZR carry;
  carry.tz = span_carry_in(g, ws, sp, f);
  carry.az = 0;
  __syncthreads();
  // 32 symbols per lane and iteration: the wave waits for its busiest lane, and
  // with 32 symbols the busiest lane is 1.6x the mean instead of 1.9x with 16; the
  // scans and the barrier are paid half as often per symbol.  The lane's symbols sit
  // in LDS (transposed), where the walk fetches them by position.
  int par = 0;
  for (int base = 0; base < s.len; base += 32 * NT, par ^= 1) {
    const int off = base + threadIdx.x * 32;
    const int nvalid = max(0, min(32, s.len - off));
    uint32_t w[8];
    load16(s.sym + off, s.len - off, w);
    load16(s.sym + off + 16, s.len - off - 16, w + 4);
#pragma unroll
    for (int q = 0; q < 8; ++q) s_sym[q * NT + threadIdx.x] = w[q];
    const uint32_t mask = nonzero_mask16(w, min(nvalid, 16)) | (nonzero_mask16(w + 4, max(nvalid - 16, 0)) << 16);
    ZR mine;
    mine.tz = mask ? nvalid - (32 - __clz(mask)) : nvalid;
    mine.az = mask ? 0 : 1;
    ZR total;
    const ZR ex = block_scan_zr<NT / 64, true>(mine, carry, sm[par], &total);
    carry = total;
    carry.az = 0;
    const bool flush = s.last_of_block && nvalid > 0 && off + nvalid == s.len;
    auto one = [&](int sym, int, int) { atomicAdd(&hist[sym], 1u); };
    const uint8_t *mysym = reinterpret_cast<const uint8_t *>(s_sym) + threadIdx.x * 4;
    int prev = -1 - ex.tz;   // the zeros in front of the chunk count as positions before it
    uint32_t m = mask;
    while (m) {
      const int k = __ffs(m) - 1;
      m &= m - 1;
      const int run = k - prev - 1;
      const int sym = mysym[(k >> 2) * (4 * NT) + (k & 3)];
      prev = k;
      // The literal always counts in the 2-D histogram (row kPairRuns: after a longer
      // run -- no branch around the common case); a longer run counts on its own.
      atomicAdd(&hist2[min(run, kPairRuns)][sym], 1u);
      if (__builtin_expect(run >= kPairRuns, 0)) {
        if (run < kRunTab) atomicAdd(&hrun[run], 1u);
        else emit_run(run, one);
      }
    }
    if (flush) {
      const int run = nvalid - 1 - prev;
      if (run) emit_run(run, one);
    }
  }
  __syncthreads();
  // Fold the pairs: the literal of every pair, and its run token (huffman_enc.cpp:
  // 111-141: one zero = literal 0, two = 256, three to six = 257).
  if (threadIdx.x < 256) {
    const int sym = threadIdx.x;   // one lane per literal value
    uint32_t lit = 0, r1 = hist2[1][sym], r2 = hist2[2][sym], r3 = 0;
#pragma unroll
    for (int r = 0; r <= kPairRuns; ++r) lit += hist2[r][sym];
#pragma unroll
    for (int r = 3; r < kPairRuns; ++r) r3 += hist2[r][sym];
    if (lit) atomicAdd(&hist[sym], lit);
    if (r1) atomicAdd(&hist[0], r1);
    if (r2) atomicAdd(&hist[256], r2);
    if (r3) atomicAdd(&hist[257], r3);
  }
  for (int r = kPairRuns + (int)threadIdx.x; r < kRunTab; r += NT) {   // 7..22 -> 258, 23..278 -> 259
    const uint32_t c = hrun[r];
    if (c) atomicAdd(&hist[r <= 22 ? 258 : 259], c);
  }
  __syncthreads();
  uint32_t *sh = (s.is_lres ? ws.span_hist_l + ((size_t)f * g.lres_spans + sp) * kHistStride
                            : ws.span_hist_f + ((size_t)f * g.rows + (sp - g.lres_spans)) * kHistStride);
  uint32_t *gh = ws.hist + ((size_t)f * 2 + (s.is_lres ? 0 : 1)) * kHistStride;
  for (int k = threadIdx.x; k < kHistStride; k += NT) {
    const uint32_t c = hist[k];
    sh[k] = c;
    if (c && k < kNumSym) atomicAdd(&gh[k], c);
  }
}

// ---------------------------------------------------------------------------
// k_tok: RLE tokeniser + token histogram of the FRES block rows of a batch, leaving the rows
// as a stream of 16-bit SLOTS for k_emit_tok (huffman_enc.cpp:98-144; replaces k_tok_hist for
// batches: the bit packer then reads 0.7 bytes per symbol instead of the dense plane, and
// every one of its lanes has eight tokens to pack on every step).
// A workgroup of four wavefronts per block row.  The row is cut into tok_nseg segments which
// the wavefronts take from a counter (the low-frequency segments hold several times the tokens
// of the high-frequency ones); a segment is walked 2048 symbols at a time WITHOUT workgroup
// barriers: the zeros in front of the segment come from a look back over the row, the zero-run
// state of a lane from one maximum scan (DPP), slot positions from one add scan.  A lane's 32
// symbols stay in registers and are walked as four quarters of 8 (the symbol byte by v_perm).
// Every non-zero symbol becomes one slot (literal | zeros in front << 8) and one LDS atomic in
// the 2-D histogram of k_tok_hist; runs beyond 255 zeros and the row's trailing zeros are split
// as the reference does (huffman_enc.cpp:111-141, trap T6) into even-aligned slot pairs (mark,
// length).  Slots are staged per wavefront (2.25 KiB: eight workgroups per CU) and leave as
// whole 16-byte pieces; an iteration that outgrows the buffer (more than ~56 % non-zero) is
// staged half by half.
// ---------------------------------------------------------------------------
constexpr int kTokThreads = 256;
constexpr int kTokWaves = kTokThreads / 64;
constexpr int kTokStage = 1152;            // slots of a wavefront's staging buffer (a half's worst case: 1024 + runs on their own + < 8 carried over)

__device__ __forceinline__ int run_token_symbol(int len) {   // huffman_enc.cpp:111-141
  return len == 1 ? 0 : len == 2 ? 256 : len <= 6 ? 257 : len <= 22 ? 258 : len <= 278 ? 259 : 260;
}
__device__ __forceinline__ uint32_t run_token_count(int len) { return (uint32_t)(len / 16662 + (len % 16662 != 0 ? 1 : 0)); }

// 32 symbols of a row (two 16-byte loads), zeros beyond `end`.
__device__ __forceinline__ void load32(const uint8_t *p, bool valid, uint32_t w[8]) {
  uint4 a = {0u, 0u, 0u, 0u}, b = a;
  if (valid) { a = reinterpret_cast<const uint4 *>(p)[0]; b = reinterpret_cast<const uint4 *>(p)[1]; }
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}

__global__ __launch_bounds__(kTokThreads) void k_tok(Geom g, EncWs ws, int r0) {
  __shared__ uint32_t hist[kHistStride];
  __shared__ uint32_t hist2[kPairRuns + 1][257];   // (as k_tok_hist)
  __shared__ uint32_t hrun[kRunTab + 1];
  __shared__ __attribute__((aligned(16))) uint16_t stage[kTokWaves][kTokStage];
  __shared__ uint32_t next_seg;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int row = (int)blockIdx.x + r0, f = blockIdx.y;
  const int row_block = g.row_block, seg = ws.tok_seg, nseg = ws.tok_nseg;
  const uint8_t *S = ws.fres_sym + (size_t)f * ws.fres_stride + (size_t)row * row_block;
  for (int k = tid; k < kHistStride; k += kTokThreads) hist[k] = 0;
  for (int k = tid; k < (kPairRuns + 1) * 257; k += kTokThreads) (&hist2[0][0])[k] = 0;
  for (int k = tid; k < kRunTab + 1; k += kTokThreads) hrun[k] = 0;
  if (tid == 0) next_seg = 0;
  __syncthreads();
  uint16_t *stg = stage[wv];
  for (;;) {
    // (No branch around the atomic: behind `if (lane == 0)` the compiler threads the two paths
    // through the loop body separately -- the body then runs with lane 0 alone, then with the rest.)
    const int sg = __builtin_amdgcn_readfirstlane((int)atomicAdd(&next_seg, lane == 0 ? 1u : 0u));
    if (sg >= nseg) break;
    const int p0 = sg * seg, p1 = min(p0 + seg, row_block);
    // Zeros immediately in front of the segment (runs cross segments, never block rows).
    int run_carry = 0;
    for (int q = p0 - kTokIter; q >= 0; q -= kTokIter) {
      uint32_t w[8];
      load32(S + q + 32 * lane, true, w);
      const uint32_t m = nonzero_mask16(w, 16) | (nonzero_mask16(w + 4, 16) << 16);
      const unsigned long long nzl = __ballot(m != 0);
      if (nzl == 0) { run_carry += kTokIter; continue; }
      const int hl = 63 - __clzll((long long)nzl);
      const uint32_t mh = (uint32_t)__builtin_amdgcn_readlane((int)m, hl);
      run_carry += (63 - hl) * 32 + __clz((int)mh);
      break;
    }
    const size_t seg_index = ((size_t)f * g.rows + row) * nseg + sg;
    uint16_t *seg_out = ws.tok + seg_index * (size_t)ws.tok_cap;
    uint32_t flushed = 0, c = 0;   // slots in HBM (a multiple of 8); slots waiting at the head of stg (< 8)
    // Whole 16-byte pieces of the c + added staged slots leave; what is left (< 8 slots) moves to
    // the head of the buffer.  (LDS operations of a wavefront execute in order.)
    auto flush = [&](uint32_t added) {
      wave_lds_sync_e();
      const uint32_t ntot = c + added, nch = ntot >> 3, rem = ntot & 7u;
      for (uint32_t j = lane; j < nch; j += 64)
        *reinterpret_cast<uint4 *>(seg_out + flushed + 8 * j) = reinterpret_cast<const uint4 *>(stg)[j];
      uint16_t keep = 0;
      if ((uint32_t)lane < rem) keep = stg[nch * 8 + lane];
      wave_lds_sync_e();
      if ((uint32_t)lane < rem) stg[lane] = keep;
      flushed += nch * 8;
      c = rem;
    };
    LoopCount lci;
    for (int q = p0; q < p1; q += kTokIter) {
      HIMG_REGION_BEGIN("tokr.iter");
      lci.step();
      uint32_t w[8];   // (no prefetch: eight wavefronts per SIMD hide the load, eight more registers would cost one of them)
      load32(S + q + 32 * lane, q + 32 * lane < p1, w);
      const int nsym = min(kTokIter, p1 - q);   // (a multiple of 64: a lane holds 32 symbols or none)
      const uint32_t mask = nonzero_mask16(w, 16) | (nonzero_mask16(w + 4, 16) << 16);
      // Zeros in front of every lane: one past the last non-zero symbol of the iteration so far
      // (0: none), an exclusive maximum scan on the DPP path.
      uint32_t pm = mask ? (uint32_t)(32 * lane + 32 - __clz((int)mask)) : 0u;
      pm = max(pm, dpp_from<kDppRowShr1, 0xf>(0, pm));
      pm = max(pm, dpp_from<kDppRowShr2, 0xf>(0, pm));
      pm = max(pm, dpp_from<kDppRowShr4, 0xf>(0, pm));
      pm = max(pm, dpp_from<kDppRowShr8, 0xf>(0, pm));
      pm = max(pm, dpp_from<kDppBcast15, 0xa>(0, pm));
      pm = max(pm, dpp_from<kDppBcast31, 0xc>(0, pm));
      const uint32_t pprev = wave_prev(0, pm);
      const int run_in = 32 * lane - (int)pprev + (pprev ? 0 : run_carry);
      const uint32_t plast = (uint32_t)__builtin_amdgcn_readlane((int)pm, 63);
      run_carry = plast ? nsym - (int)plast : run_carry + nsym;
      // The lane's two halves of 16 symbols, each with the zeros in front of it.  Slots of a half:
      // its non-zero symbols; a run of more than 255 zeros in front of its first one, and the
      // row's trailing zeros, as runs on their own (three slots per token of the reference).
      const uint32_t mA = mask & 0xffffu, mB = mask >> 16;
      const int rinA = run_in, rinB = mA ? __clz((int)mA) - 16 : run_in + 16;
      const int leadA = mA ? rinA + __ffs((int)mA) - 1 : 0, leadB = mB ? rinB + __ffs((int)mB) - 1 : 0;
      const bool longA = leadA > kTokMaxRun, longB = leadB > kTokMaxRun;
      int trail = 0;
      if (q + 32 * lane + 32 == row_block) trail = mB ? __clz((int)mB) - 16 : rinB + 16;
      uint32_t nA = (uint32_t)__popc(mA), nB = (uint32_t)__popc(mB);
      if (__builtin_expect(__any(longA || longB || trail != 0), 0)) {
        nA += longA ? 3u * run_token_count(leadA) : 0u;
        nB += (longB ? 3u * run_token_count(leadB) : 0u) + (trail ? 3u * run_token_count(trail) : 0u);
      }
      // One half's walk; `slot` is where the lane's slots of this half begin in stg.
      auto half = [&](uint32_t mh, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, int rin, int lead, bool lng,
                      int trl, uint32_t slot) {
        auto long_run = [&](int len) {
          while (len > 0) {
            const int piece = min(len, 16662);
            const uint32_t odd = slot & 1u;   // (what is in HBM is a multiple of 8 slots: the parity in stg is the parity in the segment)
            stg[slot + odd] = (uint16_t)kTokRunMark;
            stg[slot + odd + 1] = (uint16_t)piece;
            stg[slot + (odd ? 0 : 2)] = 0;
            slot += 3;
            atomicAdd(&hist[run_token_symbol(piece)], 1u);
            len -= piece;
          }
        };
        int prev1 = -rin;   // one past the previous non-zero symbol, relative to the half
        if (__builtin_expect(__any(lng), 0)) {
          if (lng) { long_run(lead); prev1 = __ffs((int)mh) - 1; }
        }
        uint16_t *tp = stg + slot;
        LoopCount lc;
        // Two walks of eight symbols each: the symbol byte then comes from a fixed register pair (one
        // v_perm; one walk of sixteen needed two selects and a compare in front of it, and the step
        // counts of a wavefront -- its busiest lane's -- are the same: k_tok 4.70 -> 4.55 ms).
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          uint32_t m = (mh >> (8 * qt)) & 0xffu;
          const uint32_t wl = qt ? w2 : w0, wh = qt ? w3 : w1;
          while (m) {
            HIMG_REGION_BEGIN("tokr.walk");
            lc.step();
            const int k8 = __ffs((int)m) - 1;
            m &= m - 1;
            const int k = k8 + 8 * qt;
            const int run = k - prev1;
            prev1 = k + 1;
            const uint32_t sy = __builtin_amdgcn_perm(wh, wl, (uint32_t)k8 | 0x0c0c0c00u);
            // The literal always counts in the 2-D histogram (row kPairRuns: after a longer run); a
            // longer run counts on its own, by exact length.
            atomicAdd(&(&hist2[0][0])[__umul24((uint32_t)min(run, kPairRuns), 257u) + sy], 1u);   // (24-bit multiply-add: one full-rate instruction)
            if (__builtin_expect(run >= kPairRuns, 0)) atomicAdd(&hrun[run], 1u);
            *tp++ = (uint16_t)(sy | ((uint32_t)run << 8));
            HIMG_REGION_END("tokr.walk");
          }
        }
        lc.done(4);
        if (__builtin_expect(__any(trl != 0), 0)) {
          slot = (uint32_t)(tp - stg);
          if (trl) long_run(trl);
        }
      };
      const uint32_t n = nA + nB;
      uint32_t incl_ = wave_scan_add(n);
      asm volatile("" : "+v"(incl_));   // (or `incl - n` is re-associated into the sum of the scan's six shifted parts: six moves and three adds more)
      const uint32_t incl = incl_;
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      // A dense iteration (more slots than the buffer holds) is staged in two parts: lanes 0..31,
      // then lanes 32..63 -- the slots of a part are consecutive in the stream (lane-major order).
      const bool split = c + total > (uint32_t)kTokStage;
      const uint32_t mid = split ? (uint32_t)__builtin_amdgcn_readlane((int)incl, 31) : total;
#pragma unroll 1
      for (int part = 0; part < (split ? 2 : 1); ++part) {
        if (!split || (lane >> 5) == part) {
          const uint32_t slot = c + incl - n - (part ? mid : 0u);
          half(mA, w[0], w[1], w[2], w[3], rinA, leadA, longA, 0, slot);
          half(mB, w[4], w[5], w[6], w[7], rinB, leadB, longB, trail, slot + nA);
        }
        flush(part ? total - mid : mid);
      }
      HIMG_REGION_END("tokr.iter");
    }
    lci.done(3);
    if (c) {   // the tail, padded with no-op slots to a whole piece
      if ((uint32_t)lane >= c && lane < 8) stg[lane] = 0;
      wave_lds_sync_e();
      if (lane == 0) *reinterpret_cast<uint4 *>(seg_out + flushed) = reinterpret_cast<const uint4 *>(stg)[0];
      wave_lds_sync_e();
    }
    if (lane == 0) ws.tok_cnt[seg_index] = flushed + c;
  }
  __syncthreads();
  // Fold the 2-D histogram into the 261 token bins (as k_tok_hist).
  {
    const int sy = tid;   // one lane per literal value
    uint32_t lit = 0, r1c = hist2[1][sy], r2c = hist2[2][sy], r3c = 0;
#pragma unroll
    for (int rr = 0; rr <= kPairRuns; ++rr) lit += hist2[rr][sy];
#pragma unroll
    for (int rr = 3; rr < kPairRuns; ++rr) r3c += hist2[rr][sy];
    if (lit) atomicAdd(&hist[sy], lit);
    if (r1c) atomicAdd(&hist[0], r1c);
    if (r2c) atomicAdd(&hist[256], r2c);
    if (r3c) atomicAdd(&hist[257], r3c);
  }
  for (int rr = kPairRuns + tid; rr < kRunTab; rr += kTokThreads) {   // 7..22 -> 258, 23..278 -> 259
    const uint32_t cnt = hrun[rr];
    if (cnt) atomicAdd(&hist[rr <= 22 ? 258 : 259], cnt);
  }
  __syncthreads();
  uint32_t *sh = ws.span_hist_f + ((size_t)f * g.rows + row) * kHistStride;
  uint32_t *gh = ws.hist + ((size_t)f * 2 + 1) * kHistStride;
  for (int k = tid; k < kHistStride; k += kTokThreads) {
    const uint32_t cnt = hist[k];
    sh[k] = cnt;
    if (cnt && k < kNumSym) atomicAdd(&gh[k], cnt);
  }
}

// k_tok_expand (debug read-back only): the symbols of one frame's block rows from their
// slots, one lane per row (dst is zeroed first by the caller).
__global__ __launch_bounds__(64) void k_tok_expand(Geom g, EncWs ws, int frame, uint8_t *out) {
  const int v = blockIdx.x * 64 + threadIdx.x;
  if (v >= g.rows) return;
  uint8_t *dst = out + (size_t)v * g.row_block;
  long long pos = 0;
  for (int sg = 0; sg < ws.tok_nseg; ++sg) {
    const size_t si = ((size_t)frame * g.rows + v) * ws.tok_nseg + sg;
    const uint16_t *t = ws.tok + si * (size_t)ws.tok_cap;
    const uint32_t n = min(ws.tok_cnt[si], (uint32_t)ws.tok_cap);
    for (uint32_t k = 0; k < n; ++k) {
      const uint32_t s = t[k];
      if (s == 0) continue;
      if (s == kTokRunMark) { pos += t[k + 1]; ++k; continue; }
      pos += s >> 8;
      if (pos < g.row_block) dst[pos] = (uint8_t)s;
      ++pos;
    }
  }
  if (pos != g.row_block) atomicMax(&ws.status[frame], 7);   // the slots do not cover the row exactly
}

// ---------------------------------------------------------------------------
// k_tree: one workgroup (nine wavefronts) per (stream, frame).  Builds the Huffman tree with the
// reference's tie-breaking (trap T5): repeatedly join the two lightest nodes
// under the total order (count ascending, node index DESCENDING); the lighter
// becomes child_a (bit 0).  Then a pre-order walk serialises the tree and
// assigns LSB-first codes (huffman_enc.cpp:148-180).
// ---------------------------------------------------------------------------
constexpr int kTreeThreads = 576;   // nine waves: one lane per node of the largest tree (2 * 261 - 1)
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }   // a wave-uniform value, in an SGPR

__global__ __launch_bounds__(kTreeThreads) void k_tree(EncWs ws, int strm0) {
  __shared__ int cnt[2 * kNumSym];
  __shared__ int s_cnt[kNumSym + 2];      // leaves sorted by (count, -index)
  __shared__ short s_idx[kNumSym + 2];
  __shared__ short ca[2 * kNumSym], cb[2 * kNumSym], nsym[2 * kNumSym];
  __shared__ short s_end[2 * kNumSym];    // for the first node of a run of equal counts: the node after the run
  __shared__ uint32_t bits[kTreeStride / 4];
  __shared__ int s_num;
  __shared__ int s_sz[2 * kNumSym];            // subtree size in bits of the serialised tree
  __shared__ int t_cnt[2 * 64 + 2], t_sz[2 * 64 + 2];   // one batch of the merge, in pick order (see below)
  __shared__ short t_id[2 * 64 + 2];
  static_assert(kTreeThreads >= 2 * kNumSym, "one lane per node");

  const int strm = blockIdx.x + strm0, f = blockIdx.y, lane = threadIdx.x;
  const size_t tab = ((size_t)f * 2 + strm) * kHistStride;
  const uint32_t *hist = ws.hist + tab;
  uint64_t *codes = ws.codes + tab;
  uint32_t *lens = ws.lens + tab;

  for (int k = lane; k < kTreeStride / 4; k += kTreeThreads) bits[k] = 0;
  for (int k = lane; k < kHistStride; k += kTreeThreads) { codes[k] = 0; lens[k] = 0; }
  // Leaves in ascending symbol order (huffman_enc.cpp:186-196): wave 0 compacts them.
  if (lane < 64) {
    int n = 0;
    for (int base = 0; base < kNumSym; base += 64) {
      const int k = base + lane;
      const uint32_t c = k < kNumSym ? hist[k] : 0;
      const unsigned long long mask = __ballot(c > 0);
      if (c > 0) {
        const int idx = n + __popcll(mask & ((1ull << lane) - 1ull));
        cnt[idx] = (int)c; ca[idx] = -1; cb[idx] = -1; nsym[idx] = (short)k; s_sz[idx] = 10;
      }
      n += __popcll(mask);
    }
    if (lane == 0) s_num = n;
  }
  __syncthreads();
  const int num = s_num;

  // Join the two lightest nodes until one is left (huffman_enc.cpp:199-227).  The
  // reference's scan picks them under the total order (count ascending, node index
  // DESCENDING).  Instead of searching all nodes every time (O(n^2)):
  //   * the leaves are sorted once by that order (rank sort, one lane per leaf);
  //   * internal nodes are created with non-decreasing counts, so among them the
  //     lightest is the LAST node of the leading group of equal counts.  A group is
  //     final once consumption from it starts (any later node weighs at least
  //     twice as much), and a new node can only join the last group while that
  //     group is still untouched -- so "current group [ib, ie)" plus "next group
  //     starts at inext" describes the queue, consumed from ie - 1 downwards;
  //   * ties between a leaf and an internal node go to the internal node (its
  //     index is larger).
  if (lane < num) {
    const int cj = cnt[lane];
    int rank = 0;
    for (int i = 0; i < num; ++i) {
      const int ci = cnt[i];
      rank += (ci < cj || (ci == cj && i > lane)) ? 1 : 0;
    }
    s_cnt[rank] = cj;
    s_idx[rank] = (short)lane;
  }
  __syncthreads();
  if (lane < 64 && num > 1) {
    const int un = uni(num);             // (the merge's copy of num and next, in SGPRs)
    int next = un;
    // The merge is serial, O(n), and wave 0 runs it with WAVE-UNIFORM control: every
    // value that steers the loop sits in an SGPR (uni()), so the loop is scalar
    // branches and scalar arithmetic instead of one lane's divergent code under exec
    // masks, and no LDS read is waited for on the way round: the head of the sorted
    // leaves is fetched one leaf ahead, and what the queue of internal nodes needs --
    // the count of the group that opens next and where it ends -- is known in
    // registers when the run of equal counts it belongs to is closed, or fetched when
    // the group before it opens.  Subtree sizes flow through VGPRs to their store.
    int lh = 0;                          // head of the sorted leaves
    // ---- round 5: the merge in BATCHES ----
    // The serial loop below is 181 dependent steps of ~1000 cycles for the bench frames' FRES
    // alphabet (88 us of a single frame's 0.38 ms).  Most of those steps do not depend on each
    // other: let x1, x2 be the two lightest nodes and s = count(x1) + count(x2).  Every node that
    // a merge creates from now on weighs at least s, so ALL nodes lighter than s are consumed
    // before any of them, in the reference's order (count ascending, index DESCENDING) and in
    // consecutive pairs: the batch B = {nodes with count < s}, sorted by that order, becomes the
    // pairs (B[0], B[1]), (B[2], B[3]), ... -- new nodes next, next + 1, ... with non-decreasing
    // counts, child_a the lighter one -- and an odd last one is left over, still the lightest,
    // for the next batch.  One batch is a handful of wave-wide steps: which leaves (sorted) and
    // which internal nodes (created with non-decreasing counts: a prefix of the unconsumed ones)
    // are lighter than s; each one's rank in B -- leaves are in order already, internal nodes
    // are in order once every run of equal counts is reversed (higher index first), a leaf
    // follows the internal nodes of its count (lower index) --; scatter by rank, pair up.
    // Counts roughly double from batch to batch: ~40 batches instead of 181 steps.
    // What the batch form does not take (more than 64 leaves or internal nodes in one batch that
    // no threshold separates: long runs of equal counts, e.g. uniform noise) is left to the
    // serial loop, which starts from wherever the batches stopped.
    int qh = un;                         // first unconsumed internal node (all of them: [qh, next))
    int yh = 0, yc = 0, yi = 0, ysz = 0; // the left-over node of the batch before (lighter than everything else)
    for (;;) {
      if ((un - lh) + (next - qh) + yh < 2) break;
      // the two smallest counts: among the left-over, two leaves, two internal nodes
      int hv = 0x7fffffff;
      if (lane == 0 && yh) hv = yc;
      if ((lane == 1 || lane == 2) && lh + lane - 1 < un) hv = s_cnt[lh + lane - 1];
      if ((lane == 3 || lane == 4) && qh + lane - 3 < next) hv = cnt[qh + lane - 3];
      int m1 = 0x7fffffff, m2 = 0x7fffffff;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int h = __builtin_amdgcn_readlane(hv, k);
        if (h < m1) { m2 = m1; m1 = h; } else if (h < m2) m2 = h;
      }
      int s_thr = m1 + m2;
      // at most 64 of either kind: a 65th one lighter than s lowers the threshold to its count
      if (lh + 64 < un) { const int c64 = uni(s_cnt[lh + 64]); if (c64 < s_thr) s_thr = c64; }
      if (qh + 64 < next) { const int c64 = uni(cnt[qh + 64]); if (c64 < s_thr) s_thr = c64; }
      const bool vl = lh + lane < un, vi = qh + lane < next;
      const int Lc = vl ? s_cnt[lh + lane] : 0x7fffffff, Lid = vl ? (int)s_idx[lh + lane] : 0;
      const int Ic = vi ? cnt[qh + lane] : 0x7fffffff, Isz = vi ? s_sz[qh + lane] : 0;
      const int nl = __popcll(__ballot(vl && Lc < s_thr)), ni = __popcll(__ballot(vi && Ic < s_thr));
      const int yin = (yh && yc < s_thr) ? 1 : 0;
      const int m = uni(yin + nl + ni);
      if (m < 2) break;                  // (the serial loop's case)
      // internal nodes in the reference's order: every run of equal counts reversed
      const int prevc = (int)wave_prev(0x80000000u, (uint32_t)Ic);
      const unsigned long long starts = __ballot(lane < ni && (lane == 0 || Ic != prevc));
      const int run_b = 63 - __clzll((long long)(starts & ((2ull << lane) - 1ull)));
      const unsigned long long above = (starts >> lane) >> 1;
      const int run_e = above ? lane + 1 + (__ffsll((long long)above) - 1) : ni;
      const int kb = run_b + (run_e - 1 - lane);
      int le = 0, lt = 0;                // internal nodes not heavier than this leaf / leaves lighter than this internal node
      for (int j = 0; j < ni; ++j) le += (__builtin_amdgcn_readlane(Ic, j) <= Lc) ? 1 : 0;
      for (int j = 0; j < nl; ++j) lt += (__builtin_amdgcn_readlane(Lc, j) < Ic) ? 1 : 0;
      if (lane < nl) { const int rk = yin + lane + le; t_cnt[rk] = Lc; t_id[rk] = (short)Lid; t_sz[rk] = 10; }
      if (lane < ni) { const int rk = yin + kb + lt; t_cnt[rk] = Ic; t_id[rk] = (short)(qh + lane); t_sz[rk] = Isz; }
      if (lane == 0 && yin) { t_cnt[0] = yc; t_id[0] = (short)yi; t_sz[0] = ysz; }
      wave_lds_sync_e();
      const int np = m >> 1;
      if (lane < np) {
        const int n = next + lane;
        ca[n] = t_id[2 * lane]; cb[n] = t_id[2 * lane + 1]; nsym[n] = -1;
        cnt[n] = t_cnt[2 * lane] + t_cnt[2 * lane + 1];
        s_sz[n] = 1 + t_sz[2 * lane] + t_sz[2 * lane + 1];
      }
      if (m & 1) { yh = 1; yc = uni(t_cnt[m - 1]); yi = uni((int)t_id[m - 1]); ysz = uni(t_sz[m - 1]); }
      else yh = 0;
      lh = uni(lh + nl); qh = uni(qh + ni); next = uni(next + np);
      wave_lds_sync_e();                 // (the new nodes are read by the next batch)
    }
    // ---- the serial loop, from wherever the batches stopped ----
    // Its view of the internal nodes [qh, next): runs of equal counts, s_end[first node of a run] =
    // the node behind the run; the last run may still grow.
    for (int q0 = qh; q0 < next; q0 += 64) {
      const int q = q0 + lane;
      if (q < next && (q == qh || cnt[q] != cnt[q - 1])) {
        int e = q + 1;
        const int c = cnt[q];
        while (e < next && cnt[e] == c) ++e;
        s_end[q] = (short)e;
      }
    }
    wave_lds_sync_e();
    int lc = uni(s_cnt[min(lh, un)]), li = uni((int)s_idx[min(lh, un)]);    // the head leaf (two entries of padding)
    int lc_nv = s_cnt[min(lh + 1, un)], li_nv = s_idx[min(lh + 1, un)];        // the one behind it, still in flight (VGPRs)
    int run0 = next, run_c = -1;         // the run of equal counts at the end of the nodes: [run0, next)
    if (qh < next) {
      run_c = uni(cnt[next - 1]);
      int r0v = next - 1;
      while (r0v > qh && uni(cnt[r0v - 1]) == run_c) --r0v;
      run0 = r0v;
    }
    int ib = qh, ie = qh, inext = qh;    // current internal group [ib, ie), next group from inext
    int g = 0;                           // count of the current group
    int pg_v = 0, pe_v = 0;              // count / end of the group at inext (VGPRs)
    if (inext < run0) { pg_v = cnt[inext]; pe_v = s_end[inext]; }
    for (int left = (un - lh) + (next - qh) + yh; left > 1; --left) {
      int pick[2], pc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t == 0 && yh) {              // the batches' left-over node: lighter than everything else
          pick[0] = yi; pc[0] = yc; yh = 0;
          continue;
        }
        if (ib == ie && inext < next) {  // open the next group of equal counts
          ib = inext;
          if (ib >= run0) { g = run_c; ie = inext = next; }   // the last run (it may still grow, below)
          else { g = uni(pg_v); ie = inext = uni(pe_v); }
          if (inext < run0) { pg_v = cnt[inext]; pe_v = s_end[inext]; }   // the group after it, for later
        }
        const bool have_int = ib < ie, have_leaf = lh < un;
        if (have_int && (!have_leaf || g <= lc)) {
          pick[t] = --ie;
          pc[t] = g;
        } else {
          pick[t] = li;
          pc[t] = lc;
          ++lh;
          lc = uni(lc_nv); li = uni(li_nv);
          const int nx = min(lh + 1, un);        // (the arrays hold two entries of padding)
          lc_nv = s_cnt[nx]; li_nv = s_idx[nx];
        }
      }
      const int c = pc[0] + pc[1];
      // Subtree size in bits of the serialisation (leaf: 1 + 9, branch: 1 + its children):
      // the children exist already.
      const int sz = 1 + s_sz[pick[0]] + s_sz[pick[1]];
      const bool grows = c == run_c;     // the new node joins the last run (untouched, or it would weigh more)
      if (lane == 0) {
        ca[next] = (short)pick[0]; cb[next] = (short)pick[1]; nsym[next] = -1;
        cnt[next] = c;
        s_sz[next] = sz;
        if (!grows) s_end[run0] = (short)next;
      }
      if (!grows) {
        if (run0 == inext) { pg_v = run_c; pe_v = next; }   // the run that closes is the group that opens next
        run0 = next; run_c = c;
      }
      // The new node extends the current group iff that group is the last one,
      // untouched, and of the same count.
      if (ib < ie && ie == inext && inext == next && g == c) ie = inext = next + 1;
      ++next;
      // (what steers the loop stays in SGPRs)
      lh = uni(lh); ib = uni(ib); ie = uni(ie); inext = uni(inext); g = uni(g); run0 = uni(run0); run_c = uni(run_c);
      next = uni(next);
    }
  }
  __syncthreads();
  const int next = num > 1 ? 2 * num - 1 : num;

  // Codes and the serialised tree (huffman_enc.cpp:148-180) without a serial walk.
  // A node's depth, code (LSB first: taking child_b at depth d sets bit d) and bit
  // position in the pre-order serialisation are sums / concatenations along its path
  // from the root: position = sum over the path of (1 + the left sibling's subtree size
  // for a right child), code = the path's side bits.  Path sums are what POINTER JUMPING
  // computes in log2(depth) rounds: every node (one lane each) keeps an ancestor, the
  // length, the side bits and the position offset of the path segment up to it, and in a
  // round takes over its ancestor's segment and ancestor -- nine rounds cover any tree
  // of 261 leaves.  Subtree sizes (leaf: 1 + 9 bits, branch: 1 + its children) come from
  // the merge itself, which creates parents after their children.
  __shared__ short s_par[2 * kNumSym];
  __shared__ uint8_t s_side[2 * kNumSym];
  __shared__ short s_depth[2 * kNumSym];       // length of the node's segment; at the end: its depth
  __shared__ int s_pos[2 * kNumSym];           // offset along the segment; at the end: bit position in the serialisation
  __shared__ unsigned long long s_code[2 * kNumSym];   // side bits of the segment; at the end: the code
  __shared__ short s_anc[2 * kNumSym];         // the node above the segment (the root: itself)
  __shared__ int s_changed[2], s_err;
  const int v = lane;
  if (v < next) s_par[v] = -1;
  if (lane == 0) { s_err = 0; s_changed[0] = 0; s_changed[1] = 0; }
  __syncthreads();
  if (v >= num && v < next) {   // internal nodes name their children
    s_par[ca[v]] = (short)v; s_side[ca[v]] = 0;
    s_par[cb[v]] = (short)v; s_side[cb[v]] = 1;
  }
  __syncthreads();
  const int root = next - 1;
  if (v < next) {
    const int p = s_par[v];
    if (p < 0) {   // the root (its record is final)
      // A single symbol is one leaf with a 1-bit code 0 (huffman_enc.cpp:231-237).
      s_anc[v] = (short)v; s_depth[v] = (short)(num == 1 ? 1 : 0); s_code[v] = 0; s_pos[v] = 0;
    } else {
      s_anc[v] = (short)p;
      s_depth[v] = 1;
      s_code[v] = s_side[v];
      s_pos[v] = 1 + (s_side[v] ? s_sz[ca[p]] : 0);
    }
  }
  __syncthreads();
  for (int round = 0; round < 12; ++round) {
    // Every node whose segment does not start at the root yet takes over its ancestor's
    // segment: all records are read first, then written, so a round only sees the round
    // before.
    short na = -1, nd = 0;
    int np = 0;
    unsigned long long nc = 0;
    if (v < next) {
      const int a = s_anc[v];
      if (a != root && a != v) {
        const int la = s_depth[a], lv = s_depth[v];
        na = s_anc[a];
        nd = (short)min(la + lv, 255);
        nc = s_code[a] | (la < 64 ? s_code[v] << la : 0ull);
        np = s_pos[a] + s_pos[v];
      }
    }
    __syncthreads();
    if (na >= 0) {
      s_anc[v] = na; s_depth[v] = nd; s_code[v] = nc; s_pos[v] = np;
      s_changed[round & 1] = 1;
    }
    if (lane == 0) s_changed[(round & 1) ^ 1] = 0;   // (read two barriers ago)
    __syncthreads();
    if (!s_changed[round & 1]) break;
  }
  // Leaves: code table entries and their 10 bits of the serialisation (branches are 0 bits).
  if (v < num) {
    const int sym = nsym[v], d = s_depth[v];
    codes[sym] = s_code[v];
    lens[sym] = (uint32_t)d;
    if (d > kMaxCodeLen) s_err = 1;
    const uint32_t pos = (uint32_t)s_pos[v];
    const unsigned long long val = (1ull | ((unsigned long long)sym << 1)) << (pos & 31);
    atomicOr(&bits[pos >> 5], (uint32_t)val);
    if (val >> 32) atomicOr(&bits[(pos >> 5) + 1], (uint32_t)(val >> 32));
  }
  __syncthreads();
  const uint32_t nb = next > 0 ? (uint32_t)s_sz[next - 1] : 0u;
  const int nbytes = (int)((nb + 7) >> 3);
  if (lane == 0) {
    ws.tree_nbytes[(size_t)f * 2 + strm] = (nb + 7) >> 3;
    if (s_err) atomicMax(&ws.status[f], 3);  // code longer than 32 bits: outside the built scope
  }
  uint8_t *tree = ws.tree + ((size_t)f * 2 + strm) * kTreeStride;
  for (int k = lane; k < nbytes; k += kTreeThreads) tree[k] = (uint8_t)(bits[k >> 2] >> ((k & 3) * 8));
}

__device__ __forceinline__ int extra_bits_of(int sym) {
  return sym < 257 ? 0 : (sym == 257 ? 2 : (sym == 258 ? 4 : (sym == 259 ? 8 : 14)));
}

// ---------------------------------------------------------------------------
// k_span_bits: payload bits of every span = its token histogram . (code length + extra
// bits) (huffman_enc.cpp:298-338 without re-reading the symbols), one WAVEFRONT per span:
// five coalesced loads per lane and a wave reduction.  (As a loop of 261 dependent loads
// per span inside k_sizes' single workgroup this was 25 us of a single frame's encode.)
// Spans [sp0, sp1) of every frame, span index as in get_span (LRES spans, then FRES rows);
// bits_out (optional): where the FRES rows' counts go instead of ws.span_bits (row-sharded
// encode: [row - first row]).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_span_bits(Geom g, EncWs ws, int sp0, int sp1, uint32_t *bits_out) {
  const int f = blockIdx.y, lane = lane_id();
  const int sp = sp0 + (int)blockIdx.x * 4 + wave_id();
  if (sp >= sp1) return;
  const bool lres = sp < g.lres_spans;
  const uint32_t *h = lres ? ws.span_hist_l + ((size_t)f * g.lres_spans + sp) * kHistStride
                           : ws.span_hist_f + ((size_t)f * g.rows + (sp - g.lres_spans)) * kHistStride;
  const uint32_t *len = ws.lens + ((size_t)f * 2 + (lres ? 0 : 1)) * kHistStride;
  uint32_t b = 0;
#pragma unroll
  for (int j = 0; j < (kNumSym + 63) / 64; ++j) {
    const int k = lane + 64 * j;
    if (k < kNumSym) b += h[k] * (len[k] + (uint32_t)extra_bits_of(k));
  }
  b = wave_scan_add(b);   // lane 63: the span's total
  if (lane == 63) {
    if (bits_out && !lres) bits_out[sp - sp0] = b;
    else ws.span_bits[(size_t)f * (g.lres_spans + g.rows) + sp] = b;
  }
}

// ---------------------------------------------------------------------------
// k_sizes: one workgroup per frame.  Span bit counts follow from the span
// histograms and the code lengths (k_span_bits), so the symbols are not re-read.  Writes every
// container byte that is not entropy payload: RIFF/FRMT/LMAP/'LRES' head, both
// serialised trees, QCFG/FMAP/'FRES', the per-row size headers
// (huffman_enc.cpp:342-352) and all size fields (encoder.cpp:131-137,347-350).
// ---------------------------------------------------------------------------
// row_bits_in (optional): payload bits of EVERY block row, used instead of the
// local span histograms (multi-GPU: rows live on other ranks).
// fres_rel != 0: lay out the FRES rows only, relative to the first row header
// (no LRES, no container, no tree) -- every rank of a row-sharded encode runs
// this identically and emits its rows at the same relative offsets.
// [hr0, hr1): the rows whose size headers this call stores (a rank's own rows when `out`
// is shared with other ranks / devices: nobody else's bytes are touched).
__global__ __launch_bounds__(256) void k_sizes(Geom g, EncWs ws, StaticChunks sc, uint8_t *out,
                                               size_t out_stride, uint32_t *sizes,
                                               const uint32_t *row_bits_in, int fres_rel, int hr0, int hr1) {
  __shared__ uint32_t sm[4];
  const int f = blockIdx.x, tid = threadIdx.x;
  uint8_t *o = out + (size_t)f * out_stride;
  const int nsp = g.lres_spans + g.rows;
  uint64_t *bit0 = ws.span_bit0 + (size_t)f * nsp;
  uint32_t *nbits = ws.span_bits + (size_t)f * nsp;

  const uint32_t tree_l = fres_rel ? 0u : ws.tree_nbytes[(size_t)f * 2 + 0];
  const uint32_t tree_f = fres_rel ? 0u : ws.tree_nbytes[(size_t)f * 2 + 1];

  // LRES spans: one continuous bit stream after the byte-aligned tree.
  unsigned long long run = 8ull * (kHeadLen + tree_l);
  for (int base = 0; base < (fres_rel ? 0 : g.lres_spans); base += 256) {
    const int s = base + tid;
    const uint32_t b = s < g.lres_spans ? nbits[s] : 0u;   // k_span_bits
    uint32_t tot;
    const uint32_t ex = block_scan_u32(b, sm, &tot);
    if (s < g.lres_spans) bit0[s] = run + ex;
    run += tot;
  }
  const unsigned long long lres_end_bit = run;
  const uint32_t lres_bytes = (uint32_t)(((lres_end_bit + 7) >> 3) - kHeadLen);

  // FRES rows: byte aligned payloads, each behind a 2- or 4-byte size header.
  const unsigned long long fres_base =
      fres_rel ? 0ull : (unsigned long long)kHeadLen + lres_bytes + sc.mid_len;
  unsigned long long pos = fres_base + tree_f;  // running byte position
  for (int base = 0; base < g.rows; base += 256) {
    const int r = base + tid;
    uint32_t b = 0, nbytes = 0, hdr = 0;
    if (r < g.rows) {
      b = row_bits_in ? row_bits_in[r] : nbits[g.lres_spans + r];   // (k_span_bits / the other ranks' rows)
      nbytes = (b + 7) >> 3;
      hdr = g.use_blocks ? (nbytes <= 0x7fffu ? 2u : 4u) : 0u;
    }
    uint32_t tot;
    const uint32_t ex = block_scan_u32(nbytes + hdr, sm, &tot);
    if (r < g.rows) {
      const unsigned long long p = pos + ex + hdr;  // first payload byte
      bit0[g.lres_spans + r] = 8ull * p;
      nbits[g.lres_spans + r] = b;
      if (p + nbytes <= out_stride && r >= hr0 && r < hr1) {
        if (hdr == 2) {
          o[p - 2] = (uint8_t)(nbytes & 255); o[p - 1] = (uint8_t)(nbytes >> 8);
        } else if (hdr == 4) {
          const uint32_t lo = (nbytes & 0x7fffu) | 0x8000u, hi = nbytes >> 15;
          o[p - 4] = (uint8_t)(lo & 255); o[p - 3] = (uint8_t)(lo >> 8);
          o[p - 2] = (uint8_t)(hi & 255); o[p - 1] = (uint8_t)(hi >> 8);
        }
      }
    }
    pos += tot;
  }
  const unsigned long long total = pos;
  const uint32_t fres_bytes = (uint32_t)(total - fres_base);
  // The LRES span edges are OR-ed into a region pre-zeroed for lres_size + kTreeStride
  // + 64 bytes from kHeadLen & ~3 (launch_encode); a payload beyond it (more than
  // 8 bits per symbol on average) would be OR-ed onto stale bytes.
  const bool lres_fits = fres_rel || (unsigned long long)lres_bytes + 8ull <= (unsigned long long)g.lres_size + kTreeStride + 64ull;
  const bool fits = total <= out_stride && total < 0x7fffffffull && lres_fits;
  if (tid == 0) {
    if (!fits) atomicMax(&ws.status[f], 5);
    sizes[f] = (fits && ws.status[f] == 0) ? (uint32_t)total : 0u;
  }
  if (!fits || fres_rel) return;

  // Static container bytes with the data-dependent size fields patched in.
  for (int k = tid; k < kHeadLen; k += 256) {
    uint8_t b = sc.head[k];
    if (k >= 4 && k < 8) b = (uint8_t)(((uint32_t)total - 8u) >> (8 * (k - 4)));
    if (k >= kHeadLen - 4) b = (uint8_t)(lres_bytes >> (8 * (k - (kHeadLen - 4))));
    o[k] = b;
  }
  for (int k = tid; k < sc.mid_len; k += 256) {
    uint8_t b = sc.mid[k];
    if (k >= sc.mid_len - 4) b = (uint8_t)(fres_bytes >> (8 * (k - (sc.mid_len - 4))));
    o[kHeadLen + lres_bytes + k] = b;
  }
  const uint8_t *tl = ws.tree + ((size_t)f * 2 + 0) * kTreeStride;
  const uint8_t *tf = ws.tree + ((size_t)f * 2 + 1) * kTreeStride;
  for (int k = tid; k < (int)tree_l; k += 256) o[kHeadLen + k] = tl[k];
  for (int k = tid; k < (int)tree_f; k += 256) o[fres_base + k] = tf[k];
}

// ---------------------------------------------------------------------------
// k_emit: RLE tokenise + Huffman bit pack of one span straight into its final
// position.  Bits are assembled in an LDS staging buffer with ds atomics and
// flushed as whole dwords; the dwords at the two ends of a span are shared
// with neighbours: FRES rows are byte aligned, so edge dwords are written with
// byte stores (single owner per byte); LRES spans meet at arbitrary bit
// positions, so their edge dwords are OR-ed into the pre-zeroed LRES region.
//
// The kernel is latency-bound and lives on occupancy (PMC: 76 % VALU-busy at 6
// workgroups per CU; 4 workgroups cost +60 %), so the staging buffer is NOT
// sized for the worst case of an iteration (4096 symbols x 46 bits = 24 KiB) but
// for 4 KiB: a circular buffer indexed by the span-relative bit position.  An
// iteration whose bits exceed it (never seen on image data: it takes > 8 bits
// per SYMBOL) is emitted in several windows, each lane skipping the tokens
// outside the current window.
// ---------------------------------------------------------------------------
constexpr int kStageWords = 1024;                                   // power of two
constexpr uint32_t kWindowBits = (uint32_t)(kStageWords - 4) * 32u; // + carry word + 46-bit spill
constexpr int kPrivWords = 8;   // lane-private words per iteration (16 symbols: 256 bits cover all but 16+-bit codes)

// Wave-level inclusive scans (no barrier); lane 63 holds the wave total.
__device__ __forceinline__ ZR wave_scan_zr(ZR v) { return zr_unpack(wave_scan_zrp(zr_pack(v))); }
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) { return wave_scan_add(v); }

// Three barriers per 4096-symbol iteration: (A) after the waves publish their
// zero-run totals, (B) after they publish their bit totals, (C) after the bits
// have been OR-ed into the staging buffer.  The exchange slots are double
// buffered by iteration parity; every thread flushes AND re-zeroes its own words
// of the circular buffer, and the next ORs only come after the next iteration's
// barriers, so the flush needs no barrier of its own.
//
// ROWS > 1 (batches): the workgroup's ROWS wavefronts take one span EACH (ROWS
// consecutive spans), 1024 symbols per iteration, sharing only the tables: what is a
// barrier above is program order inside one wavefront, nobody waits for a slower
// wavefront's busiest lane, and a workgroup of eight waves shares one set of tables
// (46 KiB for eight spans against 8 x 24.5).
// WIDE (with ROWS == 1): 1024 lanes per span, 16384 symbols per iteration -- the FRES rows of
// a SINGLE frame: 512 workgroups of 256 lanes are a quarter of the chip's slots and 32
// iterations of three barriers per row; 1024 lanes fill the chip and take eight.
template <int ROWS, bool WIDE = false>
__global__ __launch_bounds__(ROWS > 1 ? 64 * ROWS : (WIDE ? 1024 : 256)) void k_emit_t(Geom g, EncWs ws, uint8_t *out, size_t out_stride,
                                                                       const uint32_t *sizes, int sp0, int sp1) {
  static_assert(!WIDE || ROWS == 1, "the wide form takes one span per workgroup");
  constexpr int NT = ROWS > 1 ? 64 : (WIDE ? 1024 : 256);   // lanes that work on one span
  constexpr int NG = ROWS > 1 ? ROWS : 1;          // spans per workgroup
  constexpr int NWAVE = NT / 64;                   // wavefronts per span
  // (staging words per span, a power of two: 256 per wavefront send too many iterations to the
  // window-by-window path -- 3.26 ms per 64 frames against 2.92 --, 1024 cost a workgroup per CU)
  constexpr int kStage = ROWS > 1 ? 512 : (WIDE ? 4 * kStageWords : kStageWords);
  constexpr uint32_t kWindow = (uint32_t)(kStage - 4) * 32u;                // + carry word + 46-bit spill
  constexpr int kIter = NT * 16;                                            // symbols per iteration
  __shared__ uint32_t stage_all[NG * kStage];
  __shared__ unsigned long long s_cl[kHistStride];  // code | length << 32
  // Merged token pairs: a zero run of r <= 6 zeros followed by the literal `sym`
  // (the common case by far) costs ONE lookup and ONE put: bits | length << 24,
  // 0 where the pair is longer than 24 bits (then the tokens go one by one).
  __shared__ uint32_t s_pair[kPairRuns + 1][256];   // row kPairRuns: zeros ("not merged")
  __shared__ uint32_t s_run[kRunTab + 1];   // run token of r zeros: bits | length << 24 (0: not representable; [kRunTab] = 0)
  __shared__ uint32_t s_priv_all[NG * (kPrivWords + 1) * NT];   // [word][lane]: the bits a lane assembled this iteration (+ one row that absorbs an overflowing lane's stores)
  __shared__ ZR sm_zr[2][NWAVE];
  __shared__ uint32_t sm_u[2][NWAVE];

  const int f = blockIdx.y;
  const int grp = ROWS > 1 ? (int)threadIdx.x / NT : 0;
  const int tid = ROWS > 1 ? (int)threadIdx.x % NT : (int)threadIdx.x;   // lane index inside the span's group
  const int sp = (int)blockIdx.x * NG + grp + sp0;
  const int lane = lane_id(), wave = ROWS > 1 ? 0 : wave_id();
  uint32_t *stage = stage_all + grp * kStage;
  uint32_t *s_priv = s_priv_all + grp * (kPrivWords + 1) * NT;
  // One group's barrier: the workgroup's for a span of four wavefronts, program order
  // (LDS operations of one wavefront execute in order) for a span of one.
  auto gsync = [&]() {
    if (ROWS > 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    else __syncthreads();
  };
  if (sizes[f] == 0) return;  // frame failed (status says why)
  const bool live = sp < sp1;
  const Span s = get_span(g, ws, live ? sp : sp0, f);
  const int nsp = g.lres_spans + g.rows;
  const unsigned long long B0 = ws.span_bit0[(size_t)f * nsp + (live ? sp : sp0)];
  const unsigned long long B1 = B0 + ws.span_bits[(size_t)f * nsp + (live ? sp : sp0)];
  uint8_t *o8 = out + (size_t)f * out_stride;
  uint32_t *o32 = reinterpret_cast<uint32_t *>(o8);

  // The tables (every span of a workgroup belongs to the same stream of the same frame).
  const int nthreads = NT * NG, t_all = (int)threadIdx.x;
  const size_t tab = ((size_t)f * 2 + (s.is_lres ? 0 : 1)) * kHistStride;
  for (int k = t_all; k < kHistStride; k += nthreads)
    s_cl[k] = (unsigned long long)(uint32_t)ws.codes[tab + k] | ((unsigned long long)ws.lens[tab + k] << 32);
  for (int k = t_all; k < NG * kStage; k += nthreads) stage_all[k] = 0;
  __syncthreads();
  for (int k = t_all; k < kPairRuns * 256; k += nthreads) {
    const int r = k >> 8, sym = k & 255;
    // run token of r zeros (huffman_enc.cpp:111-141): none, literal 0, 256, or 257 + (r - 3)
    const int rs = r == 1 ? 0 : r == 2 ? 256 : 257;
    const unsigned long long cr = r ? s_cl[rs] : 0ull, cs = s_cl[sym];
    const int lr = (int)(cr >> 32), eb = r >= 3 ? 2 : 0, ls = (int)(cs >> 32);
    const unsigned long long bits = (uint32_t)cr | ((unsigned long long)(r >= 3 ? r - 3 : 0) << lr) |
                                    ((unsigned long long)(uint32_t)cs << (lr + eb));
    const int n = lr + eb + ls;
    s_pair[r][sym] = (sym != 0 && ls > 0 && (r == 0 || lr > 0) && n <= 24) ? ((uint32_t)bits | ((uint32_t)n << 24)) : 0u;
  }
  for (int k = t_all; k < 256; k += nthreads) s_pair[kPairRuns][k] = 0;
  if (t_all == 0) s_run[kRunTab] = 0;
  for (int r = t_all; r < kRunTab; r += nthreads) {
    const int rs = r == 1 ? 0 : r == 2 ? 256 : r <= 6 ? 257 : r <= 22 ? 258 : 259;
    const int eb = r <= 2 ? 0 : r <= 6 ? 2 : r <= 22 ? 4 : 8;
    const int ev = r <= 2 ? 0 : r <= 6 ? r - 3 : r <= 22 ? r - 7 : r - 23;
    const unsigned long long cl = s_cl[rs];
    const int len = (int)(cl >> 32);
    s_run[r] = (r > 0 && len > 0 && len + eb <= 24)
                   ? ((uint32_t)cl | ((uint32_t)ev << len) | ((uint32_t)(len + eb) << 24)) : 0u;
  }
  int run_carry = live ? span_carry_in(g, ws, sp, f) : 0;  // zeros pending in front of this iteration
  // Positions are bits relative to the dword that holds the span's first bit;
  // word k of the span is global dword gw0 + k and staging slot k % kStage.
  const unsigned long long gw0 = B0 >> 5;
  uint32_t sbit = (uint32_t)(B0 & 31);  // position of the next token
  uint32_t fw = 0;                      // words already flushed
  __syncthreads();
  if (!live) return;   // (ROWS > 1: a wavefront beyond the last span; nothing below is a workgroup barrier then)

  auto store_word = [&](unsigned long long gw, uint32_t val) {
    const unsigned long long wb0 = gw * 32ull;
    if (wb0 >= B0 && wb0 + 32ull <= B1) {
      o32[gw] = val;
    } else if (s.is_lres) {
      if (val) atomicOr(&o32[gw], val);
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const unsigned long long bb = (gw * 4ull + b) * 8ull;
        if (bb >= B0 && bb < B1) o8[gw * 4ull + b] = (uint8_t)(val >> (8 * b));
      }
    }
  };

  uint32_t w[4], wn[4];
  load16(s.sym + tid * 16, s.len - tid * 16, w);
  int par = 0;
  LoopCount lci;
  for (int base = 0; base < s.len; base += kIter, par ^= 1) {
    HIMG_REGION_BEGIN("enc.iter");
    lci.step();
    const int off = base + tid * 16;
    const int nvalid = max(0, min(16, s.len - off));
    // Prefetch the next iteration's symbols.
    if (base + kIter < s.len) load16(s.sym + off + kIter, s.len - off - kIter, wn);
    const uint32_t mask = nonzero_mask16(w, nvalid);

    // (A) zero-run state in front of every lane.
    const ZR incl = wave_scan_zr(summarize16(mask, nvalid));
    const ZR ex = zr_unpack(wave_prev(kZrIdentity, zr_pack(incl)));
    ZR pre, tot;
    pre.tz = run_carry; pre.az = 0;
    if (NWAVE > 1) {
      if (lane == 63) sm_zr[par][wave] = incl;
      __syncthreads();
      tot = pre;
#pragma unroll
      for (int k = 0; k < NWAVE; ++k) {
        if (k < wave) pre = zr_combine(pre, sm_zr[par][k]);
        tot = zr_combine(tot, sm_zr[par][k]);
      }
    } else {
      tot = zr_combine(pre, zr_unpack((uint32_t)__builtin_amdgcn_readlane((int)zr_pack(incl), 63)));
    }
    const int run_in = zr_combine(pre, ex).tz;
    run_carry = tot.tz;
    const bool flush = s.last_of_block && nvalid > 0 && off + nvalid == s.len;

    // (B) ONE walk over the tokens: the lane assembles its bits from bit 0 into
    // lane-private LDS words (plain stores, transposed so that lanes never share a
    // bank) and thereby learns its bit count; after the offset scan the words are
    // shifted into the shared staging buffer whole.  A lane with more than
    // kPrivWords words (or an iteration beyond the staging window) falls back to
    // the two-pass, window-by-window emission below.
    uint32_t mybits, nwords;
    bool ovf = false;
    {
      uint32_t nw = 0, ab = 0;
      unsigned long long a = 0;
      // Branch free: the word in progress is stored after every token (the row past
      // the last absorbs a lane that overflows), a completed word just moves on.
      auto lput = [&](uint32_t v, int n) {  // n <= 32
        a |= (unsigned long long)v << ab;
        ab += n;
        s_priv[min(nw, (uint32_t)kPrivWords) * NT + tid] = (uint32_t)a;
        const uint32_t st = ab >> 5;   // 0 or 1
        nw += st;
        a >>= (st << 5);
        ab &= 31u;
      };
      walk16_emit(w, mask, nvalid, run_in, flush, &s_pair[0][0], s_run, s_cl, lput,
                  [&](int sym, int eb, int ev) {
                    const unsigned long long cl = s_cl[sym];
                    lput((uint32_t)cl, (int)(cl >> 32));
                    if (eb) lput((uint32_t)ev, eb);
                  });
      mybits = nw * 32u + ab;
      nwords = nw + (ab ? 1u : 0u);
      s_priv[min(nw, (uint32_t)kPrivWords) * NT + tid] = (uint32_t)a;   // the bits left of a word that completed
      ovf = nwords > (uint32_t)kPrivWords;
    }
    const uint32_t bincl = wave_scan_u32(mybits);
    uint32_t bpre = 0, iter_bits = 0;
    int any_ovf;
    if (NWAVE > 1) {
      if (lane == 63) sm_u[par][wave] = bincl;
      any_ovf = __syncthreads_or(ovf ? 1 : 0);
#pragma unroll
      for (int k = 0; k < NWAVE; ++k) {
        if (k < wave) bpre += sm_u[par][k];
        iter_bits += sm_u[par][k];
      }
    } else {
      any_ovf = __any(ovf ? 1 : 0);
      iter_bits = (uint32_t)__builtin_amdgcn_readlane((int)bincl, 63);
    }
    const uint32_t my_pos = sbit + bpre + bincl - mybits;  // where this lane's first token starts
    const uint32_t iter_end = sbit + iter_bits;

    if (!any_ovf && iter_bits <= kWindow) {
      // Fast path: shift the lane's words to its bit offset and OR them in.
      // One OR per staging word: the part of word j that spills over is OR-ed in
      // together with word j + 1.
      const uint32_t sh = my_pos & 31;
      uint32_t widx = (my_pos >> 5) & (kStage - 1), carry = 0;
      // (Not unrolled: fully unrolled to kPrivWords it is 103 instructions every iteration,
      // whatever the lanes hold -- and a third of the iterations, in the sparse
      // high-frequency rows, have one word per lane at most.)
      LoopCount lcs;
#pragma unroll 1
      for (uint32_t j = 0; j < nwords; ++j) {
        HIMG_REGION_BEGIN("enc.stage");
        lcs.step();
        const unsigned long long v = (unsigned long long)s_priv[j * NT + tid] << sh;
        atomicOr(&stage[widx], (uint32_t)v | carry);
        carry = (uint32_t)(v >> 32);
        widx = (widx + 1) & (kStage - 1);
        HIMG_REGION_END("enc.stage");
      }
      lcs.done(2);
      if (carry) atomicOr(&stage[widx], carry);
      gsync();   // (C)
      const uint32_t nw = iter_end >> 5;
      for (uint32_t k = fw + tid; k < nw; k += NT) {
        const uint32_t slot = k & (kStage - 1);
        store_word(gw0 + k, stage[slot]);
        stage[slot] = 0;
      }
      if (ROWS > 1) gsync();   // (one wavefront: the next iteration's ORs are not behind a barrier of their own)
      fw = nw;
      sbit = iter_end;
      w[0] = wn[0]; w[1] = wn[1]; w[2] = wn[2]; w[3] = wn[3];
      HIMG_REGION_END("enc.iter");   // (the fast path's end: the fallback below is not part of the hot iteration)
      continue;
    }

    // Fallback: append the tokens to the circular staging buffer token by token,
    // window by window.
    uint32_t wlo = sbit;
    for (;;) {
      const uint32_t whi = iter_end - wlo <= kWindow ? iter_end : wlo + kWindow;
      const bool whole = wlo == sbit && whi == iter_end;
      uint32_t widx = 0, accb = 0;
      unsigned long long acc = 0;
      auto put = [&](uint32_t v, int n) {  // n <= 32
        acc |= (unsigned long long)v << accb;
        accb += n;
        if (accb >= 32) {
          atomicOr(&stage[widx], (uint32_t)acc);
          widx = (widx + 1) & (kStage - 1);
          acc >>= 32;
          accb -= 32;
        }
      };
      if (whole) {
        widx = (my_pos >> 5) & (kStage - 1);
        accb = my_pos & 31;
        walk16_pairs(w, mask, nvalid, run_in, flush, &s_pair[0][0], s_run,
                     [&](uint32_t pair) { put(pair & 0xffffffu, (int)(pair >> 24)); },
                     [&](int sym, int eb, int ev) {
                       const unsigned long long cl = s_cl[sym];
                       put((uint32_t)cl, (int)(cl >> 32));
                       if (eb) put((uint32_t)ev, eb);
                     });
      } else {
        // Only the tokens that START inside [wlo, whi); they are contiguous.
        uint32_t q = my_pos;
        bool started = false;
        walk16(w, mask, nvalid, run_in, flush, [&](int sym, int eb, int ev) {
          const unsigned long long cl = s_cl[sym];
          const int len = (int)(cl >> 32);
          if (q >= wlo && q < whi) {
            if (!started) { widx = (q >> 5) & (kStage - 1); accb = q & 31; started = true; }
            put((uint32_t)cl, len);
            if (eb) put((uint32_t)ev, eb);
          }
          q += (uint32_t)(len + eb);
        });
      }
      if (accb && acc) atomicOr(&stage[widx], (uint32_t)acc);
      gsync();   // (C)

      // Flush the words that are complete below whi; each thread re-zeroes what it
      // flushed.  (Bits of a token that spills past whi stay staged.)
      const uint32_t nw = whi >> 5;
      for (uint32_t k = fw + tid; k < nw; k += NT) {
        const uint32_t slot = k & (kStage - 1);
        store_word(gw0 + k, stage[slot]);
        stage[slot] = 0;
      }
      fw = nw;
      wlo = whi;
      if (wlo >= iter_end) { if (ROWS > 1) gsync(); break; }
      gsync();   // the next window's ORs must not meet this flush
    }
    sbit = iter_end;
    w[0] = wn[0]; w[1] = wn[1]; w[2] = wn[2]; w[3] = wn[3];
  }
  lci.done(0);
  gsync();
  if (tid == 0 && (sbit & 31)) store_word(gw0 + (sbit >> 5), stage[(sbit >> 5) & (kStage - 1)]);
}

// ---------------------------------------------------------------------------
// k_emit_tok: bit packing of FRES block rows from their slots (k_tok), a wavefront per row,
// ROWS rows per workgroup sharing the tables.  A lane takes eight consecutive slots (one 16-byte
// load), so every lane is busy on every step and nothing is searched for: a literal slot costs two
// table reads -- the run token of the zeros in front of it (code + extra bits, <= 24 bits) and the
// literal's code -- joined into one word; the two slots of a dword join into one piece of <= 64
// bits; a wave scan of the lanes' bit counts places the pieces, which are OR-ed into the
// wavefront's circular staging buffer and leave as whole dwords (as k_emit_t).  Slots that do
// not fit (a code beyond 24 bits, more than 32 bits for run + literal) or an iteration beyond
// the staging window take the slot-by-slot path with the reference's tokens spelled out
// (huffman_enc.cpp:290-358).
// ---------------------------------------------------------------------------
constexpr uint32_t kTokBad = 0x80000000u;   // table entry: not representable on the fast path (length field 128)
template <int ROWS>
__global__ __launch_bounds__(64 * ROWS) void k_emit_tok(Geom g, EncWs ws, uint8_t *out, size_t out_stride,
                                                        const uint32_t *sizes, int r0, int r1) {
  constexpr int kStage = 512;                                   // staging dwords per wavefront (a power of two)
  constexpr uint32_t kWindow = (uint32_t)(kStage - 4) * 32u;    // + carry word + spill
  __shared__ uint32_t stage_all[ROWS * kStage];
  __shared__ unsigned long long s_cl[kHistStride];              // code | length << 32
  __shared__ uint32_t s_runp[256];    // zeros in front of a literal, 0..255: run token bits | length << 24
  __shared__ uint32_t s_lit[256];     // literal: code | length << 24; [0] = 0: the no-op slot
  __shared__ uint32_t s_prun[kRunTab + 1];   // a run on its own of r < 279 zeros
  const int f = blockIdx.y, lane = lane_id(), wave = wave_id();
  const int row = r0 + (int)blockIdx.x * ROWS + wave;
  if (sizes[f] == 0) return;   // frame failed (status says why)
  const bool live = row < r1;
  const int nthreads = 64 * ROWS, t_all = (int)threadIdx.x;
  const size_t tab = ((size_t)f * 2 + 1) * kHistStride;
  for (int k = t_all; k < kHistStride; k += nthreads)
    s_cl[k] = (unsigned long long)(uint32_t)ws.codes[tab + k] | ((unsigned long long)ws.lens[tab + k] << 32);
  for (int k = t_all; k < ROWS * kStage; k += nthreads) stage_all[k] = 0;
  __syncthreads();
  auto run_entry = [&](int r) -> uint32_t {   // the single token of r zeros, 0 < r < 279
    const int rs = r == 1 ? 0 : r == 2 ? 256 : r <= 6 ? 257 : r <= 22 ? 258 : 259;
    const int eb = r <= 2 ? 0 : r <= 6 ? 2 : r <= 22 ? 4 : 8;
    const int ev = r <= 2 ? 0 : r <= 6 ? r - 3 : r <= 22 ? r - 7 : r - 23;
    const unsigned long long cl = s_cl[rs];
    const int len = (int)(cl >> 32);
    return (len > 0 && len + eb <= 24) ? ((uint32_t)cl | ((uint32_t)ev << len) | ((uint32_t)(len + eb) << 24)) : kTokBad;
  };
  for (int k = t_all; k < 256; k += nthreads) {
    s_runp[k] = k ? run_entry(k) : 0u;
    const unsigned long long cl = s_cl[k];
    const int len = (int)(cl >> 32);
    s_lit[k] = k == 0 ? 0u : (len > 0 && len <= 24) ? ((uint32_t)cl | ((uint32_t)len << 24)) : kTokBad;
  }
  for (int k = t_all; k <= kRunTab; k += nthreads) s_prun[k] = (k > 0 && k < kRunTab) ? run_entry(k) : kTokBad;
  __syncthreads();
  if (!live) return;   // (nothing below is a workgroup barrier)

  uint32_t *stage = stage_all + wave * kStage;
  const int nsp = g.lres_spans + g.rows;
  const unsigned long long B0 = ws.span_bit0[(size_t)f * nsp + g.lres_spans + row];
  const unsigned long long B1 = B0 + ws.span_bits[(size_t)f * nsp + g.lres_spans + row];
  uint8_t *o8 = out + (size_t)f * out_stride;
  uint32_t *o32 = reinterpret_cast<uint32_t *>(o8);
  const unsigned long long gw0 = B0 >> 5;
  uint32_t sbit = (uint32_t)(B0 & 31);  // position of the next token, in bits from dword gw0
  uint32_t fw = 0;                      // words already flushed
  // FRES rows are byte aligned: the dwords at a row's two ends are shared with its neighbours
  // and written byte by byte (one owner per byte).
  auto store_word = [&](unsigned long long gw, uint32_t val) {
    const unsigned long long wb0 = gw * 32ull;
    if (wb0 >= B0 && wb0 + 32ull <= B1) {
      o32[gw] = val;
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const unsigned long long bb = (gw * 4ull + b) * 8ull;
        if (bb >= B0 && bb < B1) o8[gw * 4ull + b] = (uint8_t)(val >> (8 * b));
      }
    }
  };
  auto flush_to = [&](uint32_t end_bit) {   // complete words below end_bit leave; each lane re-zeroes what it flushed
    const uint32_t nw = end_bit >> 5;
    for (uint32_t k = fw + lane; k < nw; k += 64) {
      const uint32_t slot = k & (kStage - 1);
      store_word(gw0 + k, stage[slot]);
      stage[slot] = 0;
    }
    fw = nw;
  };
  // n <= 32 bits at bit position pos (relative to dword gw0).
  auto put32 = [&](uint32_t pos, uint32_t v, uint32_t n) {
    if (n == 0) return;
    const unsigned long long x = (unsigned long long)v << (pos & 31);
    const uint32_t wi = (pos >> 5) & (kStage - 1);
    atomicOr(&stage[wi], (uint32_t)x);
    if ((uint32_t)(x >> 32)) atomicOr(&stage[(wi + 1) & (kStage - 1)], (uint32_t)(x >> 32));
  };

  const size_t seg0 = ((size_t)f * g.rows + row) * ws.tok_nseg;
  LoopCount lci;
  for (int sg = 0; sg < ws.tok_nseg; ++sg) {
    const uint32_t nslots = min(ws.tok_cnt[seg0 + sg], (uint32_t)ws.tok_cap);   // (a count no producer wrote must not send the walk astray)
    const uint32_t nch = (nslots + 7) >> 3;
    const uint4 *src = reinterpret_cast<const uint4 *>(ws.tok + (seg0 + sg) * (size_t)ws.tok_cap);
    for (uint32_t base = 0; base < nch; base += 64) {
      HIMG_REGION_BEGIN("tok.iter");
      lci.step();
      uint4 q = {0u, 0u, 0u, 0u};
      if (base + lane < nch) q = src[base + lane];
      const uint32_t d[4] = {q.x, q.y, q.z, q.w};
      unsigned long long piece[4];
      uint32_t np[4], bad = g.row_tokens == 2 ? 1u : 0u, mybits = 0;   // (2: test knob, every iteration takes the spelled-out path)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const uint32_t s0 = d[p] & 0xffffu, s1 = d[p] >> 16;
        const uint32_t ra = s_runp[s0 >> 8], la = s_lit[s0 & 255u], rb = s_runp[s1 >> 8], lb = s_lit[s1 & 255u];
        const uint32_t na = (ra >> 24) + (la >> 24), nb = (rb >> 24) + (lb >> 24);
        const uint32_t ba = (ra & 0xffffffu) | ((la & 0xffffffu) << (ra >> 24));
        const uint32_t bb = (rb & 0xffffffu) | ((lb & 0xffffffu) << (rb >> 24));
        piece[p] = (unsigned long long)ba | ((unsigned long long)bb << na);
        np[p] = na + nb;
        uint32_t over = (na > 32u || nb > 32u) ? 1u : 0u;
        if (__builtin_expect(__any(s0 == kTokRunMark), 0)) {
          if (s0 == kTokRunMark) {   // a run on its own: s1 zeros
            const uint32_t e = s_prun[min(s1, (uint32_t)kRunTab)];
            const unsigned long long c260 = s_cl[260];
            const uint32_t l260 = (uint32_t)(c260 >> 32);
            if (s1 < (uint32_t)kRunTab) { piece[p] = e & 0xffffffu; np[p] = e >> 24; over = e >> 31; }
            else { piece[p] = (unsigned long long)(uint32_t)c260 | ((unsigned long long)(s1 - 279u) << l260); np[p] = l260 + 14u; over = l260 == 0u; }
          }
        }
        bad |= over;
        mybits += np[p];
      }
      const uint32_t bincl = wave_scan_add(mybits);
      const uint32_t iter_bits = (uint32_t)__builtin_amdgcn_readlane((int)bincl, 63);
      if (__builtin_expect(!__any(bad) && iter_bits <= kWindow, 1)) {
        uint32_t pos = sbit + bincl - mybits;
        auto place = [&](unsigned long long v, uint32_t n) {   // a piece of n <= 64 bits at `pos`
          const uint32_t sh = pos & 31u, wi = (pos >> 5) & (kStage - 1);
          const unsigned long long lo = v << sh;
          const uint32_t hi = (uint32_t)((v >> 33) >> (31u - sh));   // bits 64.. of the shifted piece
          atomicOr(&stage[wi], (uint32_t)lo);
          if (__any((uint32_t)(lo >> 32) != 0u)) atomicOr(&stage[(wi + 1) & (kStage - 1)], (uint32_t)(lo >> 32));
          if (__any(hi != 0u)) atomicOr(&stage[(wi + 2) & (kStage - 1)], hi);
          pos += n;
        };
        // Four slots are ~23 bits on average: where every lane's two quads fit 64 bits the pieces
        // are joined once more -- half the placements (and a third fewer LDS atomics).
        const uint32_t q0 = np[0] + np[1], q1 = np[2] + np[3];
        if (!__any(max(q0, q1) > 64u)) {
          place(piece[0] | (np[0] < 64u ? piece[1] << np[0] : 0ull), q0);
          place(piece[2] | (np[2] < 64u ? piece[3] << np[2] : 0ull), q1);
        } else {
#pragma unroll
          for (int p = 0; p < 4; ++p) place(piece[p], np[p]);
        }
        wave_lds_sync_e();
        sbit += iter_bits;
        flush_to(sbit);
        wave_lds_sync_e();
        HIMG_REGION_END("tok.iter");
        continue;
      }
      // The reference's tokens spelled out (any code length), sixteen lanes at a time: 128 slots
      // of at most 78 bits fit the staging window whatever they hold.
      auto slot_tokens = [&](int k, auto &&fn) {
        const uint32_t dw = d[k >> 1];
        const uint32_t s0 = dw & 0xffffu, s1 = dw >> 16;
        int run = 0, lit = 0;   // what the slot stands for: zeros, then a literal (0: none)
        if (s0 == kTokRunMark) { if (!(k & 1)) run = (int)s1; }
        else { const uint32_t sl = (k & 1) ? s1 : s0; lit = (int)(sl & 255u); run = lit ? (int)(sl >> 8) : 0; }
        if (run) emit_run(run, fn);
        if (lit) fn(lit, 0, 0);
      };
      uint32_t nb = 0;
#pragma unroll 1
      for (int k = 0; k < 8; ++k) slot_tokens(k, [&](int sym, int eb, int) { nb += (uint32_t)(s_cl[sym] >> 32) + (uint32_t)eb; });
      const uint32_t incl = wave_scan_add(nb);
#pragma unroll 1
      for (int grp = 0; grp < 4; ++grp) {
        if ((lane >> 4) == grp) {
          uint32_t pos = sbit + incl - nb;
#pragma unroll 1
          for (int k = 0; k < 8; ++k)
            slot_tokens(k, [&](int sym, int eb, int ev) {
              const unsigned long long cl = s_cl[sym];
              const uint32_t len = (uint32_t)(cl >> 32);
              put32(pos, (uint32_t)cl, len);
              pos += len;
              put32(pos, (uint32_t)ev, (uint32_t)eb);
              pos += (uint32_t)eb;
            });
        }
        wave_lds_sync_e();
        flush_to(sbit + (uint32_t)__builtin_amdgcn_readlane((int)incl, 16 * grp + 15));
        wave_lds_sync_e();
      }
      sbit += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
  }
  lci.done(5);
  if (lane == 0 && (sbit & 31)) store_word(gw0 + (sbit >> 5), stage[(sbit >> 5) & (kStage - 1)]);
}

// ---------------------------------------------------------------------------
// k_padfix (trap T1): the reference packs every block row into ONE scratch
// buffer that is never cleared, and WriteBits only touches the bits it writes
// (huffman_enc.cpp:31-50,288,355-358).  The unused high bits of a row's last
// byte therefore keep what the most recent earlier row left at that byte index.
// One thread per row resolves its pad bits by walking back over earlier rows.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_padfix(Geom g, EncWs ws, uint8_t *out, size_t out_stride,
                                                const uint32_t *sizes) {
  // One WAVEFRONT per row: the walk back over earlier rows looks at 64 of them per
  // step (coalesced read of their bit counts, one ballot) instead of one per
  // dependent load -- a row longer than all before it used to scan them one by one.
  const int r = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), f = blockIdx.y;
  const int lane = threadIdx.x & 63;
  if (r >= g.rows || sizes[f] == 0) return;
  const int nsp = g.lres_spans + g.rows;
  const uint64_t *bit0 = ws.span_bit0 + (size_t)f * nsp + g.lres_spans;
  const uint32_t *nbits = ws.span_bits + (size_t)f * nsp + g.lres_spans;
  uint8_t *o = out + (size_t)f * out_stride;
  const uint32_t bits = nbits[r];
  const uint32_t valid = bits & 7;
  if (valid == 0) return;  // last byte full: nothing stale shows through
  const uint32_t idx = ((bits + 7) >> 3) - 1;  // index of the last byte inside the row
  uint32_t need = 0xffu & ~((1u << valid) - 1u);
  uint32_t acc = 0;
  int q_hi = r - 1;   // the most recent row not yet looked at
  while (need && q_hi >= 0) {
    const int q = q_hi - lane;
    const uint32_t qb = q >= 0 ? nbits[q] : 0u;
    const unsigned long long hit = __ballot(q >= 0 && ((qb + 7) >> 3) > idx);   // rows that wrote scratch[idx]
    if (!hit) { q_hi -= 64; continue; }
    const int l = __ffsll((long long)hit) - 1;   // the most recent of them
    const int qs = q_hi - l;
    const uint32_t qbs = (uint32_t)__shfl((int)qb, l);
    const uint32_t qn = (qbs + 7) >> 3;
    const uint32_t byte = o[(bit0[qs] >> 3) + idx];
    if (qn - 1 == idx) {
      // scratch[idx] is row qs's own last byte: its valid bits are its own, its pad
      // bits are older still.
      const uint32_t qv = qbs & 7;
      const uint32_t vmask = qv ? ((1u << qv) - 1u) : 0xffu;
      acc |= byte & vmask & need;
      need &= ~vmask;
    } else {
      acc |= byte & need;
      need = 0;
    }
    q_hi = qs - 1;
  }
  if (lane == 0 && acc) o[(bit0[r] >> 3) + idx] |= (uint8_t)acc;
}

// ---------------------------------------------------------------------------
// Row-sharded (multi-GPU) helpers.
// ---------------------------------------------------------------------------

// Copy the gathered FRES rows (headers + payloads, laid out relative to the first
// row header) behind the FRES tree of the final stream.  The destination offset
// is data dependent (LRES size) and lives on the device.
__global__ __launch_bounds__(256) void k_place_fres(Geom g, EncWs ws, const uint8_t *rel,
                                                    size_t rel_bytes, uint8_t *out,
                                                    const uint32_t *sizes) {
  if (sizes[0] == 0) return;
  const uint32_t b0 = ws.span_bits[g.lres_spans];  // row 0
  const uint32_t n0 = (b0 + 7) >> 3;
  const uint32_t hdr0 = g.use_blocks ? (n0 <= 0x7fffu ? 2u : 4u) : 0u;
  uint8_t *dst = out + (ws.span_bit0[g.lres_spans] >> 3) - hdr0;
  // 16 bytes per thread: aligned loads (rel is 16-byte aligned), stores at whatever
  // alignment the LRES size left the destination with (the hardware takes unaligned
  // global stores); the last few bytes one by one.
  const size_t n16 = rel_bytes >> 4;
  const uint4 *src16 = reinterpret_cast<const uint4 *>(rel);
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (size_t)gridDim.x * blockDim.x) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint4 q = src16[k];
    const u32x4 v = {q.x, q.y, q.z, q.w};
    uint8_t *d = dst + 16 * k;
    asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(d), "v"(v) : "memory");
  }
  for (size_t k = (n16 << 4) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < rel_bytes;
       k += (size_t)gridDim.x * blockDim.x)
    dst[k] = rel[k];
}

// ---------------------------------------------------------------------------
// Host-side launch sequence.
// ---------------------------------------------------------------------------
#define HIMG_LAUNCH(name, grid, block, ...)                    \
  do {                                                         \
    prof_begin(prof, #name, stream);                           \
    hipLaunchKernelGGL(name, grid, block, 0, stream, __VA_ARGS__); \
    prof_end(prof, stream);                                    \
  } while (0)

// Occupancy sweeps (BASELINE config 3, tools/occupancy_sweep.py): HIMG_LDS_PAD=<bytes>
// adds that much unused dynamic LDS to every workgroup of the three wide encode
// kernels, which lowers the number of workgroups a CU can hold.  0 in production.
static size_t lds_pad() {
  static const size_t v = [] { const char *e = getenv("HIMG_LDS_PAD"); const long x = e ? atol(e) : 0; return (size_t)(x > 0 && x <= 150000 ? x : 0); }();
  return v;
}
#define HIMG_LAUNCH_PAD(name, grid, block, ...)                \
  do {                                                         \
    prof_begin(prof, #name, stream);                           \
    hipLaunchKernelGGL(name, grid, block, lds_pad(), stream, __VA_ARGS__); \
    prof_end(prof, stream);                                    \
  } while (0)

// Zero `n` dwords at the head of every row of a [batch][stride] dword array, and two
// small arrays (the histograms and the status words) on the side: one launch instead
// of three memsets in front of every encode.
__global__ __launch_bounds__(256) void k_zero_rows(uint32_t *base, size_t stride, uint32_t n, uint32_t *a,
                                                   uint32_t na, uint32_t *b, uint32_t nb) {
  uint32_t *row = base + (size_t)blockIdx.y * stride;
  const uint32_t k0 = blockIdx.x * (256u * 8u) + threadIdx.x;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t k = k0 + (uint32_t)j * 256u;
    if (k < n) row[k] = 0;
  }
  if (blockIdx.y == 0) {
    const uint32_t stride_t = gridDim.x * 256u, t0 = blockIdx.x * 256u + threadIdx.x;
    for (uint32_t k = t0; k < na; k += stride_t) a[k] = 0;
    for (uint32_t k = t0; k < nb; k += stride_t) b[k] = 0;
  }
}

// The round-2 pixel stage (k_pix_fwd) serves full tiles of packed RGBA8.
static bool use_pix_path(const Geom &g) {
  return g.W % 8 == 0 && g.H % 8 == 0 && g.stride == 4 && g.C == 4;
}

static void launch_pix(const Geom &g, const EncWs &ws, const uint8_t *d_frames, const ShiftTables &st,
                       const uint8_t *d_fmap_lut, int r0, int n, int batch, hipStream_t stream,
                       Profiler *prof) {
  const unsigned gxt = (unsigned)((g.cols + kPixThreads - 1) / kPixThreads);
  const dim3 grid(gxt, n, batch), block(kPixThreads);
  const PixQuant pq = make_pix_quant(st);
#define HIMG_PIX(Y, COLS, FULL)                                                                  \
  HIMG_LAUNCH_PAD((k_pix_fwd<Y, COLS, FULL>), grid, block, g, d_frames, ws.low, ws.plane_stride, ws.fres_sym, \
              ws.fres_stride, d_fmap_lut, pq, r0)
  const bool full = g.cols % 64 == 0;
  static const bool no_cols = getenv("HIMG_NO_COLS") != nullptr;   // (A/B knob, see launch_decode)
  // Compile-time strides (immediate store offsets) for the widths of the BASELINE configurations:
  // 4096 (configs 2, 5), 2048, and 1920 (config 3: 240 tiles, the last wavefront of a row ragged).
  if (g.ycbcr) {
    if (g.cols == 512) HIMG_PIX(true, 512, true);
    else if (g.cols == 256 && !no_cols) HIMG_PIX(true, 256, true);
    else if (g.cols == 240 && !no_cols) HIMG_PIX(true, 240, false);
    else if (full) HIMG_PIX(true, 0, true);
    else HIMG_PIX(true, 0, false);
  }
  else { if (full) HIMG_PIX(false, 0, true); else HIMG_PIX(false, 0, false); }
#undef HIMG_PIX
}

// k_front (box averages + low-res plane + pixel stage in one pass over the pixels) serves batches of
// full RGBA8 frames whose rows are at most 512 tiles wide (HIMG_OPT_FRONT forces the three-kernel
// front / this one).
static bool use_front(const Geom &g, int batch) {
  if (!use_pix_path(g) || g.cols > 512 || g.front == 0) return false;
  if (g.front > 0) return true;
  // By launch size: batches; and rows of at least four wavefronts -- a workgroup holds 16 KiB of LDS per
  // wavefront plus the 8 KiB table, so narrower rows leave a CU with six wavefronts instead of eight
  // (256 x 1024^2: 97 against 109 Gpx/s with the three kernels).
  return g.cols > 192 && (long long)g.rows * batch >= 8192;
}
static void launch_front(const Geom &g, const EncWs &ws, const uint8_t *d_frames, const ShiftTables &st,
                         const uint8_t *d_fmap_lut, int batch, hipStream_t stream, Profiler *prof) {
  const int wpr = (g.cols + 63) / 64, nt = 64 * wpr;
  // ~64 block rows per workgroup (a chunk re-reads two tile rows above it), more chunks when the batch
  // alone does not give every CU two rounds of workgroups; never fewer than 16 rows.
  static const int chunk_env = [] { const char *e = getenv("HIMG_FRONT_CHUNK"); return e ? atoi(e) : 0; }();   // (A/B knob)
  const int target = chunk_env > 0 ? chunk_env : 64;
  int nchunks = g.rows >= target + target / 2 ? (g.rows + target / 2) / target : 1;
  const int slots = 256 * (8 / wpr > 0 ? 8 / wpr : 1);
  nchunks = max(nchunks, min((2 * slots + batch - 1) / batch, max(1, g.rows / 16)));
  const int chunk = (g.rows + nchunks - 1) / nchunks;
  const size_t lds = (size_t)nt * 256 + kPixLut + (size_t)kFrontExch * nt * 4;
  const dim3 grid((unsigned)((g.rows + chunk - 1) / chunk), (unsigned)batch), block((unsigned)nt);
  FrontArgs fa;
  fa.g = g; fa.frames = d_frames; fa.avg = ws.avg; fa.low = ws.low; fa.plane_stride = ws.plane_stride;
  fa.fres_sym = ws.fres_sym; fa.fres_stride = ws.fres_stride; fa.fmap_lut = d_fmap_lut; fa.chunk_rows = chunk;
  fa.pq = make_pix_quant(st);
  static_assert(sizeof(((PixQuant *)0)->rr) == 2 * 64 * 4 && offsetof(PixQuant, kk) == 512 && offsetof(PixQuant, ss) == 1024,
                "k_front indexes the quantiser's words as [table][luma / chroma][coefficient]");
#define HIMG_FRONT(Y, COLS)                                                                               \
  do {                                                                                                    \
    static bool attr_done = false;                                                                        \
    if (!attr_done) {                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_front<Y, COLS>),                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                  \
      attr_done = true;                                                                                   \
    }                                                                                                     \
    prof_begin(prof, "k_front", stream);                                                                  \
    hipLaunchKernelGGL((k_front<Y, COLS>), grid, block, lds, stream, fa);                                 \
    prof_end(prof, stream);                                                                               \
  } while (0)
  static const bool no_cols = getenv("HIMG_NO_COLS") != nullptr;
  if (g.ycbcr) {
    if (g.cols == 512) HIMG_FRONT(true, 512);
    else if (g.cols == 256 && !no_cols) HIMG_FRONT(true, 256);
    else if (g.cols == 240 && !no_cols) HIMG_FRONT(true, 240);
    else HIMG_FRONT(true, 0);
  } else {
    HIMG_FRONT(false, 0);
  }
#undef HIMG_FRONT
}

constexpr long long kWideRows = 1024;   // up to this many FRES rows per call: 1024 lanes per row (k_tok_hist, k_emit)

// k_tok_hist over the FRES rows [r0, r1) of every frame: a workgroup of 256 lanes per row,
// or of 1024 when there are too few rows to fill the GPU that way (a single frame).
static void launch_tok_hist_rows(const Geom &g, const EncWs &ws, int r0, int r1, int batch, hipStream_t stream,
                                 Profiler *prof) {
  if (r1 <= r0) return;
  prof_begin(prof, "k_tok_hist", stream);
  if ((long long)(r1 - r0) * batch <= kWideRows && g.row_block >= 32768)
    hipLaunchKernelGGL(k_tok_hist<1024>, dim3(r1 - r0, batch), dim3(1024), 0, stream, g, ws, g.lres_spans + r0);
  else
    hipLaunchKernelGGL(k_tok_hist<256>, dim3(r1 - r0, batch), dim3(256), lds_pad(), stream, g, ws, g.lres_spans + r0);
  prof_end(prof, stream);
}

// k_emit over the spans [sp0, sp1) of every frame: LRES spans (they meet at bit positions:
// one workgroup each), then the FRES rows -- a wavefront each, eight to a workgroup, when
// there are enough of them to fill the GPU that way (batches); a single frame keeps one
// workgroup per row (HIMG_EMIT_ROWS=0 / 1 forces either).
constexpr int kEmitRows = 8;   // (2 / 4 / 6 rows per workgroup: 3.35 / 2.99 / 3.49 ms per 64 frames, 8: 2.92)
// lres_stream: where the LRES spans' launch goes (the caller's stream, or a side stream the
// caller has forked and will join: the two launches touch disjoint bytes).
static void launch_emit(const Geom &g, const EncWs &ws, uint8_t *d_out, size_t out_stride, const uint32_t *d_sizes,
                        int sp0, int sp1, int batch, hipStream_t stream, Profiler *prof,
                        hipStream_t lres_stream = nullptr) {
  const int rows_env = g.emit_rows;   // HIMG_OPT_EMIT_ROWS (-1: by the number of rows)
  const int l1 = sp0 < g.lres_spans ? (sp1 < g.lres_spans ? sp1 : g.lres_spans) : sp0;   // [sp0, l1): LRES spans
  if (l1 > sp0) {
    hipStream_t ls = lres_stream ? lres_stream : stream;
    prof_begin(prof, "k_emit", ls);
    hipLaunchKernelGGL(k_emit_t<1>, dim3(l1 - sp0, batch), dim3(256), lds_pad(), ls, g, ws, d_out, out_stride,
                       d_sizes, sp0, l1);
    prof_end(prof, ls);
  }
  if (sp1 > l1) {
    const long long rows = (long long)(sp1 - l1) * batch;
    const bool by_wave = rows_env >= 0 ? rows_env != 0 : rows >= 8192;
    prof_begin(prof, "k_emit", stream);
    if (by_wave)
      hipLaunchKernelGGL(k_emit_t<kEmitRows>, dim3((sp1 - l1 + kEmitRows - 1) / kEmitRows, batch), dim3(64 * kEmitRows),
                         lds_pad(), stream, g, ws, d_out, out_stride, d_sizes, l1, sp1);
    else if (rows <= kWideRows && g.row_block >= 32768)   // a single frame (or two): 1024 lanes per row
      hipLaunchKernelGGL((k_emit_t<1, true>), dim3(sp1 - l1, batch), dim3(1024), lds_pad(), stream, g, ws, d_out,
                         out_stride, d_sizes, l1, sp1);
    else
      hipLaunchKernelGGL(k_emit_t<1>, dim3(sp1 - l1, batch), dim3(256), lds_pad(), stream, g, ws, d_out, out_stride,
                         d_sizes, l1, sp1);
    prof_end(prof, stream);
  }
}

// ---- FRES rows through the token stream (k_tok + k_emit_tok) ----
bool enc_uses_row_tokens(const Geom &g, int batch) {
  if (g.row_tokens == 0 || g.row_block % 64 != 0) return false;
  // (a single frame keeps the latency-tuned kernels: its 512 rows are a quarter of the chip's slots)
  return g.row_tokens > 0 || (long long)g.rows * batch >= 8192;
}
// Symbols per token segment (a multiple of the tokeniser's iteration): eight segments per row, four
// for rows below 64 Ki symbols -- enough for the row's four wavefronts to balance the dense
// low-frequency segments against the sparse ones, few enough that the look back in front of a
// segment and the ragged last step of k_emit_tok per segment stay small (measured, r06_experiments.md:
// sixteen segments cost 3 % at 4096 pixels, 10 % at 1080p and 20 % at 1024 pixels).
int enc_tok_seg(const Geom &g) {
  static const int seg_env = [] { const char *e = getenv("HIMG_TOK_SEG"); return e ? atoi(e) : 0; }();   // (A/B knob)
  const int nseg = g.row_block >= 65536 ? 8 : 4;
  const int per = (g.row_block + nseg - 1) / nseg;
  int seg = (per + kTokIter - 1) / kTokIter * kTokIter;
  if (seg_env > 0) seg = (seg_env + kTokIter - 1) / kTokIter * kTokIter;
  return seg;
}

static void launch_tok_rows(const Geom &g, const EncWs &ws, int r0, int r1, int batch, hipStream_t stream, Profiler *prof) {
  if (r1 <= r0) return;
  prof_begin(prof, "k_tok", stream);
  hipLaunchKernelGGL(k_tok, dim3(r1 - r0, batch), dim3(kTokThreads), lds_pad(), stream, g, ws, r0);
  prof_end(prof, stream);
}

static void launch_emit_tok(const Geom &g, const EncWs &ws, uint8_t *d_out, size_t out_stride, const uint32_t *d_sizes,
                            int r0, int r1, int batch, hipStream_t stream, Profiler *prof) {
  if (r1 <= r0) return;
  prof_begin(prof, "k_emit_tok", stream);
  static const int rows_env = [] { const char *e = getenv("HIMG_EMIT_TOK_ROWS"); return e ? atoi(e) : 0; }();   // (A/B knob)
  if (rows_env == 4)
    hipLaunchKernelGGL(k_emit_tok<4>, dim3((r1 - r0 + 3) / 4, batch), dim3(256), 0, stream, g, ws, d_out, out_stride, d_sizes, r0, r1);
  else
    hipLaunchKernelGGL(k_emit_tok<kEmitRows>, dim3((r1 - r0 + kEmitRows - 1) / kEmitRows, batch), dim3(64 * kEmitRows), 0,
                       stream, g, ws, d_out, out_stride, d_sizes, r0, r1);
  prof_end(prof, stream);
}

void launch_tok_expand(const Geom &g, const EncWs &ws, int frame, uint8_t *dst, hipStream_t stream) {
  (void)hipMemsetAsync(dst, 0, (size_t)g.fres_size, stream);
  hipLaunchKernelGGL(k_tok_expand, dim3((g.rows + 63) / 64), dim3(64), 0, stream, g, ws, frame, dst);
}

int loop_counts_read_enc(unsigned long long *out) { return loop_counts_read(out); }

void launch_encode(const Geom &g, const EncWs &ws, int batch, const uint8_t *d_frames,
                   uint8_t *d_out, size_t out_stride, uint32_t *d_sizes,
                   const StaticChunks &sc, const ShiftTables &st, const LresTables &lt,
                   const uint8_t *d_fmap_lut, hipStream_t stream, Profiler *prof,
                   hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join) {
  const int nsp = g.lres_spans + g.rows;
  const dim3 b256(256);
  const unsigned gx = (unsigned)((g.cols + 255) / 256);

  prof_begin(prof, "memset", stream);
  // Histograms, status words and the LRES payload region -- pre-zeroed because span
  // edges are OR-ed in (k_emit).
  {
    const size_t start = (size_t)(kHeadLen & ~3);
    size_t width = (size_t)g.lres_size + kTreeStride + 64;
    if (start + width > out_stride) width = out_stride - start;
    // (hipMemset2DAsync takes 90 us for these 67 MB of a 64-frame batch; plain dword
    // stores take a fifth of that.)
    const uint32_t nd = (uint32_t)((width + 3) / 4);   // start is dword aligned, the rows lie out_stride apart
    hipLaunchKernelGGL(k_zero_rows, dim3((nd + 256 * 8 - 1) / (256 * 8), batch), b256, 0, stream,
                       reinterpret_cast<uint32_t *>(d_out + start), out_stride / 4, nd,
                       reinterpret_cast<uint32_t *>(ws.hist), (uint32_t)(batch * 2 * kHistStride),
                       reinterpret_cast<uint32_t *>(ws.status), (uint32_t)batch);
  }
  prof_end(prof, stream);

  // Batches of full RGBA8 frames: one pass over the pixels (k_front) gives the box averages, the
  // low-res plane and the symbols; the LRES branch then forks behind it and runs beside the tokeniser.
  const bool front = use_front(g, batch);
  if (front) {
    launch_front(g, ws, d_frames, st, d_fmap_lut, batch, stream, prof);
  } else {
    HIMG_LAUNCH(k_lowres_avg, dim3(gx, g.rows, batch), b256, g, d_frames, ws.avg, ws.plane_stride, 0);
    if ((g.cols & 3) == 0 && (ws.plane_stride & 3) == 0)
      HIMG_LAUNCH(k_lowres_blend<true>, dim3((g.cols / 4 + 255) / 256, (g.rows + kBlendRows - 1) / kBlendRows, batch * g.C), b256,
                  g, ws.avg, ws.low, ws.plane_stride, 0, g.rows);
    else
      HIMG_LAUNCH(k_lowres_blend<false>, dim3(gx, (g.rows + kBlendRows - 1) / kBlendRows, batch * g.C), b256, g, ws.avg, ws.low,
                  ws.plane_stride, 0, g.rows);
  }
  // The LRES branch (predictor selection + delta chain, zero-run summaries, token
  // histogram of the LRES spans: 1/64 of the data, latency-bound kernels) forks to the
  // side stream and runs beside the pixel stage and the FRES histogram; the two
  // branches join in front of the tree build.  (side == nullptr: in line.)
  hipStream_t ls = side ? side : stream;
  if (side) {
    (void)hipEventRecord(ev_fork, stream);
    (void)hipStreamWaitEvent(side, ev_fork, 0);
  }
  {
    hipStream_t stream_saved = stream;
    stream = ls;
    HIMG_LAUNCH(k_lres_predict, dim3((g.mcols + 3) / 4, g.mrows, batch * g.C), dim3(64), g, ws.low,
                ws.plane_stride, ws.lres_sym, ws.lres_stride, lt);
    HIMG_LAUNCH(k_lres_summary, dim3(g.lres_spans, batch), b256, g, ws);
    HIMG_LAUNCH(k_tok_hist<256>, dim3(g.lres_spans, batch), b256, g, ws, 0);
    stream = stream_saved;
  }
  if (side) (void)hipEventRecord(ev_join, side);
  const unsigned gxt = (unsigned)((g.cols + kTileThreads - 1) / kTileThreads);
  const bool pix = use_pix_path(g);
  const bool row_tok = ws.tok != nullptr && enc_uses_row_tokens(g, batch);
  if (front) {
    // (the symbols are there already)
  } else if (pix) {
    launch_pix(g, ws, d_frames, st, d_fmap_lut, 0, g.rows, batch, stream, prof);
  } else {
    HIMG_LAUNCH((k_tile_fwd<false, 0>), dim3(gxt, g.rows, batch), dim3(kTileThreads), g, d_frames,
                ws.low, ws.plane_stride, ws.fres_sym, ws.fres_stride, d_fmap_lut, st, 0);
  }
  if (row_tok) launch_tok_rows(g, ws, 0, g.rows, batch, stream, prof);   // FRES rows: slots + histograms
  else launch_tok_hist_rows(g, ws, 0, g.rows, batch, stream, prof);
  if (side) (void)hipStreamWaitEvent(stream, ev_join, 0);
  HIMG_LAUNCH(k_tree, dim3(2, batch), dim3(kTreeThreads), ws, 0);
  HIMG_LAUNCH(k_span_bits, dim3((nsp + 3) / 4, batch), b256, g, ws, 0, nsp, (uint32_t *)nullptr);
  HIMG_LAUNCH(k_sizes, dim3(batch), b256, g, ws, sc, d_out, out_stride, d_sizes,
              (const uint32_t *)nullptr, 0, 0, g.rows);
  // The LRES spans' bit packing (1/64 of the symbols, 20 us of a single frame's 130) beside
  // the FRES rows': forked to the side stream again, joined behind the pad-bit fix-up.
  // (A single frame only: in a batch the LRES spans are thousands of workgroups of their own
  // and the two launches side by side just slow each other down -- encode 141.7 -> 138.9 Gpx/s.)
  hipStream_t lres_side = (side && (long long)g.rows * batch <= kWideRows) ? side : nullptr;
  if (lres_side) {
    (void)hipEventRecord(ev_fork, stream);
    (void)hipStreamWaitEvent(lres_side, ev_fork, 0);
  }
  if (row_tok) {
    launch_emit(g, ws, d_out, out_stride, d_sizes, 0, g.lres_spans, batch, stream, prof, lres_side);
    launch_emit_tok(g, ws, d_out, out_stride, d_sizes, 0, g.rows, batch, stream, prof);
  } else {
    launch_emit(g, ws, d_out, out_stride, d_sizes, 0, nsp, batch, stream, prof, lres_side);
  }
  if (lres_side) (void)hipEventRecord(ev_join, lres_side);
  HIMG_LAUNCH(k_padfix, dim3((g.rows + 3) / 4, batch), b256, g, ws, d_out, out_stride,
              d_sizes);
  if (lres_side) (void)hipStreamWaitEvent(stream, ev_join, 0);
}

// ---- row-sharded encode of ONE frame (see himg_hip.h, "row-sharded encode") ----

static void launch_tile_rows(const Geom &g, const EncWs &ws, const uint8_t *d_frame_base,
                             const ShiftTables &st, const uint8_t *d_fmap_lut, int r0, int n,
                             hipStream_t stream, Profiler *prof) {
  const unsigned gxt = (unsigned)((g.cols + kTileThreads - 1) / kTileThreads);
  if (use_pix_path(g)) {
    launch_pix(g, ws, d_frame_base, st, d_fmap_lut, r0, n, 1, stream, prof);
  } else {
    HIMG_LAUNCH((k_tile_fwd<false, 0>), dim3(gxt, n, 1), dim3(kTileThreads), g, d_frame_base,
                ws.low, ws.plane_stride, ws.fres_sym, ws.fres_stride, d_fmap_lut, st, r0);
  }
}

void launch_shard_stats(const Geom &g, const EncWs &ws, const uint8_t *d_frame_base,
                        const ShiftTables &st, const uint8_t *d_fmap_lut, int r0, int r1,
                        hipStream_t stream, Profiler *prof) {
  const dim3 b256(256);
  const unsigned gx = (unsigned)((g.cols + 255) / 256);
  (void)hipMemsetAsync(ws.hist, 0, 2 * kHistStride * sizeof(uint32_t), stream);
  (void)hipMemsetAsync(ws.status, 0, sizeof(int32_t), stream);
  // low[v] needs avg[v-1..v]; the tiles of row v need low[v..v+1].
  const int a0 = r0 > 0 ? r0 - 1 : 0, a1 = r1 < g.rows ? r1 + 1 : g.rows;
  const int l1 = r1 < g.rows ? r1 + 1 : g.rows;
  HIMG_LAUNCH(k_lowres_avg, dim3(gx, a1 - a0, 1), b256, g, d_frame_base, ws.avg, ws.plane_stride, a0);
  HIMG_LAUNCH(k_lowres_blend<false>, dim3(gx, (l1 - r0 + kBlendRows - 1) / kBlendRows, g.C), b256, g, ws.avg, ws.low,
              ws.plane_stride, r0, l1);
  launch_tile_rows(g, ws, d_frame_base, st, d_fmap_lut, r0, r1 - r0, stream, prof);
  // (rows through the token stream when the workspace holds one: himg_hip_shard_stats decides by the range's size)
  if (ws.tok) launch_tok_rows(g, ws, r0, r1, 1, stream, prof);
  else launch_tok_hist_rows(g, ws, r0, r1, 1, stream, prof);
}

void launch_shard_row_bits(const Geom &g, const EncWs &ws, int r0, int r1, uint32_t *d_bits_out,
                           hipStream_t stream, Profiler *prof) {
  HIMG_LAUNCH(k_tree, dim3(1, 1), dim3(kTreeThreads), ws, 1);
  if (r1 > r0)
    HIMG_LAUNCH(k_span_bits, dim3((r1 - r0 + 3) / 4, 1), dim3(256), g, ws, g.lres_spans + r0, g.lres_spans + r1, d_bits_out);
}

void launch_shard_emit(const Geom &g, const EncWs &ws, const StaticChunks &sc,
                       const uint32_t *d_all_row_bits, uint8_t *d_rel, size_t rel_cap,
                       uint32_t *d_rel_size, int r0, int r1, hipStream_t stream, Profiler *prof) {
  HIMG_LAUNCH(k_sizes, dim3(1), dim3(256), g, ws, sc, d_rel, rel_cap, d_rel_size, d_all_row_bits, 1, r0, r1);
  if (r1 > r0) {
    if (ws.tok) launch_emit_tok(g, ws, d_rel, rel_cap, d_rel_size, r0, r1, 1, stream, prof);
    else launch_emit(g, ws, d_rel, rel_cap, d_rel_size, g.lres_spans + r0, g.lres_spans + r1, 1, stream, prof);
  }
}

void launch_shard_assemble(const Geom &g, const EncWs &ws, const StaticChunks &sc,
                           const LresTables &lt, const uint32_t *d_all_row_bits,
                           const uint8_t *d_rel, size_t rel_bytes, uint8_t *d_out, size_t out_cap,
                           uint32_t *d_size, hipStream_t stream, Profiler *prof) {
  const dim3 b256(256);
  (void)hipMemsetAsync(ws.hist, 0, kHistStride * sizeof(uint32_t), stream);  // LRES histogram only
  {
    const size_t start = (size_t)(kHeadLen & ~3);
    size_t width = (size_t)g.lres_size + kTreeStride + 64;
    if (start + width > out_cap) width = out_cap - start;
    (void)hipMemsetAsync(d_out + start, 0, width, stream);
  }
  HIMG_LAUNCH(k_lres_predict, dim3((g.mcols + 3) / 4, g.mrows, g.C), dim3(64), g, ws.low, ws.plane_stride,
              ws.lres_sym, ws.lres_stride, lt);
  HIMG_LAUNCH(k_lres_summary, dim3(g.lres_spans, 1), b256, g, ws);
  HIMG_LAUNCH(k_tok_hist<256>, dim3(g.lres_spans, 1), b256, g, ws, 0);
  HIMG_LAUNCH(k_tree, dim3(1, 1), dim3(kTreeThreads), ws, 0);
  HIMG_LAUNCH(k_span_bits, dim3((g.lres_spans + 3) / 4, 1), b256, g, ws, 0, g.lres_spans, (uint32_t *)nullptr);
  HIMG_LAUNCH(k_sizes, dim3(1), b256, g, ws, sc, d_out, out_cap, d_size, d_all_row_bits, 0, 0, 0);
  launch_emit(g, ws, d_out, out_cap, d_size, 0, g.lres_spans, 1, stream, prof);
  HIMG_LAUNCH(k_place_fres, dim3(1024), b256, g, ws, d_rel, rel_bytes, d_out, d_size);
  HIMG_LAUNCH(k_padfix, dim3((g.rows + 3) / 4, 1), b256, g, ws, d_out, out_cap, d_size);
}

// Rank 0 of a row-sharded encode, final-placement form: everything of the stream that does
// not come from another rank goes straight into the stream buffer -- container, LRES stream,
// FRES tree, EVERY row's size header (rank 0 knows all row sizes) and the payloads of its own
// rows [r0, r1) -- and d_head tells the host where the first row header lies ([0], bytes)
// and how long the stream is ([1]): the other ranks' packed rows are then received at their
// final offsets (no relative buffer on rank 0, no 262 MB k_place_fres copy at 16384^2), and
// all of this runs while the peers still pack and send.  launch_shard_finish resolves the
// stale pad bits (trap T1) once every row has arrived.
__global__ void k_shard_head_info(Geom g, EncWs ws, const uint32_t *size, uint32_t *head) {
  const uint32_t b0 = ws.span_bits[g.lres_spans];   // row 0
  const uint32_t n0 = (b0 + 7) >> 3;
  const uint32_t hdr0 = g.use_blocks ? (n0 <= 0x7fffu ? 2u : 4u) : 0u;
  head[0] = (uint32_t)(ws.span_bit0[g.lres_spans] >> 3) - hdr0;
  head[1] = size[0];
}

void launch_shard_head(const Geom &g, const EncWs &ws, const StaticChunks &sc, const LresTables &lt,
                       const uint32_t *d_all_row_bits, uint8_t *d_out, size_t out_cap, uint32_t *d_size,
                       uint32_t *d_head, int r0, int r1, hipStream_t stream, Profiler *prof) {
  const dim3 b256(256);
  (void)hipMemsetAsync(ws.hist, 0, kHistStride * sizeof(uint32_t), stream);  // LRES histogram only
  {
    const size_t start = (size_t)(kHeadLen & ~3);
    size_t width = (size_t)g.lres_size + kTreeStride + 64;
    if (start + width > out_cap) width = out_cap - start;
    (void)hipMemsetAsync(d_out + start, 0, width, stream);
  }
  HIMG_LAUNCH(k_lres_predict, dim3((g.mcols + 3) / 4, g.mrows, g.C), dim3(64), g, ws.low, ws.plane_stride,
              ws.lres_sym, ws.lres_stride, lt);
  HIMG_LAUNCH(k_lres_summary, dim3(g.lres_spans, 1), b256, g, ws);
  HIMG_LAUNCH(k_tok_hist<256>, dim3(g.lres_spans, 1), b256, g, ws, 0);
  HIMG_LAUNCH(k_tree, dim3(1, 1), dim3(kTreeThreads), ws, 0);
  HIMG_LAUNCH(k_span_bits, dim3((g.lres_spans + 3) / 4, 1), b256, g, ws, 0, g.lres_spans, (uint32_t *)nullptr);
  HIMG_LAUNCH(k_sizes, dim3(1), b256, g, ws, sc, d_out, out_cap, d_size, d_all_row_bits, 0, 0, g.rows);
  hipLaunchKernelGGL(k_shard_head_info, dim3(1), dim3(1), 0, stream, g, ws, d_size, d_head);
  launch_emit(g, ws, d_out, out_cap, d_size, 0, g.lres_spans, 1, stream, prof);
  if (r1 > r0) {
    if (ws.tok) launch_emit_tok(g, ws, d_out, out_cap, d_size, r0, r1, 1, stream, prof);
    else launch_emit(g, ws, d_out, out_cap, d_size, g.lres_spans + r0, g.lres_spans + r1, 1, stream, prof);
  }
}

void launch_shard_finish(const Geom &g, const EncWs &ws, uint8_t *d_out, size_t out_cap, const uint32_t *d_size,
                         hipStream_t stream, Profiler *prof) {
  HIMG_LAUNCH(k_padfix, dim3((g.rows + 3) / 4, 1), dim3(256), g, ws, d_out, out_cap, d_size);
}

}  // namespace himg_dev
