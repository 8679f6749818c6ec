/*
 * himg_oracle.c -- scalar CPU restatement of the reference HIMG codec.
 *
 * TEST INFRASTRUCTURE ONLY (see himg_oracle.h).  Written from the algorithm
 * description in SURVEY.md Appendix A/B and checked line by line against the
 * reference; every function cites the reference file:line it follows
 * (paths relative to /root/reference/src/lib).  Plain C99, no dependencies.
 *
 * Parity status: PINNED against the real reference (oracle/_ref) and the
 * golden vectors in tests/golden/ -- see tests/test_oracle_golden.py.
 */
#include "himg_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* ------------------------------------------------------------------------ */
/* Format constants (these tables ARE the format).   */
/* ------------------------------------------------------------------------ */

/* L-shell coefficient scan order, common.cpp:13-22. */
static const uint8_t kIndexLUT[64] = {
    0,  1,  9,  8,  16, 17, 18, 10, 2,  3,  11, 19, 27, 26, 25, 24,
    32, 33, 34, 35, 36, 28, 20, 12, 4,  5,  13, 21, 29, 37, 45, 44,
    43, 42, 41, 40, 48, 49, 50, 51, 52, 53, 54, 46, 38, 30, 22, 14,
    6,  7,  15, 23, 31, 39, 47, 55, 63, 62, 61, 60, 59, 58, 57, 56};

/* quantize.cpp:19-28 */
static const uint8_t kShiftTableBase[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};

/* quantize.cpp:31-40 */
static const uint8_t kChromaShiftTableBase[64] = {
    17,  18,  24,  47,  100, 110, 115, 120, 18,  21,  26,  66,  100,
    110, 118, 121, 24,  26,  56,  100, 100, 110, 120, 122, 47,  66,
    100, 100, 100, 110, 120, 123, 100, 100, 100, 100, 100, 110, 120,
    124, 110, 110, 110, 110, 110, 110, 110, 123, 120, 120, 120, 120,
    120, 110, 100, 122, 124, 124, 126, 126, 125, 123, 122, 105};

/* quantize.cpp:55-65 */
static const int kQualityToScale[9][2] = {
    {0, 65535}, {10, 32512}, {20, 13568}, {30, 5120}, {40, 2560},
    {50, 1024}, {60, 768},   {80, 256},   {100, 0}};

/* mapper.cpp:19-36 */
static const int16_t kLowResMappingTable[128] = {
    0,   1,   2,   3,   4,   5,   6,   7,   8,   9,   10,  11,  12,  13,  14,
    15,  16,  17,  18,  19,  20,  21,  22,  23,  24,  25,  26,  27,  28,  29,
    30,  31,  32,  33,  34,  35,  36,  37,  38,  39,  40,  41,  42,  43,  44,
    45,  46,  47,  48,  49,  50,  51,  52,  53,  54,  55,  56,  57,  58,  59,
    60,  61,  62,  63,  64,  65,  67,  68,  70,  71,  73,  74,  76,  78,  79,
    81,  83,  85,  87,  89,  91,  93,  95,  97,  99,  102, 104, 106, 109, 111,
    114, 117, 119, 122, 125, 128, 131, 134, 137, 140, 143, 146, 150, 153, 156,
    160, 164, 167, 171, 175, 178, 182, 186, 190, 195, 199, 203, 207, 212, 216,
    221, 226, 230, 235, 240, 245, 250, 255};

/* mapper.cpp:38-47 */
static const int kLowResMapScale[8][2] = {{0, 120}, {5, 90},  {10, 70},
                                          {20, 40}, {30, 32}, {40, 26},
                                          {50, 20}, {100, 16}};

/* mapper.cpp:54-71 */
static const int16_t kFullResMappingTable[128] = {
    0,    1,    2,    3,    4,    5,    6,    7,    8,    9,    10,   11,
    12,   13,   14,   15,   16,   17,   18,   19,   20,   21,   22,   23,
    24,   25,   26,   27,   28,   29,   30,   31,   32,   33,   34,   35,
    36,   37,   38,   39,   40,   41,   42,   43,   44,   45,   46,   47,
    48,   49,   51,   52,   54,   57,   59,   62,   65,   68,   72,   76,
    81,   86,   92,   98,   105,  113,  121,  130,  140,  151,  163,  176,
    190,  205,  221,  239,  259,  280,  303,  327,  354,  382,  413,  446,
    482,  520,  561,  605,  653,  703,  757,  815,  876,  942,  1013, 1087,
    1167, 1252, 1342, 1438, 1540, 1649, 1764, 1885, 2015, 2151, 2296, 2450,
    2612, 2783, 2965, 3156, 3358, 3571, 3796, 4032, 4282, 4545, 4821, 5112,
    5418, 5740, 6078, 6433, 6806, 7198, 7608, 8039};

/* huffman_common.h:18-31 */
enum {
  kNumSymbols = 261,
  kSymbolSize = 9,
  kSymTwoZeros = 256,
  kSymUpTo6Zeros = 257,
  kSymUpTo22Zeros = 258,
  kSymUpTo278Zeros = 259,
  kSymUpTo16662Zeros = 260,
  kMaxTreeNodes = 2 * 261 - 1,
  kMaxTreeDataSize = ((2 + 9) * 261 + 7) / 8 /* huffman_enc.cpp:22 */
};

/* ------------------------------------------------------------------------ */
/* Tables                                                                    */
/* ------------------------------------------------------------------------ */

/* quantize.cpp:72-92 and mapper.cpp:75-97 (same interpolation). */
static int interp_scale(int quality, const int (*tab)[2], int n) {
  int idx;
  for (idx = 0; idx < n - 1; ++idx)
    if (tab[idx + 1][0] > quality) break;
  if (idx >= n - 1) return tab[n - 1][1];
  int q1 = tab[idx][0], s1 = tab[idx][1];
  int q2 = tab[idx + 1][0], s2 = tab[idx + 1][1];
  int denom = q2 - q1;
  return s1 + ((s2 - s1) * (quality - q1) + (denom >> 1)) / denom;
}

/* quantize.cpp:94-102 */
static uint8_t nearest_log2(uint16_t x) {
  uint8_t y = 0, rounding = 0;
  while (x > 1) {
    ++y;
    rounding = x & 1;
    x = x >> 1;
  }
  return (uint8_t)(y + rounding);
}

/* quantize.cpp:104-125; quality arrives as uint8_t (quantize.h:21). */
void himg_oracle_shift_table(int quality, int chroma, uint8_t out[64]) {
  const uint8_t *base = chroma ? kChromaShiftTableBase : kShiftTableBase;
  const int scale = interp_scale((uint8_t)quality, kQualityToScale, 9);
  for (int i = 0; i < 64; ++i) {
    uint16_t coeff = (uint16_t)(((int)base[i] * scale + 512) >> 10);
    uint8_t s = nearest_log2(coeff);
    out[i] = s < 15 ? s : 15;
  }
}

/* mapper.cpp:193-211 (positive half only; negatives are mirrored on use). */
void himg_oracle_lowres_map_table(int quality, int16_t out[128]) {
  int16_t index_scale = (int16_t)interp_scale(quality, kLowResMapScale, 8);
  for (int16_t i = 0; i < 128; ++i) {
    int16_t index = (int16_t)((i * index_scale + 8) >> 4);
    if (index > 127) index = 127;
    out[i] = kLowResMappingTable[index];
  }
}

/* mapper.cpp:213-223 */
void himg_oracle_fullres_map_table(int16_t out[128]) {
  memcpy(out, kFullResMappingTable, sizeof(kFullResMappingTable));
}

/* mapper.cpp:159-182 */
uint8_t himg_oracle_map_to_8bit(const int16_t t[128], int xi) {
  int16_t x = (int16_t)xi;
  if (!x) return 0;
  int16_t abs_x = (int16_t)(x < 0 ? -x : x);
  uint8_t mapped;
  for (mapped = 1; mapped < 127 - 1; ++mapped) {
    if (abs_x < t[mapped + 1]) {
      if ((abs_x - t[mapped]) < (t[mapped + 1] - abs_x)) --mapped;
      break;
    }
  }
  if (mapped < 127) ++mapped;
  return x >= 0 ? mapped : (uint8_t)(-(int8_t)mapped);
}

/* mapper.h:33-35 with the mirrored negative half (mapper.cpp:208-210). */
static inline int16_t unmap8(const int16_t t[128], uint8_t c) {
  int8_t s = (int8_t)c;
  if (s >= 0) return t[s];
  if (s == -128) return (int16_t)-t[127]; /* mapper.cpp:154 (decoder side) */
  return (int16_t)-t[-s];
}

/* mapper.cpp:184-191 */
static int single_byte_items(const int16_t t[128]) {
  int i;
  for (i = 1; i < 128; ++i)
    if (t[i] >= 256) break;
  return i - 1;
}

/* mapper.cpp:105-125; returns bytes written. */
static int put_mapping_function(const int16_t t[128], uint8_t *out) {
  int n1 = single_byte_items(t), i;
  uint8_t *p = out;
  *p++ = (uint8_t)n1;
  for (i = 0; i < n1; ++i) *p++ = (uint8_t)t[i + 1];
  for (; i < 127; ++i) {
    uint16_t x = (uint16_t)t[i + 1];
    *p++ = (uint8_t)(x & 255);
    *p++ = (uint8_t)(x >> 8);
  }
  return (int)(p - out);
}

/* mapper.cpp:127-157 */
static int get_mapping_function(int16_t t[128], const uint8_t *in, int size) {
  if (size < 1) return 0;
  int n1 = *in++;
  if (1 + n1 + 2 * (127 - n1) != size) return 0;
  int i;
  t[0] = 0;
  for (i = 0; i < n1; ++i) t[i + 1] = (int16_t)(uint16_t)(*in++);
  for (; i < 127; ++i) {
    t[i + 1] = (int16_t)((uint16_t)in[0] | ((uint16_t)in[1] << 8));
    in += 2;
  }
  return 1;
}

/* ------------------------------------------------------------------------ */
/* Colour lift  (ycbcr.cpp)                                                  */
/* ------------------------------------------------------------------------ */

static inline uint8_t clamp8(int x) { return x < 0 ? 0 : (x > 255 ? 255 : (uint8_t)x); }

/* ycbcr.cpp:24-52 */
static void rgb_to_ycbcr(uint8_t *out, const uint8_t *in, int width, int height,
                         int pixel_stride, int num_channels) {
  for (long n = (long)width * height; n > 0; --n) {
    int16_t r = in[0], g = in[1], b = in[2];
    out[0] = (uint8_t)((r + 2 * g + b + 2) >> 2);
    out[1] = (uint8_t)((b - g + 256) >> 1);
    out[2] = (uint8_t)((r - g + 256) >> 1);
    for (int c = 3; c < num_channels; ++c) out[c] = in[c];
    in += pixel_stride;
    out += pixel_stride;
  }
}

/* ycbcr.cpp:54-82 */
static void ycbcr_to_rgb(uint8_t *buf, int width, int height, int num_channels) {
  for (long n = (long)width * height; n > 0; --n) {
    int16_t y = buf[0];
    int16_t cb = (int16_t)((buf[1] << 1) - 255);
    int16_t cr = (int16_t)((buf[2] << 1) - 255);
    int16_t g = (int16_t)(y - ((cb + cr + 2) >> 2));
    int16_t b = (int16_t)(g + cb);
    int16_t r = (int16_t)(g + cr);
    buf[0] = clamp8(r);
    buf[1] = clamp8(g);
    buf[2] = clamp8(b);
    buf += num_channels;
  }
}

/* ------------------------------------------------------------------------ */
/* Low-res plane  (downsampled.cpp)                                          */
/* ------------------------------------------------------------------------ */

/* downsampled.cpp:67-114.  pixels points at the channel's first byte. */
static void sample_image(const uint8_t *pixels, int stride, int width,
                         int height, uint8_t *avg, uint8_t *m) {
  const int rows = (height + 7) >> 3, cols = (width + 7) >> 3;
  for (int v = 0; v < rows; ++v) {
    int y_min = v * 8 - 3 < 0 ? 0 : v * 8 - 3;
    int y_max = v * 8 + 4 > height - 1 ? height - 1 : v * 8 + 4;
    for (int u = 0; u < cols; ++u) {
      int x_min = u * 8 - 3 < 0 ? 0 : u * 8 - 3;
      int x_max = u * 8 + 4 > width - 1 ? width - 1 : u * 8 + 4;
      uint16_t sum = 0;
      for (int y = y_min; y <= y_max; ++y)
        for (int x = x_min; x <= x_max; ++x)
          sum = (uint16_t)(sum + pixels[((long)y * width + x) * stride]);
      int cnt = (x_max - x_min + 1) * (y_max - y_min + 1);
      avg[v * cols + u] = (uint8_t)((sum + (cnt >> 1)) / cnt);
    }
  }
  for (int v = 0; v < rows; ++v) {
    int row1 = v - 1 < 0 ? 0 : v - 1, row2 = v;
    for (int u = 0; u < cols; ++u) {
      int col1 = u - 1 < 0 ? 0 : u - 1, col2 = u;
      uint16_t x11 = avg[row1 * cols + col1], x12 = avg[row1 * cols + col2];
      uint16_t x21 = avg[row2 * cols + col1], x22 = avg[row2 * cols + col2];
      uint16_t a1 = (uint16_t)((1 * x11 + 15 * x12 + 8) >> 4);
      uint16_t a2 = (uint16_t)((1 * x21 + 15 * x22 + 8) >> 4);
      m[v * cols + u] = (uint8_t)((1 * a1 + 15 * a2 + 8) >> 4);
    }
  }
}

/* downsampled.cpp:116-169 */
static void interp9(int16_t a[9]) {
  a[4] = (int16_t)((a[0] + a[8] + 1) >> 1);
  a[2] = (int16_t)((a[0] + a[4] + 1) >> 1);
  a[6] = (int16_t)((a[4] + a[8] + 1) >> 1);
  a[1] = (int16_t)((a[0] + a[2] + 1) >> 1);
  a[3] = (int16_t)((a[2] + a[4] + 1) >> 1);
  a[5] = (int16_t)((a[4] + a[6] + 1) >> 1);
  a[7] = (int16_t)((a[6] + a[8] + 1) >> 1);
}

static void get_lowres_block(const uint8_t *m, int rows, int cols, int16_t *out,
                             int u, int v) {
  int row2 = v + 1 > rows - 1 ? rows - 1 : v + 1;
  int col2 = u + 1 > cols - 1 ? cols - 1 : u + 1;
  int16_t left[9], right[9];
  left[0] = m[v * cols + u];
  left[8] = m[row2 * cols + u];
  right[0] = m[v * cols + col2];
  right[8] = m[row2 * cols + col2];
  interp9(left);
  interp9(right);
  for (int y = 0; y < 8; ++y) {
    int16_t a[9];
    a[0] = left[y];
    a[8] = right[y];
    interp9(a);
    for (int x = 0; x < 8; ++x) *out++ = a[x];
  }
}

/* downsampled.cpp:41-60 */
static int16_t predict_sample(int16_t s1, int16_t s2, int16_t s3, int predictor) {
  switch (predictor) {
    default:
    case 0: return clamp8((3 * (s2 + s3) - 2 * s1 + 2) >> 2);
    case 1: return s2;
    case 2: return s3;
    case 3: return (int16_t)((s2 + s3 + 1) >> 1);
    case 4: return clamp8(s2 + s3 - s1);
  }
}

static int num_macro(int blocks) { return (blocks + 15) / 16; }

/* downsampled.cpp:171-175 */
static int block_data_size_per_channel(int rows, int cols) {
  return num_macro(rows) * num_macro(cols) + rows * cols;
}

/* downsampled.cpp:177-316 */
static void get_block_data(const uint8_t *m, int rows, int cols, uint8_t *out,
                           const int16_t *map) {
  const int mrows = num_macro(rows), mcols = num_macro(cols);
  uint8_t *predictor_selection = out;
  for (int mv = 0; mv < mrows; ++mv) {
    for (int mu = 0; mu < mcols; ++mu) {
      int err[5] = {0, 0, 0, 0, 0};
      for (int dv = 0; dv < 16; ++dv) {
        int v = mv * 16 + dv;
        if (v >= rows) break;
        for (int du = 0; du < 16; ++du) {
          int u = mu * 16 + du;
          if (u >= cols) break;
          int16_t s1, s2, s3;
          if (du > 0 && dv > 0) {
            s1 = m[(v - 1) * cols + u - 1];
            s2 = m[(v - 1) * cols + u];
            s3 = m[v * cols + u - 1];
          } else if (du > 0) {
            s1 = s2 = s3 = m[v * cols + u - 1];
          } else if (dv > 0) {
            s1 = s2 = s3 = m[(v - 1) * cols + u];
          } else {
            s1 = s2 = s3 = 128;
          }
          for (int p = 0; p < 5; ++p) {
            int delta = (int)m[v * cols + u] - predict_sample(s1, s2, s3, p);
            err[p] += delta * delta;
          }
        }
      }
      int best = 0, best_err = err[0];
      for (int p = 1; p < 5; ++p)
        if (err[p] < best_err) {
          best = p;
          best_err = err[p];
        }
      *out++ = (uint8_t)(best - 2); /* downsampled.cpp:33-35 */
    }
  }

  uint8_t work[32];
  uint8_t *lines[2] = {&work[0], &work[16]};
  for (int mv = 0; mv < mrows; ++mv) {
    for (int mu = 0; mu < mcols; ++mu) {
      /* DecodePredictor (downsampled.cpp:37-39) adds 2 to the uint8 in int
       * arithmetic: selections {0,1} were stored as {254,255} and come back
       * as {256,257}, which PredictSample's `default:` maps to case 0.  So a
       * macro block that SELECTED predictor 1 is CODED with predictor 0, on
       * both the encoder and the decoder side. */
      int predictor = (int)predictor_selection[mv * mcols + mu] + 2;
      for (int dv = 0; dv < 16; ++dv) {
        int v = mv * 16 + dv;
        if (v >= rows) break;
        for (int du = 0; du < 16; ++du) {
          int u = mu * 16 + du;
          if (u >= cols) break;
          int16_t s1, s2, s3;
          if (du > 0 && dv > 0) {
            s1 = lines[0][du - 1];
            s2 = lines[0][du];
            s3 = lines[1][du - 1];
          } else if (du > 0) {
            s1 = s2 = s3 = lines[1][du - 1];
          } else if (dv > 0) {
            s1 = s2 = s3 = lines[0][du];
          } else {
            s1 = s2 = s3 = 128;
          }
          int16_t predicted = predict_sample(s1, s2, s3, predictor);
          int16_t actual = m[v * cols + u];
          int16_t delta = (int16_t)(actual - predicted);
          uint8_t delta8 = himg_oracle_map_to_8bit(map, delta);
          actual = (int16_t)(predicted + unmap8(map, delta8));
          lines[1][du] = clamp8(actual);
          *out++ = delta8;
        }
        uint8_t *t = lines[0];
        lines[0] = lines[1];
        lines[1] = t;
      }
    }
  }
}

/* downsampled.cpp:318-382 */
static void set_block_data(uint8_t *m, const uint8_t *in, int rows, int cols,
                           const int16_t *map) {
  const int mrows = num_macro(rows), mcols = num_macro(cols);
  const uint8_t *predictor_selection = in;
  in += mrows * mcols;
  for (int mv = 0; mv < mrows; ++mv) {
    for (int mu = 0; mu < mcols; ++mu) {
      int predictor = (int)predictor_selection[mv * mcols + mu] + 2;
      for (int dv = 0; dv < 16; ++dv) {
        int v = mv * 16 + dv;
        if (v >= rows) break;
        for (int du = 0; du < 16; ++du) {
          int u = mu * 16 + du;
          if (u >= cols) break;
          int16_t s1, s2, s3;
          if (du > 0 && dv > 0) {
            s1 = m[(v - 1) * cols + u - 1];
            s2 = m[(v - 1) * cols + u];
            s3 = m[v * cols + u - 1];
          } else if (du > 0) {
            s1 = s2 = s3 = m[v * cols + u - 1];
          } else if (dv > 0) {
            s1 = s2 = s3 = m[(v - 1) * cols + u];
          } else {
            s1 = s2 = s3 = 128;
          }
          int16_t predicted = predict_sample(s1, s2, s3, predictor);
          int16_t actual = (int16_t)(predicted + unmap8(map, *in++));
          m[v * cols + u] = clamp8(actual);
        }
      }
    }
  }
}

/* ------------------------------------------------------------------------ */
/* Hadamard  (hadamard.cpp)                                                  */
/* ------------------------------------------------------------------------ */

/* hadamard.cpp:18-44 */
static void forward8(int16_t *out, const int16_t *in, int s) {
  int16_t a0 = (int16_t)(in[0 * s] + in[4 * s]), a1 = (int16_t)(in[1 * s] + in[5 * s]);
  int16_t a2 = (int16_t)(in[2 * s] + in[6 * s]), a3 = (int16_t)(in[3 * s] + in[7 * s]);
  int16_t a4 = (int16_t)(in[0 * s] - in[4 * s]), a5 = (int16_t)(in[1 * s] - in[5 * s]);
  int16_t a6 = (int16_t)(in[2 * s] - in[6 * s]), a7 = (int16_t)(in[3 * s] - in[7 * s]);
  int16_t b0 = (int16_t)(a0 + a2), b1 = (int16_t)(a1 + a3);
  int16_t b2 = (int16_t)(a0 - a2), b3 = (int16_t)(a1 - a3);
  int16_t b4 = (int16_t)(a4 + a6), b5 = (int16_t)(a5 + a7);
  int16_t b6 = (int16_t)(a4 - a6), b7 = (int16_t)(a5 - a7);
  out[0 * s] = (int16_t)(b0 + b1);
  out[1 * s] = (int16_t)(b4 + b5);
  out[2 * s] = (int16_t)(b6 + b7);
  out[3 * s] = (int16_t)(b2 + b3);
  out[4 * s] = (int16_t)(b2 - b3);
  out[5 * s] = (int16_t)(b6 - b7);
  out[6 * s] = (int16_t)(b4 - b5);
  out[7 * s] = (int16_t)(b0 - b1);
}

/* hadamard.cpp:47-74 (SHIFT = 3) */
static void inverse8(int16_t *out, const int16_t *in, int s) {
  int32_t a0 = in[0 * s] + in[4 * s], a1 = in[1 * s] + in[5 * s];
  int32_t a2 = in[2 * s] + in[6 * s], a3 = in[3 * s] + in[7 * s];
  int32_t a4 = in[0 * s] - in[4 * s], a5 = in[1 * s] - in[5 * s];
  int32_t a6 = in[2 * s] - in[6 * s], a7 = in[3 * s] - in[7 * s];
  int32_t b0 = a0 + a2, b1 = a1 + a3, b2 = a0 - a2, b3 = a1 - a3;
  int32_t b4 = a4 + a6, b5 = a5 + a7, b6 = a4 - a6, b7 = a5 - a7;
  out[0 * s] = (int16_t)((b0 + b1) >> 3);
  out[1 * s] = (int16_t)((b4 + b5) >> 3);
  out[2 * s] = (int16_t)((b6 + b7) >> 3);
  out[3 * s] = (int16_t)((b2 + b3) >> 3);
  out[4 * s] = (int16_t)((b2 - b3) >> 3);
  out[5 * s] = (int16_t)((b6 - b7) >> 3);
  out[6 * s] = (int16_t)((b4 - b5) >> 3);
  out[7 * s] = (int16_t)((b0 - b1) >> 3);
}

/* hadamard.cpp:78-88 */
void himg_oracle_hadamard_forward(int16_t *out, const int16_t *in) {
  for (int i = 0; i < 8; ++i) forward8(&out[i * 8], &in[i * 8], 1);
  for (int i = 0; i < 8; ++i) forward8(&out[i], &out[i], 8);
}

/* hadamard.cpp:90-103 */
void himg_oracle_hadamard_inverse(int16_t *out, const int16_t *in) {
  for (int i = 0; i < 8; ++i) inverse8(&out[i * 8], &in[i * 8], 1);
  for (int i = 0; i < 8; ++i) inverse8(&out[i], &out[i], 8);
}

/* ------------------------------------------------------------------------ */
/* Quantize  (quantize.cpp)                                                  */
/* ------------------------------------------------------------------------ */

/* quantize.cpp:127-151 */
static void quantize_pack(uint8_t *out, const int16_t *in, const uint8_t *shift,
                          const int16_t *map) {
  for (int i = 0; i < 64; ++i) {
    uint8_t s = shift[i];
    int16_t round = (int16_t)(s != 0 ? 1 << (s - 1) : 0);
    int16_t x = in[i];
    if (x < 0)
      x = (int16_t)(-((-x + round) >> s));
    else
      x = (int16_t)((x + round) >> s);
    out[i] = himg_oracle_map_to_8bit(map, x);
  }
}

/* quantize.cpp:153-165; the int16 store wraps. */
static void quantize_unpack(int16_t *out, const uint8_t *in, const uint8_t *shift,
                            const int16_t *map) {
  for (int i = 0; i < 64; ++i)
    out[i] = (int16_t)((int32_t)unmap8(map, in[i]) * (1 << shift[i]));
}

/* ------------------------------------------------------------------------ */
/* Entropy coder  (huffman_enc.cpp)                                          */
/* ------------------------------------------------------------------------ */

typedef struct {
  uint8_t *base, *p;
  int bit; /* bits already used in *p */
} bitw;

/* huffman_enc.cpp:31-50.  Bit-at-a-time like the reference so that only the
 * bits actually written change (this is what makes trap T1 observable). */
static void write_bits(bitw *w, uint64_t x, int bits) {
  while (bits--) {
    *w->p = (uint8_t)((*w->p & (0xff ^ (1 << w->bit))) | ((x & 1) << w->bit));
    x >>= 1;
    w->bit = (w->bit + 1) & 7;
    if (!w->bit) ++w->p;
  }
}
static void align_to_byte(bitw *w) { /* huffman_enc.cpp:53-58 */
  if (w->bit) {
    w->bit = 0;
    ++w->p;
  }
}
static int bitw_size(const bitw *w) { /* huffman_enc.cpp:65-71 */
  return (int)(w->p - w->base) + (w->bit > 0);
}

/* Zero-run classification shared by histogram and emit
 * (huffman_enc.cpp:105-141 and :298-338). */
static inline int run_symbol(int zeros, int *extra_bits, uint32_t *extra) {
  if (zeros == 1) { *extra_bits = 0; *extra = 0; return 0; }
  if (zeros == 2) { *extra_bits = 0; *extra = 0; return kSymTwoZeros; }
  if (zeros <= 6) { *extra_bits = 2; *extra = (uint32_t)(zeros - 3); return kSymUpTo6Zeros; }
  if (zeros <= 22) { *extra_bits = 4; *extra = (uint32_t)(zeros - 7); return kSymUpTo22Zeros; }
  if (zeros <= 278) { *extra_bits = 8; *extra = (uint32_t)(zeros - 23); return kSymUpTo278Zeros; }
  *extra_bits = 14; *extra = (uint32_t)(zeros - 279); return kSymUpTo16662Zeros;
}

static inline int zero_run(const uint8_t *block, int k, int block_size) {
  int zeros; /* huffman_enc.cpp:111-115 */
  for (zeros = 1; zeros < 16662 && (k + zeros) < block_size; ++zeros)
    if (block[k + zeros] != 0) break;
  return zeros;
}

typedef struct {
  int child_a, child_b, count, symbol;
} enode;

typedef struct {
  uint32_t count[kNumSymbols];
  uint64_t code[kNumSymbols];
  uint8_t bits[kNumSymbols];
} symtab;

/* huffman_enc.cpp:148-180 */
static void store_tree(const enode *nodes, int n, symtab *st, bitw *w,
                       uint64_t code, int bits) {
  if (nodes[n].symbol >= 0) {
    write_bits(w, 1, 1);
    write_bits(w, (uint64_t)nodes[n].symbol, kSymbolSize);
    st->code[nodes[n].symbol] = code;
    st->bits[nodes[n].symbol] = (uint8_t)bits;
    return;
  }
  write_bits(w, 0, 1);
  store_tree(nodes, nodes[n].child_a, st, w, code, bits + 1);
  store_tree(nodes, nodes[n].child_b, st, w, code + ((uint64_t)1 << bits), bits + 1);
}

/* huffman_enc.cpp:183-238 */
static void make_tree(symtab *st, bitw *w) {
  enode nodes[kMaxTreeNodes];
  int num = 0;
  for (int k = 0; k < kNumSymbols; ++k)
    if (st->count[k] > 0) {
      nodes[num].symbol = k;
      nodes[num].count = (int)st->count[k];
      nodes[num].child_a = nodes[num].child_b = -1;
      ++num;
    }
  int root = -1, left = num, next = num;
  while (left > 1) {
    int n1 = -1, n2 = -1;
    for (int k = 0; k < next; ++k) {
      if (nodes[k].count > 0) {
        if (n1 < 0 || nodes[k].count <= nodes[n1].count) {
          n2 = n1;
          n1 = k;
        } else if (n2 < 0 || nodes[k].count <= nodes[n2].count) {
          n2 = k;
        }
      }
    }
    root = next;
    nodes[root].child_a = n1;
    nodes[root].child_b = n2;
    nodes[root].count = nodes[n1].count + nodes[n2].count;
    nodes[root].symbol = -1;
    nodes[n1].count = 0;
    nodes[n2].count = 0;
    ++next;
    --left;
  }
  if (root >= 0)
    store_tree(nodes, root, st, w, 0, 0);
  else
    store_tree(nodes, 0, st, w, 0, 1); /* single symbol: huffman_enc.cpp:231-237 */
}

/* huffman_enc.cpp:246-363.  Returns the packed size.  row_bytes (optional)
 * receives the payload size of every block; tree_bytes the tree size. */
static int huffman_compress(uint8_t *out, const uint8_t *in, int in_size,
                            int block_size, symtab *st_out, int *tree_bytes,
                            int *row_bytes) {
  if (in_size < 1) return 0;
  if (block_size < 1) block_size = in_size;
  const int use_blocks = block_size < in_size;
  if (in_size % block_size != 0) return 0;

  bitw stream = {out, out, 0};
  symtab st;
  memset(&st, 0, sizeof(st));

  /* Histogram over tokens of all blocks, huffman_enc.cpp:98-144. */
  for (int b = 0; b < in_size; b += block_size) {
    const uint8_t *block = in + b;
    for (int k = 0; k < block_size;) {
      if (block[k] == 0) {
        int eb;
        uint32_t ev;
        int zeros = zero_run(block, k, block_size);
        st.count[run_symbol(zeros, &eb, &ev)]++;
        k += zeros;
      } else {
        st.count[block[k]]++;
        k++;
      }
    }
  }

  /* The output buffer is not zero-initialised by the reference either
   * (encoder.cpp:340: vector::resize value-initialises -> zeros), so start
   * from zeros for the tree bits. */
  memset(out, 0, (size_t)kMaxTreeDataSize + 1);
  make_tree(&st, &stream);
  align_to_byte(&stream);
  if (tree_bytes) *tree_bytes = bitw_size(&stream);

  /* One scratch buffer for all blocks, zero-initialised once and never
   * cleared (huffman_enc.cpp:288) -> stale pad bits, trap T1. */
  uint8_t *scratch = (uint8_t *)calloc((size_t)block_size + 16, 1);
  int row = 0;
  for (int b = 0; b < in_size; b += block_size, ++row) {
    const uint8_t *block = in + b;
    bitw bs = {scratch, scratch, 0};
    for (int k = 0; k < block_size;) {
      uint8_t symbol = block[k];
      if (symbol == 0) {
        int eb;
        uint32_t ev;
        int zeros = zero_run(block, k, block_size);
        int s = run_symbol(zeros, &eb, &ev);
        write_bits(&bs, st.code[s], st.bits[s]);
        if (eb) write_bits(&bs, ev, eb);
        k += zeros;
      } else {
        write_bits(&bs, st.code[symbol], st.bits[symbol]);
        k++;
      }
    }
    const int packed = bitw_size(&bs);
    if (row_bytes) row_bytes[row] = packed;
    if (use_blocks) { /* huffman_enc.cpp:342-352 */
      align_to_byte(&stream);
      stream.p[0] = stream.p[1] = 0;
      if (packed <= 0x7fff) {
        write_bits(&stream, (uint64_t)packed, 16);
      } else {
        stream.p[2] = stream.p[3] = 0;
        write_bits(&stream, (uint64_t)((packed & 0x7fff) | 0x8000), 16);
        write_bits(&stream, (uint64_t)(packed >> 15), 16);
      }
    }
    memcpy(stream.p, scratch, (size_t)packed);
    stream.p += packed;
  }
  free(scratch);
  if (st_out) *st_out = st;
  return bitw_size(&stream);
}

/* huffman_enc.cpp:242-244 */
static int max_compressed_size(int n) { return n + kMaxTreeDataSize; }

/* ------------------------------------------------------------------------ */
/* Encoder  (encoder.cpp)                                                    */
/* ------------------------------------------------------------------------ */

typedef struct {
  uint8_t *data;
  size_t size, cap;
} vec;

static void vec_reserve(vec *v, size_t n) {
  if (n > v->cap) {
    size_t c = v->cap ? v->cap : 256;
    while (c < n) c *= 2;
    v->data = (uint8_t *)realloc(v->data, c);
    v->cap = c;
  }
}
static void vec_push(vec *v, uint8_t b) {
  vec_reserve(v, v->size + 1);
  v->data[v->size++] = b;
}
static void vec_push_u32(vec *v, uint32_t x) {
  for (int i = 0; i < 4; ++i) vec_push(v, (uint8_t)(x >> (8 * i)));
}
static void vec_push_tag(vec *v, const char *t) {
  for (int i = 0; i < 4; ++i) vec_push(v, (uint8_t)t[i]);
}

/* encoder.cpp:337-353 */
static int append_packed(vec *v, const uint8_t *unpacked, int unpacked_size,
                         int block_size, symtab *st, int *tree_bytes,
                         int *row_bytes) {
  size_t base = v->size;
  vec_reserve(v, base + 4 + (size_t)max_compressed_size(unpacked_size) + 64);
  int packed = huffman_compress(v->data + base + 4, unpacked, unpacked_size,
                                block_size, st, tree_bytes, row_bytes);
  v->data[base + 0] = (uint8_t)(packed & 255);
  v->data[base + 1] = (uint8_t)((packed >> 8) & 255);
  v->data[base + 2] = (uint8_t)((packed >> 16) & 255);
  v->data[base + 3] = (uint8_t)((packed >> 24) & 255);
  v->size = base + 4 + (size_t)packed;
  return packed;
}

/* encoder.cpp:26-52 */
static void extract_channel_block(int16_t *out, const uint8_t *in, int channel,
                                  int pixel_stride, int row_stride, int bw, int bh) {
  int16_t col = 0;
  int x, y;
  for (y = 0; y < bh; y++) {
    for (x = 0; x < bw; x++) {
      col = in[channel];
      in += pixel_stride;
      *out++ = col;
    }
    for (; x < 8; x++) *out++ = col;
    in += row_stride - (pixel_stride * bw);
  }
  for (; y < 8; y++)
    for (x = 0; x < 8; x++) *out++ = col;
}

static void fill_trace_codes(const symtab *st, uint32_t *hist, uint8_t *len,
                             uint64_t *code) {
  for (int i = 0; i < kNumSymbols; ++i) {
    hist[i] = st->count[i];
    len[i] = st->bits[i];
    code[i] = st->code[i];
  }
}

/* encoder.cpp:59-109 */
int himg_oracle_encode(const uint8_t *data, int width, int height,
                       int pixel_stride, int num_channels, int quality,
                       int use_ycbcr, uint8_t **out, int *out_size,
                       himg_oracle_trace *tr) {
  if (!data || width < 1 || height < 1 || num_channels < 1 ||
      pixel_stride < num_channels)
    return -1;
  const int ycbcr = use_ycbcr && num_channels >= 3;
  const int rows = (height + 7) >> 3, cols = (width + 7) >> 3;
  vec v = {0, 0, 0};
  if (tr) {
    memset(tr, 0, sizeof(*tr));
    tr->width = width; tr->height = height; tr->channels = num_channels;
    tr->rows = rows; tr->cols = cols; tr->use_ycbcr = ycbcr;
  }

  /* RIFF start + FRMT, encoder.cpp:111-166. */
  vec_push_tag(&v, "RIFF");
  vec_push_u32(&v, 0);
  vec_push_tag(&v, "HIMG");
  vec_push_tag(&v, "FRMT");
  vec_push_u32(&v, 11);
  vec_push(&v, 1);
  vec_push_u32(&v, (uint32_t)width);
  vec_push_u32(&v, (uint32_t)height);
  vec_push(&v, (uint8_t)num_channels);
  vec_push(&v, ycbcr ? 1 : 0);

  /* Colour lift into a temporary copy, encoder.cpp:78-85. */
  const uint8_t *cs = data;
  uint8_t *lifted = NULL;
  if (ycbcr) {
    lifted = (uint8_t *)malloc((size_t)width * height * pixel_stride);
    rgb_to_ycbcr(lifted, data, width, height, pixel_stride, num_channels);
    cs = lifted;
  }

  /* LMAP, encoder.cpp:88-89,168-184. */
  int16_t lmap[128], fmap[128];
  himg_oracle_lowres_map_table(quality, lmap);
  {
    uint8_t buf[256];
    int n = put_mapping_function(lmap, buf);
    vec_push_tag(&v, "LMAP");
    vec_push_u32(&v, (uint32_t)n);
    for (int i = 0; i < n; ++i) vec_push(&v, buf[i]);
  }

  /* LRES, encoder.cpp:186-220. */
  uint8_t *avg = (uint8_t *)malloc((size_t)rows * cols * num_channels);
  uint8_t *low = (uint8_t *)malloc((size_t)rows * cols * num_channels);
  const int chan_size = block_data_size_per_channel(rows, cols);
  const int lres_size = chan_size * num_channels;
  uint8_t *lres = (uint8_t *)malloc((size_t)lres_size);
  vec_push_tag(&v, "LRES");
  for (int c = 0; c < num_channels; ++c)
    sample_image(cs + c, pixel_stride, width, height, avg + (size_t)c * rows * cols,
                 low + (size_t)c * rows * cols);
  for (int c = 0; c < num_channels; ++c)
    get_block_data(low + (size_t)c * rows * cols, rows, cols,
                   lres + (size_t)c * chan_size, lmap);
  {
    symtab st;
    int tb = 0;
    append_packed(&v, lres, lres_size, 0, &st, &tb, NULL);
    if (tr) {
      fill_trace_codes(&st, tr->lres_hist, tr->lres_len, tr->lres_code);
      tr->lres_tree_bytes = tb;
    }
  }

  /* QCFG, encoder.cpp:95-96,222-238 + quantize.cpp:174-187. */
  uint8_t shift_l[64], shift_c[64];
  himg_oracle_shift_table(quality, 0, shift_l);
  himg_oracle_shift_table(quality, 1, shift_c);
  vec_push_tag(&v, "QCFG");
  vec_push_u32(&v, ycbcr ? 64u : 32u);
  for (int i = 0; i < 32; ++i)
    vec_push(&v, (uint8_t)((shift_l[i * 2] << 4) | shift_l[i * 2 + 1]));
  if (ycbcr)
    for (int i = 0; i < 32; ++i)
      vec_push(&v, (uint8_t)((shift_c[i * 2] << 4) | shift_c[i * 2 + 1]));

  /* FMAP, encoder.cpp:99-100,240-256. */
  himg_oracle_fullres_map_table(fmap);
  {
    uint8_t buf[256];
    int n = put_mapping_function(fmap, buf);
    vec_push_tag(&v, "FMAP");
    vec_push_u32(&v, (uint32_t)n);
    for (int i = 0; i < n; ++i) vec_push(&v, buf[i]);
  }

  /* FRES, encoder.cpp:258-335. */
  const int fres_size = rows * cols * 64 * num_channels;
  uint8_t *fres = (uint8_t *)malloc((size_t)fres_size);
  vec_push_tag(&v, "FRES");
  {
    long idx = 0;
    for (int y = 0; y < height; y += 8) {
      int vv = y >> 3;
      for (int c = 0; c < num_channels; ++c) {
        const uint8_t *m = low + (size_t)c * rows * cols;
        const uint8_t *shift = (ycbcr && (c == 1 || c == 2)) ? shift_c : shift_l;
        for (int x = 0; x < width; x += 8) {
          int u = x >> 3;
          int bw = width - x < 8 ? width - x : 8;
          int bh = height - y < 8 ? height - y : 8;
          int16_t buf0[64], lowres[64], buf1[64];
          uint8_t packed[64];
          extract_channel_block(buf0, &cs[((long)y * width + x) * pixel_stride], c,
                                pixel_stride, width * pixel_stride, bw, bh);
          get_lowres_block(m, rows, cols, lowres, u, vv);
          for (int i = 0; i < 64; ++i) buf0[i] = (int16_t)(buf0[i] - lowres[i]);
          himg_oracle_hadamard_forward(buf1, buf0);
          quantize_pack(packed, buf1, shift, fmap);
          for (int i = 0; i < 64; ++i)
            fres[idx + u + (long)i * cols] = packed[kIndexLUT[i]];
        }
        idx += (long)cols * 64;
      }
    }
  }
  {
    symtab st;
    int tb = 0;
    int *rb = (int *)calloc((size_t)rows, sizeof(int));
    append_packed(&v, fres, fres_size, cols * num_channels * 64, &st, &tb, rb);
    if (tr) {
      fill_trace_codes(&st, tr->fres_hist, tr->fres_len, tr->fres_code);
      tr->fres_tree_bytes = tb;
      tr->fres_row_bytes = rb;
    } else {
      free(rb);
    }
  }

  /* encoder.cpp:131-137 */
  {
    uint32_t fs = (uint32_t)(v.size - 8);
    for (int i = 0; i < 4; ++i) v.data[4 + i] = (uint8_t)(fs >> (8 * i));
  }

  if (tr) {
    tr->lifted = lifted; lifted = NULL;
    tr->avg = avg; tr->lowres = low;
    tr->lres_sym = lres; tr->lres_sym_size = lres_size;
    tr->fres_sym = fres; tr->fres_sym_size = fres_size;
    memcpy(tr->shift_luma, shift_l, 64);
    memcpy(tr->shift_chroma, shift_c, 64);
    memcpy(tr->lmap, lmap, sizeof(lmap));
    memcpy(tr->fmap, fmap, sizeof(fmap));
  } else {
    free(avg); free(low); free(lres); free(fres);
  }
  free(lifted);
  *out = v.data;
  *out_size = (int)v.size;
  return 0;
}

/* ------------------------------------------------------------------------ */
/* Entropy decoder  (huffman_dec.cpp)                                        */
/* ------------------------------------------------------------------------ */

typedef struct {
  int child_a, child_b, symbol;
} dnode;

typedef struct {
  const uint8_t *p, *end;
  int bit;
  int failed;
} bitr;

typedef struct {
  dnode nodes[kMaxTreeNodes + 1];
  int root;
  /* 8-bit peek LUT, huffman_dec.cpp:173-185,191-198 */
  int lut_node[256], lut_symbol[256], lut_bits[256];
  bitr stream; /* positioned after the byte-aligned tree */
  int block_size, use_blocks;
  int one_bit_leaf; /* fixed mode only: the tree is one leaf, read 1 bit per code */
  int num_blocks;
  const uint8_t **block_ptr;
  int *block_len;
} hdec;

/* Safe byte fetch: the reference's unchecked fast loop may peek one byte past
 * the payload (huffman_dec.cpp:109-112); out-of-range bytes read as zero here. */
static inline unsigned rd(const bitr *r, const uint8_t *p, const uint8_t *hard_end) {
  (void)r;
  return p < hard_end ? *p : 0u;
}

static int read_bit_checked(bitr *r) { /* huffman_dec.cpp:51-60 */
  if (r->p >= r->end) {
    r->failed = 1;
    return 0;
  }
  int x = (*r->p >> r->bit) & 1;
  r->bit = (r->bit + 1) & 7;
  if (!r->bit) ++r->p;
  return x;
}

static uint32_t read_bits_checked(bitr *r, int bits) { /* huffman_dec.cpp:62-106 */
  int nb = r->bit + bits;
  const uint8_t *np = r->p + (nb >> 3);
  if (np > r->end || (np == r->end && (nb & 7) > 0)) {
    r->failed = 1;
    return 0;
  }
  uint32_t x = 0;
  for (int i = 0; i < bits; ++i) {
    x |= (uint32_t)((*r->p >> r->bit) & 1) << i;
    r->bit = (r->bit + 1) & 7;
    if (!r->bit) ++r->p;
  }
  return x;
}

/* huffman_dec.cpp:152-213 */
static int recover_tree(hdec *d, int *nodenum, uint32_t code, int bits) {
  if (*nodenum >= kMaxTreeNodes) return -1; /* intent of :157-158 */
  int me = (*nodenum)++;
  d->nodes[me].symbol = -1;
  d->nodes[me].child_a = d->nodes[me].child_b = -1;
  int is_leaf = read_bit_checked(&d->stream) != 0;
  if (d->stream.failed) return -1;
  if (is_leaf) {
    int symbol = (int)read_bits_checked(&d->stream, kSymbolSize);
    if (d->stream.failed) return -1;
    d->nodes[me].symbol = symbol;
    if (bits <= 8) {
      uint32_t dups = 256u >> bits;
      for (uint32_t i = 0; i < dups; ++i) {
        uint32_t e = (i << bits) | code;
        d->lut_node[e] = -1;
        d->lut_bits[e] = bits;
        d->lut_symbol[e] = symbol;
      }
    }
    return me;
  }
  if (bits == 8) {
    d->lut_node[code] = me;
    d->lut_bits[code] = 8;
    d->lut_symbol[code] = 0;
  }
  if (bits >= 40) return -1; /* keep the recursion bounded on hostile input */
  d->nodes[me].child_a = recover_tree(d, nodenum, code, bits + 1);
  if (d->nodes[me].child_a < 0) return -1;
  d->nodes[me].child_b = recover_tree(d, nodenum, code + (1u << (bits & 31)), bits + 1);
  if (d->nodes[me].child_b < 0) return -1;
  return me;
}

/* huffman_dec.cpp:140-145 */
static int at_the_end(const uint8_t *p, int bit, const uint8_t *end) {
  return (p == end && bit == 0) || (p == end - 1 && bit > 0);
}

/* Opt-in "fixed" mode used only to check the product's own compatibility switch
 * (himg_hip_set_option HIMG_OPT_FIX_T2): derive use_blocks the way the ENCODER
 * does (huffman_enc.cpp:256), from the uncompressed size.  Off by default: the
 * oracle then is the reference, trap T2 included. */
static int g_fix_t2 = 0;
void himg_oracle_set_compat_fix(int on) { g_fix_t2 = on; }

/* huffman_dec.cpp:215-251.  total_out: uncompressed size of the whole stream
 * (only looked at in the fixed mode above). */
static int hdec_init(hdec *d, const uint8_t *in, int in_size, int block_size, long total_out) {
  memset(d, 0, sizeof(*d));
  d->stream.p = in;
  d->stream.end = in + in_size;
  d->block_size = block_size > 0 ? block_size : in_size;
  d->use_blocks = d->block_size < in_size; /* trap T2: compares with the COMPRESSED size */
  if (g_fix_t2 && block_size > 0) d->use_blocks = (long)block_size < total_out;
  d->one_bit_leaf = 0;
  int count = 0;
  d->root = recover_tree(d, &count, 0, 0);
  if (d->root < 0) return 0;
  if (g_fix_t2 && d->nodes[d->root].symbol >= 0) {
    /* fixed mode: a one-symbol tree was WRITTEN with 1-bit codes
     * (huffman_enc.cpp:231-237) although the reference reads it with 0 bits */
    for (int e = 0; e < 256; ++e) d->lut_bits[e] = 1;
    d->one_bit_leaf = 1;
  }
  if (d->stream.bit) { /* AlignToByte */
    d->stream.bit = 0;
    ++d->stream.p;
  }
  if (d->use_blocks) {
    const uint8_t *p = d->stream.p, *end = d->stream.end;
    int cap = 64;
    d->block_ptr = (const uint8_t **)malloc(sizeof(*d->block_ptr) * (size_t)cap);
    d->block_len = (int *)malloc(sizeof(int) * (size_t)cap);
    /* tmp_stream is byte aligned here, so AtTheEnd() reduces to p == end. */
    while (p != end) {
      if (p + 2 > end) return 0; /* reference reads unchecked (:244 TODO) */
      uint32_t n = (uint32_t)p[0] | ((uint32_t)p[1] << 8);
      p += 2;
      if (n & 0x8000) {
        if (p + 2 > end) return 0;
        n = (n & 0x7fff) | (((uint32_t)p[0] | ((uint32_t)p[1] << 8)) << 15);
        p += 2;
      }
      if ((long)n > end - p) return 0; /* reference would walk off the buffer */
      if (d->num_blocks == cap) {
        cap *= 2;
        d->block_ptr = (const uint8_t **)realloc(d->block_ptr, sizeof(*d->block_ptr) * (size_t)cap);
        d->block_len = (int *)realloc(d->block_len, sizeof(int) * (size_t)cap);
      }
      d->block_ptr[d->num_blocks] = p;
      d->block_len[d->num_blocks] = (int)n;
      ++d->num_blocks;
      p += n;
    }
  }
  return 1;
}

static void hdec_free(hdec *d) {
  free((void *)d->block_ptr);
  free(d->block_len);
}

/* huffman_dec.cpp:274-418 */
static int uncompress_stream(const hdec *d, uint8_t *out, int out_size,
                             const uint8_t *sp, const uint8_t *send,
                             const uint8_t *hard_end) {
  /* huffman_dec.cpp:277-278: tests the OBJECT's stream, not the block. */
  if (at_the_end(d->stream.p, d->stream.bit, d->stream.end)) return out_size == 0;

  const uint8_t *p = sp;
  int bit = 0;
  uint8_t *buf = out;
  uint8_t *const buf_end = out + out_size;
  uint8_t *const buf_fast_end = buf_end - 6;

  while (buf < buf_fast_end) {
    unsigned peek = ((rd(0, p + 1, hard_end) << 8 | rd(0, p, hard_end)) >> bit) & 0xff;
    int nb = bit + d->lut_bits[peek];
    bit = nb & 7;
    p += nb >> 3;
    int symbol;
    if (d->lut_node[peek] < 0) {
      symbol = d->lut_symbol[peek];
    } else {
      int node = d->lut_node[peek];
      while (d->nodes[node].symbol < 0) {
        int b = (rd(0, p, hard_end) >> bit) & 1;
        bit = (bit + 1) & 7;
        if (!bit) ++p;
        node = b ? d->nodes[node].child_b : d->nodes[node].child_a;
      }
      symbol = d->nodes[node].symbol;
    }
    if (symbol <= 255) {
      *buf++ = (uint8_t)symbol;
    } else {
      int nbits, base;
      switch (symbol) {
        case kSymTwoZeros: nbits = 0; base = 2; break;
        case kSymUpTo6Zeros: nbits = 2; base = 3; break;
        case kSymUpTo22Zeros: nbits = 4; base = 7; break;
        case kSymUpTo278Zeros: nbits = 8; base = 23; break;
        case kSymUpTo16662Zeros: nbits = 14; base = 279; break;
        default: return 0;
      }
      uint32_t x = 0;
      for (int i = 0; i < nbits; ++i) {
        x |= (uint32_t)((rd(0, p, hard_end) >> bit) & 1) << i;
        bit = (bit + 1) & 7;
        if (!bit) ++p;
      }
      int zero_count = (int)x + base;
      if (buf + zero_count > buf_end) return 0;
      memset(buf, 0, (size_t)zero_count);
      buf += zero_count;
    }
    if (p > hard_end) return 0; /* hostile input guard; never hit on valid streams */
  }

  bitr r = {p, send, bit, 0};
  while (buf < buf_end) {
    int node = d->root;
    if (d->one_bit_leaf) { (void)read_bit_checked(&r); if (r.failed) return 0; }
    while (d->nodes[node].symbol < 0) {
      int b = read_bit_checked(&r);
      if (r.failed) return 0;
      node = b ? d->nodes[node].child_b : d->nodes[node].child_a;
    }
    int symbol = d->nodes[node].symbol;
    if (symbol <= 255) {
      *buf++ = (uint8_t)symbol;
    } else {
      int nbits, base;
      switch (symbol) {
        case kSymTwoZeros: nbits = 0; base = 2; break;
        case kSymUpTo6Zeros: nbits = 2; base = 3; break;
        case kSymUpTo22Zeros: nbits = 4; base = 7; break;
        case kSymUpTo278Zeros: nbits = 8; base = 23; break;
        case kSymUpTo16662Zeros: nbits = 14; base = 279; break;
        default: return 0;
      }
      int zero_count = (nbits ? (int)read_bits_checked(&r, nbits) : 0) + base;
      if (r.failed || buf + zero_count > buf_end) return 0;
      memset(buf, 0, (size_t)zero_count);
      buf += zero_count;
    }
  }
  return at_the_end(r.p, r.bit, send);
}

/* ------------------------------------------------------------------------ */
/* Decoder  (decoder.cpp)                                                    */
/* ------------------------------------------------------------------------ */

typedef struct {
  const uint8_t *data;
  int size, idx;
} riff;

/* decoder.cpp:428-461 */
static int find_chunk(riff *r, const char *tag, int *size) {
  for (;;) {
    if (r->idx + 8 > r->size) return 0;
    const uint8_t *p = r->data + r->idx;
    int match = memcmp(p, tag, 4) == 0;
    int sz = (int)((uint32_t)p[4] | ((uint32_t)p[5] << 8) | ((uint32_t)p[6] << 16) |
                   ((uint32_t)p[7] << 24));
    r->idx += 8;
    if (sz < 0 || (long)r->idx + sz > r->size) return 0;
    if (match) {
      *size = sz;
      return 1;
    }
    r->idx += sz;
  }
}

typedef struct {
  const hdec *hd;
  const uint8_t *hard_end;
  uint8_t *out, *fres_sym;
  const uint8_t *low;
  int width, height, channels, rows, cols, ycbcr;
  const uint8_t *shift_l, *shift_c;
  const int16_t *fmap;
  volatile int next_row, failed;
} rowjob;

/* decoder.cpp:331-426 */
static int decode_block_row(rowjob *j, int y) {
  const int cols = j->cols, C = j->channels, v = y >> 3;
  const int bh = j->height - y < 8 ? j->height - y : 8;
  const int row_size = cols * C * 64;
  uint8_t *sym = j->fres_sym ? j->fres_sym + (size_t)v * row_size
                             : (uint8_t *)malloc((size_t)row_size);
  int ok = 0;
  /* UncompressBlock, huffman_dec.cpp:261-272 */
  if (j->hd->root >= 0 && j->hd->use_blocks && v < j->hd->num_blocks)
    ok = uncompress_stream(j->hd, sym, row_size, j->hd->block_ptr[v],
                           j->hd->block_ptr[v] + j->hd->block_len[v], j->hard_end);
  else if (g_fix_t2 && j->hd->root >= 0 && !j->hd->use_blocks && v == 0)
    /* fixed mode, one block row: the encoder wrote the payload without a header */
    ok = uncompress_stream(j->hd, sym, row_size, j->hd->stream.p, j->hd->stream.end, j->hard_end);
  if (ok) {
    for (int c = 0; c < C; ++c) {
      const uint8_t *m = j->low + (size_t)c * j->rows * cols;
      const uint8_t *shift = (j->ycbcr && (c == 1 || c == 2)) ? j->shift_c : j->shift_l;
      for (int x = 0; x < j->width; x += 8) {
        int u = x >> 3;
        int bw = j->width - x < 8 ? j->width - x : 8;
        uint8_t packed[64];
        int16_t buf1[64], buf0[64], lowres[64];
        const uint8_t *src = sym + (size_t)c * cols * 64 + u;
        for (int i = 0; i < 64; ++i) packed[kIndexLUT[i]] = src[(size_t)i * cols];
        quantize_unpack(buf1, packed, shift, j->fmap);
        himg_oracle_hadamard_inverse(buf0, buf1);
        get_lowres_block(m, j->rows, cols, lowres, u, v);
        /* RestoreChannelBlock, decoder.cpp:36-75 (W%8==0 path; partial-width
         * tiles are undefined in the reference, trap T9 -- clip here). */
        for (int yy = 0; yy < bh; ++yy)
          for (int xx = 0; xx < bw; ++xx)
            j->out[((size_t)(y + yy) * j->width + x + xx) * C + c] =
                clamp8((int16_t)(buf0[yy * 8 + xx] + lowres[yy * 8 + xx]));
      }
    }
    if (j->ycbcr && C >= 3)
      ycbcr_to_rgb(j->out + (size_t)y * j->width * C, j->width, bh, C);
  }
  if (!j->fres_sym) free(sym);
  return ok;
}

static void *row_worker(void *arg) { /* decoder.cpp:298-309 */
  rowjob *j = (rowjob *)arg;
  for (;;) {
    int y = __sync_fetch_and_add(&j->next_row, 8);
    if (y >= j->height) break;
    if (!decode_block_row(j, y)) {
      j->failed = 1;
      break;
    }
  }
  return NULL;
}

static int decode_impl(const uint8_t *packed, int packed_size, int max_threads,
                       uint8_t **out, int *width, int *height, int *channels,
                       uint8_t **lres_sym_out, int *lres_sym_size,
                       uint8_t **fres_sym_out, int *fres_sym_size,
                       uint8_t **lowres_out) {
  riff r = {packed, packed_size, 0};
  /* decoder.cpp:144-166 */
  if (packed_size < 12 || memcmp(packed, "RIFF", 4) != 0) return -1;
  int file_size = (int)((uint32_t)packed[4] | ((uint32_t)packed[5] << 8) |
                        ((uint32_t)packed[6] << 16) | ((uint32_t)packed[7] << 24));
  if (file_size + 8 != packed_size) return -1;
  if (memcmp(packed + 8, "HIMG", 4) != 0) return -1;
  r.idx = 12;

  int sz;
  /* decoder.cpp:168-200 */
  if (!find_chunk(&r, "FRMT", &sz)) return -2;
  const uint8_t *h = packed + r.idx;
  r.idx += sz;
  if (sz < 11 || h[0] != 1) return -2;
  const int W = (int)((uint32_t)h[1] | ((uint32_t)h[2] << 8) | ((uint32_t)h[3] << 16) | ((uint32_t)h[4] << 24));
  const int H = (int)((uint32_t)h[5] | ((uint32_t)h[6] << 8) | ((uint32_t)h[7] << 16) | ((uint32_t)h[8] << 24));
  const int C = h[9];
  const int use_ycbcr = h[10] != 0;
  const int has_chroma = use_ycbcr && C >= 3;
  if (W < 1 || H < 1 || C < 1) return -2;
  const int rows = (H + 7) >> 3, cols = (W + 7) >> 3;

  /* decoder.cpp:202-212 */
  int16_t lmap[128], fmap[128];
  if (!find_chunk(&r, "LMAP", &sz)) return -3;
  if (!get_mapping_function(lmap, packed + r.idx, sz)) return -3;
  r.idx += sz;

  /* decoder.cpp:214-248 */
  if (!find_chunk(&r, "LRES", &sz)) return -4;
  const int chan_size = block_data_size_per_channel(rows, cols);
  const int lres_size = chan_size * C;
  uint8_t *lres = (uint8_t *)malloc((size_t)lres_size + 8);
  {
    hdec d;
    int ok = hdec_init(&d, packed + r.idx, sz, 0, 0);
    /* Uncompress, huffman_dec.cpp:253-259 */
    if (ok) ok = d.root >= 0 && !d.use_blocks;
    if (ok)
      ok = uncompress_stream(&d, lres, lres_size, d.stream.p, d.stream.end,
                             packed + packed_size);
    hdec_free(&d);
    if (!ok) {
      free(lres);
      return -4;
    }
  }
  r.idx += sz;
  uint8_t *low = (uint8_t *)malloc((size_t)rows * cols * C);
  for (int c = 0; c < C; ++c)
    set_block_data(low + (size_t)c * rows * cols, lres + (size_t)c * chan_size, rows, cols, lmap);

  /* decoder.cpp:250-260 + quantize.cpp:190-213 */
  uint8_t shift_l[64], shift_c[64];
  memset(shift_c, 0, 64);
  if (!find_chunk(&r, "QCFG", &sz) || sz != (has_chroma ? 64 : 32)) {
    free(lres); free(low);
    return -5;
  }
  for (int i = 0; i < 32; ++i) {
    uint8_t x = packed[r.idx + i];
    shift_l[i * 2] = x >> 4;
    shift_l[i * 2 + 1] = x & 15;
  }
  if (has_chroma)
    for (int i = 0; i < 32; ++i) {
      uint8_t x = packed[r.idx + 32 + i];
      shift_c[i * 2] = x >> 4;
      shift_c[i * 2 + 1] = x & 15;
    }
  r.idx += sz;

  /* decoder.cpp:262-272 */
  if (!find_chunk(&r, "FMAP", &sz) || !get_mapping_function(fmap, packed + r.idx, sz)) {
    free(lres); free(low);
    return -6;
  }
  r.idx += sz;

  /* decoder.cpp:274-329 */
  if (!find_chunk(&r, "FRES", &sz)) {
    free(lres); free(low);
    return -7;
  }
  hdec d;
  if (!hdec_init(&d, packed + r.idx, sz, cols * 64 * C, (long)rows * cols * 64 * C)) {
    hdec_free(&d);
    free(lres); free(low);
    return -7;
  }
  uint8_t *pix = (uint8_t *)malloc((size_t)W * H * C);
  uint8_t *fres = fres_sym_out ? (uint8_t *)malloc((size_t)rows * cols * 64 * C) : NULL;
  /* Note: m_use_ycbcr (not HasChroma) selects the chroma table at
   * decoder.cpp:376; for C<3 the chroma table is unset there.  The encoder
   * never produces colourspace=1 with C<3, so use has_chroma. */
  rowjob j = {&d, packed + packed_size, pix, fres, low, W, H, C, rows, cols,
              has_chroma, shift_l, shift_c, fmap, 0, 0};
  int threads = max_threads > 0 ? max_threads : (int)sysconf(_SC_NPROCESSORS_ONLN);
  if (threads > rows) threads = rows;
  if (threads < 1) threads = 1;
  pthread_t tid[256];
  if (threads > 256) threads = 256;
  for (int i = 0; i < threads - 1; ++i) pthread_create(&tid[i], NULL, row_worker, &j);
  row_worker(&j);
  for (int i = 0; i < threads - 1; ++i) pthread_join(tid[i], NULL);
  hdec_free(&d);

  if (j.failed) {
    free(lres); free(low); free(pix); free(fres);
    return -7;
  }
  *out = pix;
  *width = W; *height = H; *channels = C;
  if (lres_sym_out) { *lres_sym_out = lres; *lres_sym_size = lres_size; } else free(lres);
  if (fres_sym_out) { *fres_sym_out = fres; *fres_sym_size = rows * cols * 64 * C; }
  if (lowres_out) *lowres_out = low; else free(low);
  return 0;
}

int himg_oracle_decode(const uint8_t *packed, int packed_size, int max_threads,
                       uint8_t **out, int *width, int *height, int *channels) {
  return decode_impl(packed, packed_size, max_threads, out, width, height, channels,
                     NULL, NULL, NULL, NULL, NULL);
}

int himg_oracle_decode_trace(const uint8_t *packed, int packed_size,
                             uint8_t **out, int *width, int *height,
                             int *channels, uint8_t **lres_sym,
                             int *lres_sym_size, uint8_t **fres_sym,
                             int *fres_sym_size, uint8_t **lowres) {
  return decode_impl(packed, packed_size, 1, out, width, height, channels,
                     lres_sym, lres_sym_size, fres_sym, fres_sym_size, lowres);
}

void himg_oracle_free(void *p) { free(p); }

void himg_oracle_trace_free(himg_oracle_trace *t) {
  if (!t) return;
  free(t->lifted); free(t->avg); free(t->lowres); free(t->lres_sym);
  free(t->fres_sym); free(t->fres_row_bytes);
  memset(t, 0, sizeof(*t));
}
