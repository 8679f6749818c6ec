/*
 * himg_tables.c -- host-side derivation of the HIMG format tables.
 *
 * These tables parameterise the kernels (shift tables, companding tables and
 * the LUTs derived from them) and are serialised into the QCFG / LMAP / FMAP
 * chunks.  The constant arrays are the format itself;
 * the derivations follow reference quantize.cpp:72-125 and mapper.cpp:75-223.
 */
#include "himg_hip.h"
#include "himg_tables.h"

#include <string.h>

/* quantize.cpp:19-28 (luma) and :31-40 (chroma) base matrices. */
static const uint8_t kLumaBase[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t kChromaBase[64] = {
    17,  18,  24,  47,  100, 110, 115, 120, 18,  21,  26,  66,  100,
    110, 118, 121, 24,  26,  56,  100, 100, 110, 120, 122, 47,  66,
    100, 100, 100, 110, 120, 123, 100, 100, 100, 100, 100, 110, 120,
    124, 110, 110, 110, 110, 110, 110, 110, 123, 120, 120, 120, 120,
    120, 110, 100, 122, 124, 124, 126, 126, 125, 123, 122, 105};

typedef struct {
  int q, s;
} qs_t;

/* quantize.cpp:55-65 */
static const qs_t kQuantScale[] = {{0, 65535}, {10, 32512}, {20, 13568},
                                   {30, 5120}, {40, 2560},  {50, 1024},
                                   {60, 768},  {80, 256},   {100, 0}};
/* mapper.cpp:38-47 */
static const qs_t kLowMapScale[] = {{0, 120}, {5, 90},  {10, 70}, {20, 40},
                                    {30, 32}, {40, 26}, {50, 20}, {100, 16}};

/* mapper.cpp:19-36: identity up to 65, then a widening ramp to 255. */
static const int16_t kLowMapCurve[128] = {
    0,   1,   2,   3,   4,   5,   6,   7,   8,   9,   10,  11,  12,  13,  14,
    15,  16,  17,  18,  19,  20,  21,  22,  23,  24,  25,  26,  27,  28,  29,
    30,  31,  32,  33,  34,  35,  36,  37,  38,  39,  40,  41,  42,  43,  44,
    45,  46,  47,  48,  49,  50,  51,  52,  53,  54,  55,  56,  57,  58,  59,
    60,  61,  62,  63,  64,  65,  67,  68,  70,  71,  73,  74,  76,  78,  79,
    81,  83,  85,  87,  89,  91,  93,  95,  97,  99,  102, 104, 106, 109, 111,
    114, 117, 119, 122, 125, 128, 131, 134, 137, 140, 143, 146, 150, 153, 156,
    160, 164, 167, 171, 175, 178, 182, 186, 190, 195, 199, 203, 207, 212, 216,
    221, 226, 230, 235, 240, 245, 250, 255};

/* mapper.cpp:54-71: identity up to 49, then ~7.5 % steps up to 8039. */
static const int16_t kFullMapCurve[128] = {
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

/* Piecewise-linear quality -> scale, rounding like the reference
 * (quantize.cpp:72-92, mapper.cpp:75-97): C integer division, truncating. */
static int scale_for_quality(int quality, const qs_t *t, int n) {
  int k = 0;
  while (k < n - 1 && t[k + 1].q <= quality) ++k;
  if (k >= n - 1) return t[n - 1].s;
  const int dq = t[k + 1].q - t[k].q;
  return t[k].s + ((t[k + 1].s - t[k].s) * (quality - t[k].q) + (dq >> 1)) / dq;
}

void himg_tables_shift(int quality, int chroma, uint8_t out[64]) {
  const uint8_t *base = chroma ? kChromaBase : kLumaBase;
  /* The reference narrows quality to uint8_t here (quantize.h:21). */
  const int scale = scale_for_quality((int)(uint8_t)quality, kQuantScale,
                                      (int)(sizeof(kQuantScale) / sizeof(kQuantScale[0])));
  for (int i = 0; i < 64; ++i) {
    unsigned c = (unsigned)(((int)base[i] * scale + 512) >> 10) & 0xffffu;
    /* floor(log2) plus the bit just below the MSB (quantize.cpp:94-102). */
    unsigned s = 0;
    if (c > 1) {
      unsigned msb = 31u - (unsigned)__builtin_clz(c);
      s = msb + ((c >> (msb - 1)) & 1u);
    }
    out[i] = (uint8_t)(s > 15 ? 15 : s);
  }
}

void himg_tables_lowres_map(int quality, int16_t out[128]) {
  const int scale = (int16_t)scale_for_quality(
      quality, kLowMapScale, (int)(sizeof(kLowMapScale) / sizeof(kLowMapScale[0])));
  for (int i = 0; i < 128; ++i) {
    int idx = (int16_t)((i * scale + 8) >> 4);
    out[i] = kLowMapCurve[idx > 127 ? 127 : idx];
  }
}

void himg_tables_fullres_map(int16_t out[128]) {
  memcpy(out, kFullMapCurve, sizeof(kFullMapCurve));
}

/* Companding search with the reference's quirks (mapper.cpp:159-182, trap T7):
 * non-zero never maps to 0, >= table[126] always maps to 127, ties go up. */
uint8_t himg_tables_map_to_8bit(const int16_t t[128], int xi) {
  const int16_t x = (int16_t)xi;
  if (x == 0) return 0;
  const int16_t a = (int16_t)(x < 0 ? -x : x); /* wraps for -32768 like std::abs -> int16 */
  int m = 1;
  while (m < 126 && !(a < t[m + 1])) ++m;
  if (m < 126 && (a - t[m]) < (t[m + 1] - a)) --m;
  if (m < 127) ++m;
  return x >= 0 ? (uint8_t)m : (uint8_t)(-(int8_t)m);
}

int himg_tables_unmap(const int16_t t[128], uint8_t code) {
  const int s = (int8_t)code;
  if (s >= 0) return t[s];
  return s == -128 ? -t[127] : -t[-s]; /* mapper.cpp:148-154 */
}

int himg_tables_mapping_function(const int16_t t[128], uint8_t *out) {
  /* mapper.cpp:105-125,184-191 */
  int n1 = 1;
  while (n1 < 128 && t[n1] < 256) ++n1;
  n1 -= 1;
  uint8_t *p = out;
  *p++ = (uint8_t)n1;
  for (int i = 1; i <= 127; ++i) {
    uint16_t x = (uint16_t)t[i];
    *p++ = (uint8_t)(x & 255);
    if (i > n1) *p++ = (uint8_t)(x >> 8);
  }
  return (int)(p - out);
}

int himg_tables_parse_mapping_function(int16_t t[128], const uint8_t *in, int size) {
  /* mapper.cpp:127-157 */
  if (size < 1) return 0;
  const int n1 = in[0];
  if (n1 > 127 || 1 + n1 + 2 * (127 - n1) != size) return 0;
  const uint8_t *p = in + 1;
  t[0] = 0;
  for (int i = 1; i <= 127; ++i) {
    if (i <= n1) {
      t[i] = (int16_t)*p++;
    } else {
      t[i] = (int16_t)(uint16_t)(p[0] | (p[1] << 8));
      p += 2;
    }
  }
  return 1;
}
