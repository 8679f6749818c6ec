/*
 * himg_synth.c -- deterministic synthetic RGBA frame generators + FNV-1a-64.
 *
 * Host-side utility of the product library (bench inputs, test inputs).  The
 * generators are the ones SURVEY.md Appendix C.1 defines, so that hashes of
 * the produced frames and of their encodings can be compared with the golden
 * values recorded from the real reference (tests/golden/golden.json).
 */
#include "himg_hip.h"

static inline uint64_t xs(uint64_t *s) {
  uint64_t x = *s;
  x ^= x << 13;
  x ^= x >> 7;
  x ^= x << 17;
  *s = x;
  return x;
}

static inline uint8_t clamp_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v); }

int himg_synth_fill(int kind, uint64_t seed, int width, int height, uint8_t *rgba) {
  if (!rgba || width < 1 || height < 1) return HIMG_ERR_ARG;
  uint64_t S = 0x9E3779B97F4A7C15ull ^ (seed * 0xD1B54A32D192ED03ull);
  if (S == 0) S = 1;
  uint8_t *p = rgba;
  for (int y = 0; y < height; ++y) {
    for (int x = 0; x < width; ++x, p += 4) {
      switch (kind) {
        case HIMG_SYNTH_GRAD:
        case HIMG_SYNTH_GRADN: {
          int base[4];
          base[0] = width > 1 ? x * 255 / (width - 1) : 0;
          base[1] = height > 1 ? y * 255 / (height - 1) : 0;
          base[2] = (width + height > 2) ? (x + y) * 255 / (width + height - 2) : 0;
          base[3] = 255;
          for (int c = 0; c < 4; ++c) {
            int n = 0;
            if (kind == HIMG_SYNTH_GRADN) n = (int)((xs(&S) >> 40) & 15) - 8;
            p[c] = clamp_u8(base[c] + n);
          }
          break;
        }
        case HIMG_SYNTH_RAND:
          for (int c = 0; c < 4; ++c) p[c] = (uint8_t)((xs(&S) >> 32) & 255);
          break;
        case HIMG_SYNTH_RANDTILE: {
          uint64_t t = ((uint64_t)(y / 8) * 1315423911ull + (uint64_t)(x / 8)) *
                           2654435761ull + 12345ull + seed;
          for (int c = 0; c < 4; ++c) {
            int base = (int)((xs(&t) >> 32) & 255);
            int n = (int)((xs(&S) >> 40) & 15) - 8;
            p[c] = clamp_u8(base + n);
          }
          break;
        }
        default:
          return HIMG_ERR_ARG;
      }
    }
  }
  return HIMG_OK;
}

uint64_t himg_fnv1a64(const uint8_t *data, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) {
    h ^= data[i];
    h *= 1099511628211ull;
  }
  return h;
}
