// ref_harness.cpp -- C-ABI wrapper around the REAL reference classes.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; the reference sources it is
// linked with are compiled where they lie under /root/reference/src/lib by
// oracle/Makefile and nothing of them is copied into this repository.  The
// resulting oracle/_ref/libhimg_ref.so is git-ignored but travels to the GPU
// box, where it pins the oracle and serves as the timed CPU baseline
// (bench.py cpu_baseline.kind == "reference").
//
// Reference interface wrapped: himg::Encoder (src/lib/encoder.h:20-64) and
// himg::Decoder (src/lib/decoder.h:22-67).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <sstream>

#include "decoder.h"
#include "encoder.h"

namespace {
// The library prints progress lines to std::cout from inside Encode/Decode
// (encoder.cpp:219,334; decoder.cpp:96-135).  Swallow them.  Calls may come from
// several threads at once (bench.py times one encoder per core), so the stream
// buffer is swapped under a reference count, not per call.
struct NullBuf : std::streambuf {
  int overflow(int c) override { return c; }
  std::streamsize xsputn(const char *, std::streamsize n) override { return n; }
};
struct QuietCout {
  static std::mutex &mu() { static std::mutex m; return m; }
  static int &users() { static int n = 0; return n; }
  static std::streambuf *&saved() { static std::streambuf *p = nullptr; return p; }
  static NullBuf &sink() { static NullBuf b; return b; }
  QuietCout() {
    std::lock_guard<std::mutex> g(mu());
    if (users()++ == 0) saved() = std::cout.rdbuf(&sink());
  }
  ~QuietCout() {
    std::lock_guard<std::mutex> g(mu());
    if (--users() == 0) std::cout.rdbuf(saved());
  }
};
}  // namespace

extern "C" {

// A fresh Encoder per call: Encoder objects are single-use.
int himg_ref_encode(const uint8_t *data, int width, int height, int pixel_stride,
                    int num_channels, int quality, int use_ycbcr, uint8_t **out,
                    int *out_size) {
  QuietCout q;
  himg::Encoder enc;
  if (!enc.Encode(data, width, height, pixel_stride, num_channels, quality,
                  use_ycbcr != 0))
    return -1;
  *out_size = enc.packed_size();
  *out = static_cast<uint8_t *>(std::malloc(static_cast<size_t>(*out_size)));
  std::memcpy(*out, enc.packed_data(), static_cast<size_t>(*out_size));
  return 0;
}

int himg_ref_decode(const uint8_t *packed, int packed_size, int max_threads,
                    uint8_t **out, int *width, int *height, int *channels) {
  QuietCout q;
  himg::Decoder dec(max_threads);
  if (!dec.Decode(packed, packed_size)) return -1;
  *width = dec.width();
  *height = dec.height();
  *channels = dec.num_channels();
  size_t n = static_cast<size_t>(dec.unpacked_size());
  *out = static_cast<uint8_t *>(std::malloc(n));
  std::memcpy(*out, dec.unpacked_data(), n);
  return 0;
}

// Decode the same buffer `iters` times on ONE Decoder object, like
// src/benchmark.cpp:108-125 does; returns seconds per iteration via *secs.
int himg_ref_decode_loop(const uint8_t *packed, int packed_size, int max_threads,
                         int iters) {
  QuietCout q;
  himg::Decoder dec(max_threads);
  for (int i = 0; i < iters; ++i)
    if (!dec.Decode(packed, packed_size)) return -1;
  return 0;
}

void himg_ref_free(void *p) { std::free(p); }

}  // extern "C"
