// chimg -- compress a picture to .himg on the MI355X engine.
//
// Command line, messages and exit codes follow the reference tool
// (src/chimg.cpp:36-169): "chimg [-q N] [-rgb] image outfile"; exit 0 after the
// usage text, -1 when the input cannot be read or the output cannot be written.
// Input: binary PGM / PPM / PAM (pnm_io.h) instead of the formats FreeImage reads.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "encoder.h"
#include "pnm_io.h"

namespace {

struct Request {
  int quality = 50;        // the reference's default (src/chimg.cpp:22)
  bool ycbcr = true;
  const char *input = nullptr;
  const char *output = nullptr;
};

// Returns false (after the reference's message, if any) when the usage text is due.
bool parse(int argc, const char **argv, Request *rq) {
  int nfiles = 0;
  for (int i = 1; i < argc; ++i) {
    const char *a = argv[i];
    if (a[0] != '-') {
      if (nfiles == 0) rq->input = a;
      else if (nfiles == 1) rq->output = a;
      ++nfiles;
    } else if (!strcmp(a, "-rgb")) {
      rq->ycbcr = false;
    } else if (!strcmp(a, "-q")) {
      if (++i >= argc) return false;
      char *end = nullptr;
      const long v = strtol(argv[i], &end, 10);
      if (end == argv[i] || *end) {
        printf("Invalid integer expression: %s\n", argv[i]);
        return false;
      }
      rq->quality = static_cast<int>(v);
      if (v < 0 || v > 100) {
        printf("Invalid quality level: %d\n", rq->quality);
        return false;
      }
    } else {
      printf("Invalid option: %s\n", a);
      return false;
    }
  }
  return nfiles == 2;
}

}  // namespace

int main(int argc, const char **argv) {
  Request rq;
  if (!parse(argc, argv, &rq)) {
    printf("Usage: %s [options] image outfile\n"
           "Options:\n"
           " -q <quality> Set the quality (0-100)\n"
           " -rgb         Use RGB color space (instead of YCbCr)\n",
           argv[0]);
    return 0;
  }

  pnm::Image picture;
  switch (pnm::read(rq.input, &picture)) {
    case 0: break;
    case 2: fprintf(stderr, "Unknown file format for %s\n", rq.input); return -1;
    default: fprintf(stderr, "Unable to load %s\n", rq.input); return -1;
  }
  // The reference hands the codec FreeImage's memory layout (bottom-up scanlines,
  // BGR(A)); do the same so that both tools make the same stream of one picture.
  std::vector<uint8_t> pixels(picture.data.size());
  pnm::flip_and_swap(picture.data.data(), pixels.data(), picture.width, picture.height, picture.channels);

  fflush(stdout);   // the library reports through std::cout
  himg::Encoder encoder;
  const int c = picture.channels;
  if (!encoder.Encode(pixels.data(), picture.width, picture.height, c, c, rq.quality, rq.ycbcr)) {
    fprintf(stderr, "Unable to encode %s\n", rq.input);   // no GPU engine: there is no CPU fallback
    return -1;
  }
  printf("Compressed size: %d\n", encoder.packed_size());

  FILE *f = fopen(rq.output, "wb");
  const size_t n = static_cast<size_t>(encoder.packed_size());
  const bool ok = f && fwrite(encoder.packed_data(), 1, n, f) == n;
  if (f) fclose(f);
  return ok ? 0 : -1;
}
