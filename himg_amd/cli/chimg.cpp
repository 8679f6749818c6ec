// chimg -- compress an image to .himg on the MI355X engine.
//
// Same command line, messages and exit codes as the reference tool
// (src/chimg.cpp:36-169): "chimg [-q N] [-rgb] image outfile", exit 0 on bad
// arguments, -1 on I/O failure.  Input: binary PGM / PPM / PAM (see pnm_io.h).
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "encoder.h"
#include "pnm_io.h"

namespace {

const int kDefaultQuality = 50;  // src/chimg.cpp:20

bool to_int(const char *arg, int *value) {
  char *end = nullptr;
  const long v = strtol(arg, &end, 10);
  if (end != arg && *end == '\0') { *value = (int)v; return true; }
  std::cout << "Invalid integer expression: " << arg << "\n";
  return false;
}

}  // namespace

int main(int argc, const char **argv) {
  int quality = kDefaultQuality;
  bool use_ycbcr = true, ok = true;
  std::vector<const char *> files;
  for (int k = 1; k < argc && ok; ++k) {
    const std::string arg = argv[k];
    if (!arg.empty() && arg[0] == '-') {
      if (arg == "-q") {
        if (k + 1 < argc && to_int(argv[++k], &quality)) {
          if (quality < 0 || quality > 100) {
            std::cout << "Invalid quality level: " << quality << "\n";
            ok = false;
          }
        } else {
          ok = false;
        }
      } else if (arg == "-rgb") {
        use_ycbcr = false;
      } else {
        std::cout << "Invalid option: " << arg << "\n";
        ok = false;
      }
    } else {
      files.push_back(argv[k]);
    }
  }
  if (!ok || files.size() != 2) {
    std::cout << "Usage: " << argv[0] << " [options] image outfile\n";
    std::cout << "Options:\n";
    std::cout << " -q <quality> Set the quality (0-100)\n";
    std::cout << " -rgb         Use RGB color space (instead of YCbCr)\n";
    return 0;
  }

  pnm::Image img;
  const int rc = pnm::read(files[0], &img);
  if (rc == 2) { std::cerr << "Unknown file format for " << files[0] << std::endl; return -1; }
  if (rc != 0) { std::cerr << "Unable to load " << files[0] << std::endl; return -1; }

  // FreeImage memory convention (bottom-up, BGR(A)), so that the stream equals
  // what the reference chimg makes of the same picture.
  std::vector<uint8_t> pixels(img.data.size());
  pnm::flip_and_swap(img.data.data(), pixels.data(), img.width, img.height, img.channels);

  himg::Encoder encoder;
  if (!encoder.Encode(pixels.data(), img.width, img.height, img.channels, img.channels, quality, use_ycbcr)) {
    std::cerr << "Unable to encode " << files[0] << std::endl;  // no GPU engine: there is no CPU fallback
    return -1;
  }
  std::cout << "Compressed size: " << encoder.packed_size() << std::endl;

  std::ofstream f(files[1], std::ofstream::out | std::ofstream::binary);
  f.write(reinterpret_cast<const char *>(encoder.packed_data()), encoder.packed_size());
  return f.good() ? 0 : -1;
}
