// benchmark -- the reference's timing tool on the MI355X engine.
//
// Same command line and output as src/benchmark.cpp:49-158: "benchmark [-d][-e]
// image", 30 iterations, "Iteration i/30" lines, then Min / Max / Average in
// milliseconds.  -d (default) times himg::Decoder::Decode of a .himg file with ONE
// Decoder object (benchmark.cpp:108,122); a file that is not HIMG is parsed as
// Netpbm instead (the reference times FreeImage there).  -e is an empty branch in
// the reference (benchmark.cpp:137-138); here it times himg::Encoder::Encode
// (quality 50, YCbCr) of a Netpbm picture, a fresh Encoder per iteration like
// chimg uses it.
#include <chrono>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "decoder.h"
#include "encoder.h"
#include "pnm_io.h"

namespace {

const int kNumIterations = 30;

void usage(const char *arg0) {
  std::cout << "Usage: " << arg0 << " [-d][-e] image" << std::endl;
  std::cout << "  -d Decode (default)" << std::endl;
  std::cout << "  -e Encode" << std::endl;
}

bool is_himg(const std::vector<uint8_t> &b) {
  return b.size() >= 12 && b[0] == 'R' && b[1] == 'I' && b[2] == 'F' && b[3] == 'F' && b[8] == 'H' &&
         b[9] == 'I' && b[10] == 'M' && b[11] == 'G';
}

// The library's progress lines would drown the iteration lines.
struct QuietCout {
  std::streambuf *old;
  QuietCout() : old(std::cout.rdbuf(nullptr)) {}
  ~QuietCout() { std::cout.rdbuf(old); }
};

}  // namespace

int main(int argc, const char **argv) {
  bool encode = false;
  std::string file_name;
  for (int i = 1; i < argc; ++i) {
    const char *arg = argv[i];
    if (arg[0] == '-' && arg[1] != 0 && arg[2] == 0) {
      if (arg[1] == 'd') encode = false;
      else if (arg[1] == 'e') encode = true;
    } else if (file_name.empty()) {
      file_name = arg;
    } else {
      usage(argv[0]);
      return 0;
    }
  }
  if (file_name.empty()) {
    usage(argv[0]);
    return 0;
  }

  std::vector<uint8_t> buffer;
  {
    std::ifstream f(file_name.c_str(), std::ifstream::in | std::ifstream::binary);
    if (!f.good()) {
      std::cout << "Unable to read file " << file_name << std::endl;
    } else {
      f.seekg(0, std::ifstream::end);
      const std::streamoff file_size = f.tellg();
      f.seekg(0, std::ifstream::beg);
      std::cout << "File size: " << file_size << std::endl;
      buffer.resize((size_t)file_size);
      f.read(reinterpret_cast<char *>(buffer.data()), file_size);
    }
  }

  pnm::Image img;
  std::vector<uint8_t> pixels;
  if (encode) {
    if (pnm::read(file_name.c_str(), &img) != 0) {
      std::cout << "Unable to load " << file_name << std::endl;
      return -1;
    }
    pixels.resize(img.data.size());
    pnm::flip_and_swap(img.data.data(), pixels.data(), img.width, img.height, img.channels);
  }

  himg::Decoder himg_decoder;
  double min_dt = -1.0, max_dt = -1.0, total_t = 0.0;
  for (int iteration = 1; iteration <= kNumIterations; ++iteration) {
    std::cout << "Iteration " << iteration << "/" << kNumIterations << std::endl;
    const auto t0 = std::chrono::steady_clock::now();
    bool failed = false;
    {
      QuietCout quiet;
      if (!encode) {
        if (is_himg(buffer)) {
          failed = !himg_decoder.Decode(buffer.data(), (int)buffer.size());
        } else {
          pnm::Image other;
          pnm::read(file_name.c_str(), &other);
        }
      } else {
        himg::Encoder encoder;
        failed = !encoder.Encode(pixels.data(), img.width, img.height, img.channels, img.channels, 50, true);
      }
    }
    if (failed) {
      std::cout << (encode ? "Unable to encode image." : "Unable to decode image.") << std::endl;
      return -1;
    }
    const double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (min_dt < 0.0 || dt < min_dt) min_dt = dt;
    if (max_dt < 0.0 || dt > max_dt) max_dt = dt;
    total_t += dt;
  }
  const double average = total_t / static_cast<double>(kNumIterations);
  std::cout << "    Min: " << min_dt << " ms\n";
  std::cout << "    Max: " << max_dt << " ms\n";
  std::cout << "Average: " << average << " ms\n";
  return 0;
}
