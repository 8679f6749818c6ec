// dhimg -- decompress a .himg file on the MI355X engine.
//
// Same command line, messages and exit codes as the reference tool
// (src/dhimg.cpp:17-72): "dhimg image outfile".  Output: binary PGM / PPM / PAM
// by channel count (the reference writes PNG through FreeImage).
#include <fstream>
#include <iostream>
#include <vector>

#include "decoder.h"
#include "pnm_io.h"

int main(int argc, const char **argv) {
  if (argc < 3) {
    std::cout << "Usage: " << argv[0] << " image outfile" << std::endl;
    return 0;
  }
  std::vector<uint8_t> packed;
  {
    std::ifstream f(argv[1], std::ifstream::in | std::ifstream::binary);
    if (!f.good()) {
      std::cout << "Unable to read file " << argv[1] << std::endl;
      return -1;
    }
    f.seekg(0, std::ifstream::end);
    const std::streamoff file_size = f.tellg();
    f.seekg(0, std::ifstream::beg);
    std::cout << "File size: " << file_size << std::endl;
    packed.resize((size_t)file_size);
    f.read(reinterpret_cast<char *>(packed.data()), file_size);
  }

  himg::Decoder decoder;
  if (!decoder.Decode(packed.data(), (int)packed.size())) {
    std::cout << "Unable to decode image." << std::endl;
    return -1;
  }

  pnm::Image img;
  img.width = decoder.width();
  img.height = decoder.height();
  img.channels = decoder.num_channels();
  img.data.resize((size_t)decoder.unpacked_size());
  // Back from FreeImage's bottom-up BGR(A) to top-down RGB(A).
  pnm::flip_and_swap(decoder.unpacked_data(), img.data.data(), img.width, img.height, img.channels);
  if ((img.channels != 1 && img.channels != 3 && img.channels != 4) || !pnm::write(argv[2], img)) {
    std::cout << "Unable to write file " << argv[2] << std::endl;
    return -1;
  }
  return 0;
}
