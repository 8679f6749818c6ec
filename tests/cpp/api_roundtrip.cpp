// api_roundtrip.cpp -- exercises himg::Encoder / himg::Decoder exactly the way
// the reference's callers do (src/chimg.cpp:140-163, src/dhimg.cpp:45-65,
// src/benchmark.cpp:108-125): construct, Encode, packed_data()/packed_size(),
// one Decoder reused for several Decode calls, unpacked_data()/width()/...
//
//   api_roundtrip <in.rgba> <width> <height> <quality> <out.himg> <out.rgba>
#include <cstdint>
#include <fstream>
#include <iostream>
#include <vector>

#include "decoder.h"
#include "encoder.h"

int main(int argc, const char **argv) {
  if (argc != 7) return 2;
  const int w = std::stoi(argv[2]), h = std::stoi(argv[3]), q = std::stoi(argv[4]);
  std::vector<uint8_t> img(static_cast<size_t>(w) * h * 4);
  std::ifstream(argv[1], std::ios::binary).read(reinterpret_cast<char *>(img.data()), img.size());

  himg::Encoder encoder;
  if (!encoder.Encode(img.data(), w, h, 4, 4, q, true)) return 3;
  std::cout << "Compressed size: " << encoder.packed_size() << "\n";  // chimg.cpp:153
  std::ofstream(argv[5], std::ios::binary)
      .write(reinterpret_cast<const char *>(encoder.packed_data()), encoder.packed_size());

  himg::Decoder decoder;  // max_threads defaults to 0 like the reference
  for (int iteration = 0; iteration < 3; ++iteration) {  // reusable (benchmark.cpp:111-125)
    if (!decoder.Decode(encoder.packed_data(), encoder.packed_size())) {
      std::cout << "Unable to decode image." << std::endl;
      return 4;
    }
  }
  std::cout << "Decoded " << decoder.width() << "x" << decoder.height() << "x"
            << decoder.num_channels() << " " << decoder.unpacked_size() << "\n";
  std::ofstream(argv[6], std::ios::binary)
      .write(reinterpret_cast<const char *>(decoder.unpacked_data()), decoder.unpacked_size());

  // A stream the reference refuses (truncated) must be refused with its message.
  std::vector<uint8_t> bad(encoder.packed_data(), encoder.packed_data() + 40);
  if (decoder.Decode(bad.data(), static_cast<int>(bad.size()))) return 5;
  return 0;
}
