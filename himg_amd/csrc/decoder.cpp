// decoder.cpp -- himg::Decoder over the C ABI (see include/decoder.h).
#include "decoder.h"

#include <iostream>

#include "himg_hip.h"

namespace himg {

Decoder::Decoder(int max_threads)
    : m_ctx(nullptr), m_max_threads(max_threads), m_width(0), m_height(0), m_num_channels(0) {}

Decoder::~Decoder() {
  if (m_ctx) himg_hip_destroy(m_ctx);
}

bool Decoder::Decode(const uint8_t *packed_data, int packed_size) {
  m_unpacked_data.clear();
  if (!m_ctx && himg_hip_create(0, &m_ctx) != HIMG_OK) {
    std::cout << "Error: no usable MI355X device (the HIMG engine has no CPU fallback).\n";
    return false;
  }
  uint8_t *out = nullptr;
  int w = 0, h = 0, c = 0;
  const int rc = himg_hip_decode(m_ctx, packed_data,
                                 packed_size < 0 ? 0 : static_cast<size_t>(packed_size), &out, &w,
                                 &h, &c);
  if (rc != HIMG_OK) {
    // For HIMG_ERR_FORMAT the message is the reference's own text
    // (decoder.cpp:96-135,232,287,345), newline-terminated.
    const char *msg = himg_hip_last_error(m_ctx);
    std::cout << msg;
    if (!*msg || msg[std::char_traits<char>::length(msg) - 1] != '\n') std::cout << "\n";
    return false;
  }
  m_width = w;
  m_height = h;
  m_num_channels = c;
  m_unpacked_data.assign(out, out + static_cast<size_t>(w) * h * c);
  himg_hip_free(out);
  return true;
}

}  // namespace himg
