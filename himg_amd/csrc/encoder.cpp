// encoder.cpp -- himg::Encoder over the C ABI (see include/encoder.h).
#include "encoder.h"

#include <cstring>
#include <iostream>

#include "himg_hip.h"

namespace himg {

namespace {
// Size field of chunk `tag` in a finished stream (the reference prints the
// packed sizes of LRES and FRES, encoder.cpp:219,334).
long ChunkSize(const std::vector<uint8_t> &s, const char tag[4]) {
  size_t idx = 12;
  while (idx + 8 <= s.size()) {
    const uint32_t sz = s[idx + 4] | (s[idx + 5] << 8) | (s[idx + 6] << 16) |
                        (static_cast<uint32_t>(s[idx + 7]) << 24);
    if (std::memcmp(&s[idx], tag, 4) == 0) return static_cast<long>(sz);
    idx += 8 + sz;
  }
  return -1;
}
}  // namespace

Encoder::Encoder() : m_ctx(nullptr) {}

Encoder::~Encoder() {
  if (m_ctx) himg_hip_destroy(m_ctx);
}

bool Encoder::Encode(const uint8_t *data, int width, int height, int pixel_stride,
                     int num_channels, int quality, bool use_ycbcr) {
  m_packed_data.clear();
  if (!m_ctx && himg_hip_create(0, &m_ctx) != HIMG_OK) {
    std::cout << "Error: no usable MI355X device (the HIMG engine has no CPU fallback).\n";
    return false;
  }
  uint8_t *out = nullptr;
  size_t n = 0;
  const int rc = himg_hip_encode(m_ctx, data, width, height, pixel_stride, num_channels,
                                 quality, use_ycbcr ? 1 : 0, &out, &n);
  if (rc != HIMG_OK) {
    std::cout << "Error: " << himg_hip_last_error(m_ctx) << "\n";
    return false;
  }
  m_packed_data.assign(out, out + n);
  himg_hip_free(out);
  std::cout << "Low resolution data: " << ChunkSize(m_packed_data, "LRES") << " bytes.\n";
  std::cout << "Full resolution data: " << ChunkSize(m_packed_data, "FRES") << " bytes.\n";
  return true;
}

}  // namespace himg
