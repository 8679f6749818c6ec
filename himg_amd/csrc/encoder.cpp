// encoder.cpp -- himg::Encoder over the C ABI (see include/encoder.h).
#include "encoder.h"

#include <cstring>
#include <iostream>

#include "ctx_pool.h"
#include "himg_hip.h"

namespace himg {

namespace {
// Size field of chunk `tag` in a finished stream (the reference prints the
// packed sizes of LRES and FRES, encoder.cpp:219,334).
long ChunkSize(const uint8_t *s, size_t size, const char tag[4]) {
  size_t idx = 12;
  while (idx + 8 <= size) {
    const uint32_t sz = s[idx + 4] | (s[idx + 5] << 8) | (s[idx + 6] << 16) |
                        (static_cast<uint32_t>(s[idx + 7]) << 24);
    if (std::memcmp(&s[idx], tag, 4) == 0) return static_cast<long>(sz);
    idx += 8 + sz;
  }
  return -1;
}
}  // namespace

Encoder::Encoder() : m_ctx(nullptr), m_multi(nullptr), m_packed_size(0) {}

Encoder::~Encoder() {
  detail::release_ctx(m_ctx);
  detail::release_multi(m_multi);
}

Encoder::Encoder(const Encoder &other) : m_ctx(nullptr), m_multi(nullptr), m_packed_size(0) { *this = other; }

Encoder &Encoder::operator=(const Encoder &other) {
  if (this != &other) {
    m_packed_data.reset(other.m_packed_size ? new uint8_t[other.m_packed_size] : nullptr);
    m_packed_size = other.m_packed_size;
    if (m_packed_size) std::memcpy(m_packed_data.get(), other.m_packed_data.get(), m_packed_size);
  }
  return *this;
}

bool Encoder::Encode(const uint8_t *data, int width, int height, int pixel_stride,
                     int num_channels, int quality, bool use_ycbcr) {
  m_packed_data.reset();
  m_packed_size = 0;
  if (detail::use_multi()) {
    // Several devices (HIMG_DEVICES): one frame, block rows sharded over them.
    if (!m_multi) m_multi = detail::acquire_multi();
    if (!m_multi) {
      std::cout << "Error: no usable MI355X devices (the HIMG engine has no CPU fallback).\n";
      return false;
    }
    uint8_t *out = nullptr;
    size_t n = 0;
    const int rc = himg_hip_multi_encode(m_multi, data, width, height, pixel_stride, num_channels, quality,
                                         use_ycbcr ? 1 : 0, &out, &n);
    if (rc != HIMG_OK) {
      std::cout << "Error: " << himg_hip_multi_last_error(m_multi) << "\n";
      return false;
    }
    m_packed_data.reset(new uint8_t[n]);
    std::memcpy(m_packed_data.get(), out, n);
    himg_hip_free(out);
    m_packed_size = n;
    std::cout << "Low resolution data: " << ChunkSize(m_packed_data.get(), n, "LRES") << " bytes.\n";
    std::cout << "Full resolution data: " << ChunkSize(m_packed_data.get(), n, "FRES") << " bytes.\n";
    return true;
  }
  if (!m_ctx) m_ctx = detail::acquire_ctx();
  if (!m_ctx) {
    std::cout << "Error: no usable MI355X device (the HIMG engine has no CPU fallback).\n";
    return false;
  }
  // Encode on the device, learn the size, then fetch exactly that many bytes into
  // storage of exactly that size (no zero fill, no second copy).
  size_t n = 0;
  int rc = himg_hip_encode_to(m_ctx, data, width, height, pixel_stride, num_channels, quality,
                              use_ycbcr ? 1 : 0, nullptr, 0, &n);
  if (rc == HIMG_ERR_CAPACITY && n > 0) {
    m_packed_data.reset(new uint8_t[n]);
    rc = himg_hip_fetch_last(m_ctx, m_packed_data.get(), n, &n);
  }
  if (rc != HIMG_OK) {
    std::cout << "Error: " << himg_hip_last_error(m_ctx) << "\n";
    m_packed_data.reset();
    return false;
  }
  m_packed_size = n;
  std::cout << "Low resolution data: " << ChunkSize(m_packed_data.get(), n, "LRES") << " bytes.\n";
  std::cout << "Full resolution data: " << ChunkSize(m_packed_data.get(), n, "FRES") << " bytes.\n";
  return true;
}

}  // namespace himg
