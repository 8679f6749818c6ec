// decoder.cpp -- himg::Decoder over the C ABI (see include/decoder.h).
#include "decoder.h"

#include <cstring>
#include <iostream>
#include <new>

#include "ctx_pool.h"
#include "himg_hip.h"

namespace himg {

Decoder::Decoder(int max_threads)
    : m_ctx(nullptr), m_multi(nullptr), m_max_threads(max_threads), m_unpacked_size(0), m_capacity(0),
      m_width(0), m_height(0), m_num_channels(0) {}

Decoder::~Decoder() {
  detail::release_ctx(m_ctx);
  detail::release_multi(m_multi);
}

Decoder::Decoder(const Decoder &other)
    : m_ctx(nullptr), m_multi(nullptr), m_max_threads(other.m_max_threads), m_unpacked_size(0), m_capacity(0),
      m_width(0), m_height(0), m_num_channels(0) {
  *this = other;
}

Decoder &Decoder::operator=(const Decoder &other) {
  if (this != &other) {
    m_max_threads = other.m_max_threads;
    if (other.m_unpacked_size > m_capacity) {
      m_unpacked_data.reset(new uint8_t[other.m_unpacked_size]);
      m_capacity = other.m_unpacked_size;
    }
    m_unpacked_size = other.m_unpacked_size;
    if (m_unpacked_size) std::memcpy(m_unpacked_data.get(), other.m_unpacked_data.get(), m_unpacked_size);
    m_width = other.m_width;
    m_height = other.m_height;
    m_num_channels = other.m_num_channels;
  }
  return *this;
}

bool Decoder::Decode(const uint8_t *packed_data, int packed_size) {
  m_unpacked_size = 0;
  if (detail::use_multi()) {
    // Several devices (HIMG_DEVICES): block rows sharded over them.
    if (!m_multi) m_multi = detail::acquire_multi();
    if (!m_multi) {
      std::cout << "Error: no usable MI355X devices (the HIMG engine has no CPU fallback).\n";
      return false;
    }
    uint8_t *out = nullptr;
    int w = 0, h = 0, c = 0;
    const int rc = himg_hip_multi_decode(m_multi, packed_data, packed_size < 0 ? 0 : static_cast<size_t>(packed_size),
                                         &out, &w, &h, &c);
    if (rc != HIMG_OK) {
      const char *msg = himg_hip_multi_last_error(m_multi);
      std::cout << msg;
      if (!*msg || msg[std::char_traits<char>::length(msg) - 1] != '\n') std::cout << "\n";
      return false;
    }
    const size_t need = static_cast<size_t>(w) * h * c;
    if (need > m_capacity) {
      m_unpacked_data.reset(new (std::nothrow) uint8_t[need]);
      m_capacity = m_unpacked_data ? need : 0;
      if (!m_unpacked_data) {
        himg_hip_free(out);
        std::cout << "Error: out of memory for a " << w << "x" << h << "x" << c << " image.\n";
        return false;
      }
    }
    std::memcpy(m_unpacked_data.get(), out, need);
    himg_hip_free(out);
    m_width = w;
    m_height = h;
    m_num_channels = c;
    m_unpacked_size = need;
    return true;
  }
  if (!m_ctx) m_ctx = detail::acquire_ctx();
  if (!m_ctx) {
    std::cout << "Error: no usable MI355X device (the HIMG engine has no CPU fallback).\n";
    return false;
  }
  const size_t size = packed_size < 0 ? 0 : static_cast<size_t>(packed_size);
  int w = 0, h = 0, c = 0;
  // Size the (reused) output buffer from the header, then decode straight into it.
  if (himg_hip_peek(packed_data, size, &w, &h, &c) == HIMG_OK) {
    const size_t need = static_cast<size_t>(w) * h * c;
    if (need > m_capacity) {
      // Decode returns false, it never throws (decoder.cpp:87-138).
      m_unpacked_data.reset(new (std::nothrow) uint8_t[need]);
      m_capacity = m_unpacked_data ? need : 0;
      if (!m_unpacked_data) {
        std::cout << "Error: out of memory for a " << w << "x" << h << "x" << c << " image.\n";
        return false;
      }
    }
  }
  const int rc = himg_hip_decode_to(m_ctx, packed_data, size, m_unpacked_data.get(), m_capacity, &w, &h, &c);
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
  m_unpacked_size = static_cast<size_t>(w) * h * c;
  return true;
}

}  // namespace himg
