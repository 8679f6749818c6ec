// decoder.h -- himg::Decoder, source compatible with the reference's public
// surface (reference src/lib/decoder.h:22-67), backed by the MI355X engine.
//
// Callers written against the reference (src/dhimg.cpp:45-65,
// src/benchmark.cpp:108-125) compile unchanged.
#ifndef DECODER_H_
#define DECODER_H_

#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>  // the reference's headers pull it in (encoder.h:13, decoder.h:13) and its callers rely on that

struct himg_hip_ctx;
struct himg_hip_multi;

namespace himg {

class Decoder {
 public:
  // max_threads is accepted for source compatibility (decoder.cpp:79-85); the
  // block rows are decoded in parallel on the GPU regardless of its value.
  Decoder(int max_threads = 0);
  ~Decoder();
  // Copyable like the reference's class (decoder.h:22-67): a copy holds a copy of the
  // unpacked picture; engine contexts are borrowed per object.
  Decoder(const Decoder &other);
  Decoder &operator=(const Decoder &other);

  // Same contract as the reference (decoder.cpp:87-138): the input is borrowed
  // for the call, the output is owned by the object until the next Decode.
  // Returns false and prints the reference's messages to std::cout on exactly
  // the streams the reference rejects.  Reusable across calls.
  bool Decode(const uint8_t *packed_data, int packed_size);

  const uint8_t *unpacked_data() const { return m_unpacked_data.get(); }
  int unpacked_size() const { return static_cast<int>(m_unpacked_size); }

  int width() const { return m_width; }
  int height() const { return m_height; }
  int num_channels() const { return m_num_channels; }

 private:
  himg_hip_ctx *m_ctx;
  himg_hip_multi *m_multi;   // HIMG_DEVICES names several devices
  int m_max_threads;
  // Kept (and its pages kept mapped) across Decode calls: the copy from the GPU
  // then runs at PCIe speed instead of page-faulting through a fresh allocation.
  std::unique_ptr<uint8_t[]> m_unpacked_data;
  size_t m_unpacked_size, m_capacity;
  int m_width;
  int m_height;
  int m_num_channels;
};

}  // namespace himg

#endif  // DECODER_H_
