// encoder.h -- himg::Encoder, source compatible with the reference's public
// surface (reference src/lib/encoder.h:20-64), backed by the MI355X engine.
//
// Callers written against the reference (src/chimg.cpp:140-163) compile
// unchanged: same header name, namespace, constructor and public methods.  The
// private part is different by design -- everything below the API is the C ABI
// of include/himg_hip.h (hand-written HIP kernels), not a port of src/lib.
#ifndef ENCODER_H_
#define ENCODER_H_

#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>  // the reference's headers pull it in (encoder.h:13, decoder.h:13) and its callers rely on that

struct himg_hip_ctx;
struct himg_hip_multi;

namespace himg {

class Encoder {
 public:
  Encoder();
  ~Encoder();
  // Copyable like the reference's class (its members are plain vectors, encoder.h:36-64):
  // a copy holds a copy of the packed stream; engine contexts are borrowed per object.
  Encoder(const Encoder &other);
  Encoder &operator=(const Encoder &other);

  // Same contract as the reference (encoder.cpp:59-109): `data` is read during
  // the call only; rows are tightly packed width*pixel_stride bytes.  Prints the
  // reference's two progress lines to std::cout (encoder.cpp:219,334).  Unlike
  // the reference, which always returns true, this returns false when the GPU
  // engine reports an error (there is no CPU fallback).  Every call has
  // fresh-object semantics (the reference's objects are single-use).
  bool Encode(const uint8_t *data,
              int width,
              int height,
              int pixel_stride,
              int num_channels,
              int quality,
              bool use_ycbcr);

  const uint8_t *packed_data() const { return m_packed_data.get(); }

  int packed_size() const { return static_cast<int>(m_packed_size); }

 private:
  himg_hip_ctx *m_ctx;
  himg_hip_multi *m_multi;                    // HIMG_DEVICES names several devices
  std::unique_ptr<uint8_t[]> m_packed_data;  // uninitialised storage, exactly the stream
  size_t m_packed_size;
};

}  // namespace himg

#endif  // ENCODER_H_
