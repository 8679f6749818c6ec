// dhimg -- decompress a .himg file on the MI355X engine.
//
// Command line, messages and exit codes follow the reference tool
// (src/dhimg.cpp:17-72): "dhimg image outfile"; exit 0 on bad arguments, -1 when
// the input cannot be read, decoded or the output cannot be written.  The picture
// is written as binary PGM / PPM / PAM by channel count (the reference writes PNG
// through FreeImage, which this image does not have).
#include <cstdio>
#include <vector>

#include "decoder.h"
#include "pnm_io.h"

namespace {

int fail(const char *what, const char *path) {
  if (path) printf("%s %s\n", what, path);
  else printf("%s\n", what);
  return -1;
}

}  // namespace

int main(int argc, const char **argv) {
  if (argc < 3) {
    printf("Usage: %s image outfile\n", argv[0]);
    return 0;
  }
  const char *in_path = argv[1], *out_path = argv[2];

  std::vector<uint8_t> stream;
  if (!pnm::slurp(in_path, &stream)) return fail("Unable to read file", in_path);
  printf("File size: %zu\n", stream.size());
  fflush(stdout);   // the library reports through std::cout

  himg::Decoder decoder;
  if (!decoder.Decode(stream.data(), static_cast<int>(stream.size()))) return fail("Unable to decode image.", nullptr);

  // The codec hands back FreeImage's memory convention (bottom-up, BGR(A)); files
  // are top-down RGB(A).
  pnm::Image picture;
  picture.width = decoder.width();
  picture.height = decoder.height();
  picture.channels = decoder.num_channels();
  picture.data.resize(static_cast<size_t>(decoder.unpacked_size()));
  pnm::flip_and_swap(decoder.unpacked_data(), picture.data.data(), picture.width, picture.height, picture.channels);
  const bool writable = picture.channels == 1 || picture.channels == 3 || picture.channels == 4;
  if (!writable || !pnm::write(out_path, picture)) return fail("Unable to write file", out_path);
  return 0;
}
