// pnm_io.h -- minimal Netpbm reader/writer for the command line tools.
//
// The reference's chimg/dhimg use FreeImage for file I/O (chimg.cpp:101-137,
// dhimg.cpp:50-68), which is not part of this image; these tools read and write
// binary PGM (P5), PPM (P6) and PAM (P7: GRAYSCALE, RGB, RGB_ALPHA), 8 bits per
// sample.  To stay interchangeable with files made by the reference tools the
// pixels are handed to the codec in FreeImage's memory convention: bottom-up
// scanlines, BGR(A) channel order (SURVEY.md 8b "CLI contract").
#ifndef HIMG_CLI_PNM_IO_H_
#define HIMG_CLI_PNM_IO_H_

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace pnm {

struct Image {
  int width = 0, height = 0, channels = 0;
  std::vector<uint8_t> data;  // top-down, RGB(A) / grey, tightly packed
};

inline bool next_token(FILE *f, std::string *tok) {
  tok->clear();
  int c;
  for (;;) {
    c = fgetc(f);
    if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
    if (c == EOF) return false;
    if (c != ' ' && c != '\t' && c != '\n' && c != '\r') break;
  }
  while (c != EOF && c != ' ' && c != '\t' && c != '\n' && c != '\r') { tok->push_back((char)c); c = fgetc(f); }
  return true;
}

// Returns 0 on success, 1 if the file cannot be opened / read, 2 if the format is not supported.
inline int read(const char *path, Image *img) {
  FILE *f = fopen(path, "rb");
  if (!f) return 1;
  std::string t;
  int rc = 2;
  do {
    if (!next_token(f, &t)) break;
    int maxval = 0;
    if (t == "P5" || t == "P6") {
      img->channels = t == "P5" ? 1 : 3;
      std::string a, b, c;
      if (!next_token(f, &a) || !next_token(f, &b) || !next_token(f, &c)) break;
      img->width = atoi(a.c_str()); img->height = atoi(b.c_str()); maxval = atoi(c.c_str());
    } else if (t == "P7") {
      std::string key, val;
      int depth = 0;
      bool ok = false;
      while (next_token(f, &key)) {
        if (key == "ENDHDR") { ok = true; break; }
        if (!next_token(f, &val)) break;
        if (key == "WIDTH") img->width = atoi(val.c_str());
        else if (key == "HEIGHT") img->height = atoi(val.c_str());
        else if (key == "DEPTH") depth = atoi(val.c_str());
        else if (key == "MAXVAL") maxval = atoi(val.c_str());
      }
      if (!ok) break;
      // ENDHDR is followed by exactly one newline, which next_token consumed.
      img->channels = depth;
    } else {
      break;
    }
    if (img->width <= 0 || img->height <= 0 || maxval != 255 ||
        (img->channels != 1 && img->channels != 3 && img->channels != 4)) break;
    const size_t n = (size_t)img->width * img->height * img->channels;
    img->data.resize(n);
    rc = fread(img->data.data(), 1, n, f) == n ? 0 : 1;
  } while (0);
  fclose(f);
  return rc;
}

inline bool write(const char *path, const Image &img) {
  FILE *f = fopen(path, "wb");
  if (!f) return false;
  if (img.channels == 1) fprintf(f, "P5\n%d %d\n255\n", img.width, img.height);
  else if (img.channels == 3) fprintf(f, "P6\n%d %d\n255\n", img.width, img.height);
  else fprintf(f, "P7\nWIDTH %d\nHEIGHT %d\nDEPTH %d\nMAXVAL 255\nTUPLTYPE %s\nENDHDR\n", img.width, img.height,
               img.channels, img.channels == 4 ? "RGB_ALPHA" : "GRAYSCALE");
  const bool ok = fwrite(img.data.data(), 1, img.data.size(), f) == img.data.size();
  fclose(f);
  return ok;
}

// Whole file into memory.  Returns false if it cannot be opened.
inline bool slurp(const char *path, std::vector<uint8_t> *bytes) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  bytes->clear();
  uint8_t chunk[1 << 16];
  size_t n;
  while ((n = fread(chunk, 1, sizeof(chunk), f)) > 0) bytes->insert(bytes->end(), chunk, chunk + n);
  fclose(f);
  return true;
}

// Top-down RGB(A) <-> FreeImage's bottom-up BGR(A) (the conversion is its own inverse).
inline void flip_and_swap(const uint8_t *src, uint8_t *dst, int width, int height, int channels) {
  const size_t pitch = (size_t)width * channels;
  for (int y = 0; y < height; ++y) {
    const uint8_t *s = src + (size_t)y * pitch;
    uint8_t *d = dst + (size_t)(height - 1 - y) * pitch;
    if (channels >= 3) {
      for (int x = 0; x < width; ++x) {
        d[x * channels + 0] = s[x * channels + 2];
        d[x * channels + 1] = s[x * channels + 1];
        d[x * channels + 2] = s[x * channels + 0];
        if (channels == 4) d[x * channels + 3] = s[x * channels + 3];
      }
    } else {
      memcpy(d, s, pitch);
    }
  }
}

}  // namespace pnm
#endif  // HIMG_CLI_PNM_IO_H_
