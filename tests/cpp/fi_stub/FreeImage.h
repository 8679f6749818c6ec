// FreeImage.h -- TEST-ONLY declarations of the few FreeImage entry points the
// reference's three mains touch (src/chimg.cpp:101-166, src/dhimg.cpp:53-68,
// src/benchmark.cpp:107-156).  Declarations only, no definitions: it exists so
// that tests/test_reference_callers.py can compile those files, where they lie
// under /root/reference, against include/encoder.h + include/decoder.h and prove
// the drop-in claim of INTEGRATION.md.  Not part of the product, never linked.
#ifndef HIMG_TEST_FREEIMAGE_STUB_H_
#define HIMG_TEST_FREEIMAGE_STUB_H_

typedef unsigned char BYTE;
typedef int BOOL;
typedef unsigned int DWORD;
struct FIBITMAP;
struct FIMEMORY;

enum FREE_IMAGE_FORMAT { FIF_UNKNOWN = -1, FIF_PNG = 13 };
enum FREE_IMAGE_COLOR_TYPE { FIC_MINISWHITE = 0, FIC_MINISBLACK = 1, FIC_RGB = 2, FIC_PALETTE = 3, FIC_RGBALPHA = 4, FIC_CMYK = 5 };

extern "C" {
void FreeImage_Initialise(BOOL load_local_plugins_only = 0);
void FreeImage_DeInitialise(void);
FREE_IMAGE_FORMAT FreeImage_GetFileType(const char *filename, int size = 0);
FREE_IMAGE_FORMAT FreeImage_GetFIFFromFilename(const char *filename);
FIBITMAP *FreeImage_Load(FREE_IMAGE_FORMAT fif, const char *filename, int flags = 0);
BOOL FreeImage_Save(FREE_IMAGE_FORMAT fif, FIBITMAP *dib, const char *filename, int flags = 0);
void FreeImage_Unload(FIBITMAP *dib);
FREE_IMAGE_COLOR_TYPE FreeImage_GetColorType(FIBITMAP *dib);
FIBITMAP *FreeImage_ConvertToGreyscale(FIBITMAP *dib);
FIBITMAP *FreeImage_ConvertTo24Bits(FIBITMAP *dib);
FIBITMAP *FreeImage_ConvertTo32Bits(FIBITMAP *dib);
unsigned FreeImage_GetWidth(FIBITMAP *dib);
unsigned FreeImage_GetHeight(FIBITMAP *dib);
BYTE *FreeImage_GetBits(FIBITMAP *dib);
FIBITMAP *FreeImage_ConvertFromRawBits(BYTE *bits, int width, int height, int pitch, unsigned bpp,
                                       unsigned red_mask, unsigned green_mask, unsigned blue_mask,
                                       BOOL topdown = 0);
FIMEMORY *FreeImage_OpenMemory(BYTE *data = 0, DWORD size_in_bytes = 0);
void FreeImage_CloseMemory(FIMEMORY *stream);
FIBITMAP *FreeImage_LoadFromMemory(FREE_IMAGE_FORMAT fif, FIMEMORY *stream, int flags = 0);
}

#endif  // HIMG_TEST_FREEIMAGE_STUB_H_
