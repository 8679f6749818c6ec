// freeimage_link_stub.cpp -- TEST-ONLY definitions of the FreeImage entry points the
// reference's benchmark main calls (src/benchmark.cpp:107,129-135,156), so that
// /root/reference/src/benchmark.cpp -- compiled where it lies, unchanged -- LINKS and RUNS
// against this engine's himg::Decoder (tests/test_reference_benchmark.py).  Every loader
// here fails: the only usable branch of that main is the one a .himg file takes
// (IsHimg -> himg::Decoder::Decode, src/benchmark.cpp:119-126), which is the drop-in
// claim under test.  Not part of the product; not an oracle build (nothing of the
// codec is stood in for -- FreeImage is file I/O of the three mains only).
#include "FreeImage.h"

extern "C" {
void FreeImage_Initialise(BOOL) {}
void FreeImage_DeInitialise(void) {}
FREE_IMAGE_FORMAT FreeImage_GetFIFFromFilename(const char *) { return FIF_UNKNOWN; }
FIMEMORY *FreeImage_OpenMemory(BYTE *, DWORD) { return 0; }
void FreeImage_CloseMemory(FIMEMORY *) {}
FIBITMAP *FreeImage_LoadFromMemory(FREE_IMAGE_FORMAT, FIMEMORY *, int) { return 0; }
void FreeImage_Unload(FIBITMAP *) {}
}
