// Dynamic trip counts of the hottest loops (tools/dynamic_mix.py): which share of a kernel's
// instructions its loops execute, so that the class-weighted issue cost of a kernel is
// taken over what it RUNS, not over what it contains.
//   HIMG_REGION_BEGIN / _END("name")  always compiled in: assembly comments by which
//                           tools/isa_mix.py finds the loop body in the listing.
//   -DHIMG_LOOP_COUNTS      (HIMG_EXTRA_HIPCC_FLAGS, never in the product build) adds the
//                           counters: per loop the iterations the WAVEFRONT executed (its
//                           busiest lane's) and the sum over its lanes; read and reset
//                           through himg_hip_debug_read(HIMG_DBG_LOOP_COUNTS).
#pragma once
#include <hip/hip_runtime.h>

#include "himg_dev.h"

// Around the BODY of a counted loop, or around straight-line code every wavefront runs once
// per unit of work: what lies between the two comments in the listing is the hot path
// (the compiler moves unlikely blocks out of line).
#define HIMG_REGION_BEGIN(name) asm volatile("; HIMG_REGION_BEGIN " name)
#define HIMG_REGION_END(name) asm volatile("; HIMG_REGION_END " name)
// A straight-line span that may sit INSIDE a loop (the persistent row kernel's transform: once per
// pass through the loop body); tools/isa_mix.py lists it under "regions" like a span outside loops.
#define HIMG_SPAN_BEGIN(name) asm volatile("; HIMG_SPAN_BEGIN " name)
#define HIMG_SPAN_END(name) asm volatile("; HIMG_SPAN_END " name)

namespace himg_dev {
#ifdef HIMG_LOOP_COUNTS
static __device__ unsigned long long g_loop_counts[2 * kLoopCounters];
struct LoopCount {
  uint32_t n = 0;
  __device__ __forceinline__ void step() { ++n; }
  // Behind the loop (whatever lanes are active here: the others did not run it).
  __device__ __forceinline__ void done(int id) {
    unsigned long long m = __ballot(1);
    const int first = __ffsll((long long)m) - 1;
    uint32_t mx = 0;
    while (m) {
      const int l = __ffsll((long long)m) - 1;
      m &= m - 1;
      const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)n, l);
      mx = v > mx ? v : mx;
    }
    if ((int)(threadIdx.x & 63) == first && mx) atomicAdd(&g_loop_counts[2 * id], (unsigned long long)mx);
    if (n) atomicAdd(&g_loop_counts[2 * id + 1], (unsigned long long)n);
    n = 0;
  }
};
inline int loop_counts_read(unsigned long long *out) {   // read and reset
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_loop_counts), sizeof(g_loop_counts)) != hipSuccess) return -1;
  unsigned long long z[2 * kLoopCounters] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_loop_counts), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#else
struct LoopCount {
  __device__ __forceinline__ void step() {}
  __device__ __forceinline__ void done(int) {}
};
inline int loop_counts_read(unsigned long long *) { return -1; }   // not compiled in
#endif
}  // namespace himg_dev
