// ctx_pool.h -- process-wide pool of engine contexts for the C++ wrapper classes.
//
// The reference's Encoder objects are single-use (SURVEY.md trap T4), so callers
// construct one per picture (src/chimg.cpp:140).  A fresh himg_hip_ctx per object
// would mean a fresh device workspace (hipMalloc of ~200 MB, several ms) per
// picture; instead objects borrow a context and hand it back on destruction.
// Pooled contexts are deliberately not destroyed at exit (the HIP runtime may
// already be gone when static destructors run).
//
// HIMG_DEVICES names the devices the classes use ("3", "0-7", "0,2,4", "0,0" = two
// slots on one GPU); with more than one, the objects borrow a multi-device handle
// (himg_hip_create_multi) and large frames are sharded by block rows over the devices
// -- the reference's callers (src/chimg.cpp, src/dhimg.cpp, src/benchmark.cpp) stay
// source-unchanged.
#ifndef HIMG_CTX_POOL_H_
#define HIMG_CTX_POOL_H_

#include <cstdlib>
#include <mutex>
#include <vector>

#include "himg_hip.h"

namespace himg {
namespace detail {

inline std::mutex &pool_mutex() { static std::mutex m; return m; }
inline std::vector<himg_hip_ctx *> &pool() { static std::vector<himg_hip_ctx *> *p = new std::vector<himg_hip_ctx *>(); return *p; }
inline std::vector<himg_hip_multi *> &multi_pool() { static std::vector<himg_hip_multi *> *p = new std::vector<himg_hip_multi *>(); return *p; }

// HIMG_DEVICES parsed once: comma-separated device numbers and ranges a-b.
inline const std::vector<int> &configured_devices() {
  static const std::vector<int> *devs = [] {
    std::vector<int> *v = new std::vector<int>();
    const char *e = std::getenv("HIMG_DEVICES");
    while (e && *e) {
      char *end = nullptr;
      const long a = std::strtol(e, &end, 10);
      if (end == e) break;
      long b = a;
      if (*end == '-') {
        const char *q = end + 1;
        b = std::strtol(q, &end, 10);
        if (end == q) break;
      }
      for (long d = a; d <= b && d - a < 64 && v->size() < 64; ++d) v->push_back((int)d);
      e = *end == ',' ? end + 1 : end;
      if (*end != ',' ) break;
    }
    if (v->empty()) v->push_back(0);
    return v;
  }();
  return *devs;
}

inline himg_hip_ctx *acquire_ctx() {
  {
    std::lock_guard<std::mutex> g(pool_mutex());
    if (!pool().empty()) {
      himg_hip_ctx *c = pool().back();
      pool().pop_back();
      return c;
    }
  }
  himg_hip_ctx *c = nullptr;
  return himg_hip_create(configured_devices()[0], &c) == HIMG_OK ? c : nullptr;
}

inline void release_ctx(himg_hip_ctx *c) {
  if (!c) return;
  std::lock_guard<std::mutex> g(pool_mutex());
  pool().push_back(c);
}

inline bool use_multi() { return configured_devices().size() > 1; }

inline himg_hip_multi *acquire_multi() {
  {
    std::lock_guard<std::mutex> g(pool_mutex());
    if (!multi_pool().empty()) {
      himg_hip_multi *m = multi_pool().back();
      multi_pool().pop_back();
      return m;
    }
  }
  himg_hip_multi *m = nullptr;
  const std::vector<int> &d = configured_devices();
  return himg_hip_create_multi(d.data(), (int)d.size(), &m) == HIMG_OK ? m : nullptr;
}

inline void release_multi(himg_hip_multi *m) {
  if (!m) return;
  std::lock_guard<std::mutex> g(pool_mutex());
  multi_pool().push_back(m);
}

}  // namespace detail
}  // namespace himg
#endif  // HIMG_CTX_POOL_H_
