// ctx_pool.h -- process-wide pool of engine contexts for the C++ wrapper classes.
//
// The reference's Encoder objects are single-use (SURVEY.md trap T4), so callers
// construct one per picture (src/chimg.cpp:140).  A fresh himg_hip_ctx per object
// would mean a fresh device workspace (hipMalloc of ~200 MB, several ms) per
// picture; instead objects borrow a context and hand it back on destruction.
// Pooled contexts are deliberately not destroyed at exit (the HIP runtime may
// already be gone when static destructors run).
#ifndef HIMG_CTX_POOL_H_
#define HIMG_CTX_POOL_H_

#include <mutex>
#include <vector>

#include "himg_hip.h"

namespace himg {
namespace detail {

inline std::mutex &pool_mutex() { static std::mutex m; return m; }
inline std::vector<himg_hip_ctx *> &pool() { static std::vector<himg_hip_ctx *> *p = new std::vector<himg_hip_ctx *>(); return *p; }

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
  return himg_hip_create(0, &c) == HIMG_OK ? c : nullptr;
}

inline void release_ctx(himg_hip_ctx *c) {
  if (!c) return;
  std::lock_guard<std::mutex> g(pool_mutex());
  pool().push_back(c);
}

}  // namespace detail
}  // namespace himg
#endif  // HIMG_CTX_POOL_H_
