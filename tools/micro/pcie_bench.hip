// Micro-benchmark: host<->device copy strategies for one 64 MiB frame.
// hipcc --offload-arch=gfx950 -O2 -o pcie_bench pcie_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t n = 64u << 20;
  void *d; hipMalloc(&d, n);
  void *pin; hipHostMalloc(&pin, n, hipHostMallocDefault);
  char *pg = (char *)malloc(n); memset(pg, 1, n);
  hipMemcpy(d, pin, n, hipMemcpyHostToDevice);
  auto rep = [&](const char *name, auto fn) {
    fn();
    double t = now();
    for (int i = 0; i < 5; ++i) fn();
    double ms = (now() - t) / 5 * 1e3;
    printf("%-46s %7.2f ms  %6.1f GB/s\n", name, ms, n / ms / 1e6);
  };
  rep("H2D pageable hipMemcpy", [&] { hipMemcpy(d, pg, n, hipMemcpyHostToDevice); });
  rep("H2D pinned hipMemcpy", [&] { hipMemcpy(d, pin, n, hipMemcpyHostToDevice); });
  rep("H2D register + copy + unregister", [&] {
    hipHostRegister(pg, n, hipHostRegisterDefault); hipMemcpy(d, pg, n, hipMemcpyHostToDevice); hipHostUnregister(pg); });
  rep("H2D memcpy to pinned + copy", [&] { memcpy(pin, pg, n); hipMemcpy(d, pin, n, hipMemcpyHostToDevice); });
  rep("D2H pageable (reused buffer)", [&] { hipMemcpy(pg, d, n, hipMemcpyDeviceToHost); });
  rep("D2H pageable (fresh malloc each time)", [&] { char *q = (char *)malloc(n); hipMemcpy(q, d, n, hipMemcpyDeviceToHost); free(q); });
  rep("D2H pinned", [&] { hipMemcpy(pin, d, n, hipMemcpyDeviceToHost); });
  rep("D2H pinned + memcpy to fresh malloc", [&] { char *q = (char *)malloc(n); hipMemcpy(pin, d, n, hipMemcpyDeviceToHost); memcpy(q, pin, n); free(q); });
  rep("hipHostMalloc + hipHostFree of 64 MiB", [&] { void *q; hipHostMalloc(&q, n, hipHostMallocDefault); hipHostFree(q); });
  return 0;
}
