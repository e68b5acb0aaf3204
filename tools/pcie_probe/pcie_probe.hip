// Developer probe: how fast can a host batch (pageable caller memory) reach the GPU and come back?
//   hipcc -O2 --offload-arch=gfx950 tools/pcie_probe/pcie_probe.hip -o tools/pcie_probe/pcie_probe
// Prints GB/s for: pageable hipMemcpy, pinned hipMemcpyAsync, host memcpy into pinned staging (1 thread),
// hipHostRegister of the caller's buffer (+ the registered copy), both directions, and full duplex on two streams.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv)
{
  const size_t bytes = (argc > 1 ? atol(argv[1]) : 160) * 1000000ul;
  char *pg = (char *)malloc(bytes), *pg2 = (char *)malloc(bytes), *pin = NULL, *pin2 = NULL, *dev = NULL, *dev2 = NULL;
  memset(pg, 1, bytes); memset(pg2, 2, bytes);
  CK(hipHostMalloc((void **)&pin, bytes, hipHostMallocPortable)); CK(hipHostMalloc((void **)&pin2, bytes, hipHostMallocPortable));
  memset(pin, 3, bytes); memset(pin2, 4, bytes);
  CK(hipMalloc((void **)&dev, bytes)); CK(hipMalloc((void **)&dev2, bytes));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  for (int rep = 0; rep < 2; rep++) {
    double t = now(); CK(hipMemcpy(dev, pg, bytes, hipMemcpyHostToDevice)); double a = now() - t;
    t = now(); CK(hipMemcpy(pg2, dev, bytes, hipMemcpyDeviceToHost)); double b = now() - t;
    t = now(); CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); double c = now() - t;
    t = now(); CK(hipMemcpyAsync(pin2, dev, bytes, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); double d = now() - t;
    t = now(); CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(pin2, dev2, bytes, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); double e = now() - t;
    t = now(); memcpy(pin, pg, bytes); double f = now() - t;
    t = now(); memcpy(pg2, pin2, bytes); double g = now() - t;
    t = now(); CK(hipHostRegister(pg, bytes, hipHostRegisterDefault)); double h = now() - t;
    t = now(); CK(hipMemcpyAsync(dev, pg, bytes, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); double i = now() - t;
    t = now(); CK(hipHostUnregister(pg)); double j = now() - t;
    const double gb = bytes / 1e9;
    printf("rep %d, %.0f MB:  pageable H2D %.1f GB/s  D2H %.1f | pinned H2D %.1f  D2H %.1f  duplex %.1f+%.1f | memcpy->pinned %.1f  pinned->pageable %.1f |"
           " register %.2f ms (%.1f GB/s)  registered H2D %.1f  unregister %.2f ms\n", rep, bytes / 1e6, gb / a, gb / b, gb / c, gb / d, gb / e, gb / e,
           gb / f, gb / g, h * 1e3, gb / h, gb / i, j * 1e3);
  }
  return 0;
}
