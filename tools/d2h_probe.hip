// tools/d2h_probe.hip — device-to-host copy bandwidth into pinned memory (sizing the D2H leg).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
  for (size_t mb : {16, 64, 256}) {
    size_t bytes = mb << 20;
    void *d, *h;
    hipMalloc(&d, bytes);
    hipMemset(d, 1, bytes);
    for (unsigned flags : {0u, (unsigned)hipHostMallocNonCoherent, (unsigned)hipHostMallocNumaUser}) {
      if (hipHostMalloc(&h, bytes, flags) != hipSuccess) { printf("alloc flags %u failed\n", flags); continue; }
      hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
      for (int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rep == 2) printf("%4zu MB flags %u: %.3f ms = %.1f GB/s\n", mb, flags, ms, bytes / ms / 1e6);
      }
      hipHostFree(h);
      hipStreamDestroy(st);
    }
    hipFree(d);
  }
  return 0;
}
