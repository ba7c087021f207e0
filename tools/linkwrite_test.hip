// GPU box (hipcc -O2 --offload-arch=gfx950 tools/linkwrite_test.hip -o /tmp/lw && /tmp/lw): how fast do 10 000 waves write
// 640 bytes each into page-locked host memory with 1-, 4- and 16-byte stores per lane?  (What bounds g2s_d3_trace on
// a long list is this link: DESIGN.md 3.4.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int W>
__global__ __launch_bounds__(64) void k(char* out, int bytes_per_wave) {
  char* p = out + (size_t)blockIdx.x * bytes_per_wave;
  const int lane = threadIdx.x;
  if (W == 1) for (int i = lane; i < bytes_per_wave; i += 64) p[i] = (char)(i + lane);
  if (W == 4) for (int i = lane * 4; i < bytes_per_wave; i += 256) *(uint32_t*)(p + i) = 0x01020304u + i;
  if (W == 16) for (int i = lane * 16; i < bytes_per_wave; i += 1024) *(uint4*)(p + i) = make_uint4(i, i + 1, i + 2, i + 3);
}
int main() {
  const int waves = 10000, bpw = 768;
  char* h;
  hipHostMalloc(&h, (size_t)waves * bpw, hipHostMallocMapped | hipHostMallocCoherent);
  char* d;
  hipHostGetDevicePointer((void**)&d, h, 0);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; rep++) {
    float ms[3];
    for (int w = 0; w < 3; w++) {
      hipEventRecord(a, 0);
      for (int it = 0; it < 10; it++) {
        if (w == 0) hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, d, bpw);
        if (w == 1) hipLaunchKernelGGL(k<4>, dim3(waves), dim3(64), 0, 0, d, bpw);
        if (w == 2) hipLaunchKernelGGL(k<16>, dim3(waves), dim3(64), 0, 0, d, bpw);
      }
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
      hipEventElapsedTime(&ms[w], a, b);
    }
    printf("7.68 MB per launch: byte stores %.1f us (%.1f GB/s), dword stores %.1f us (%.1f GB/s), 16-byte stores %.1f us (%.1f GB/s)\n",
           ms[0] * 100, 7.68e-3 / (ms[0] / 10 * 1e-3), ms[1] * 100, 7.68e-3 / (ms[1] / 10 * 1e-3), ms[2] * 100, 7.68e-3 / (ms[2] / 10 * 1e-3));
  }
  return 0;
}
