// tools/lds_load_probe.hip — where __builtin_amdgcn_global_load_lds puts its data on gfx950: lane L of the wave
// writes at the LDS base + L * size (16-byte and 4-byte forms), whatever lanes are active.  Measured on an MI355X:
//   hipcc --offload-arch=gfx950 tools/lds_load_probe.hip -o /tmp/probe && /tmp/probe
// (the record prefetch DESIGN.md §8 lists as a next step would use it)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint32_t* __restrict__ g, uint32_t* out, int which) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[64 * 4 + 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 4 + 64; i += 64) buf[i] = 0xdeadbeefu;
  __syncthreads();
  if (which == 0) {
    if (lane == 5 || lane == 9) __builtin_amdgcn_global_load_lds(g + 8 * lane, (__attribute__((address_space(3))) void*)buf, 16, 0, 0);
  } else {
    if (lane == 5 || lane == 9) __builtin_amdgcn_global_load_lds(g + 8 * lane + 4, (__attribute__((address_space(3))) void*)(buf + 256), 4, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 64 * 4 + 64; i += 64) out[i] = buf[i];
}
int main() {
  std::vector<uint32_t> h(64 * 8);
  for (int i = 0; i < 64 * 8; i++) h[i] = i;
  uint32_t *g, *o;
  hipMalloc(&g, h.size() * 4); hipMalloc(&o, 320 * 4);
  hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int which = 0; which < 2; which++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, which);
    std::vector<uint32_t> r(320);
    hipMemcpy(r.data(), o, 320 * 4, hipMemcpyDeviceToHost);
    printf("which %d:", which);
    for (int i = 0; i < 320; i++) if (r[i] != 0xdeadbeefu) printf(" [%d]=%u", i, r[i]);
    printf("\n");
  }
  return 0;
}
