// GPU box (hipcc -O2 --offload-arch=gfx950 tools/ldsdma_test.hip -o /tmp/t && /tmp/t): where global_load_lds_dwordx4 /
// _dword put a lane's data (LDS base + 16 or 4 bytes x lane id) and that a single active lane works.  Behind the round-3
// experiment that asked for the exit records of new events with LDS-direct loads when the events are created: no gain —
// a round waits for the record of its LAST child, which is asked for at the end of the round before.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* __restrict__ g, unsigned* out, int only) {
  __shared__ unsigned stage[64 * 4 + 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 4 + 64; i += 64) stage[i] = 0xDEADu;
  __syncthreads();
  if (only < 0 || lane == only) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 8 * lane), (__attribute__((address_space(3))) void*)stage, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 8 * lane + 4), (__attribute__((address_space(3))) void*)(stage + 256), 4, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 64 * 4 + 64; i += 64) out[i] = stage[i];
}
int main() {
  std::vector<unsigned> h(64 * 8);
  for (int i = 0; i < 64 * 8; i++) h[i] = 1000u * (i / 8) + (i % 8);
  unsigned *d, *o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 320 * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int only : {-1, 5}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, only);
    std::vector<unsigned> r(320);
    hipMemcpy(r.data(), o, 320 * 4, hipMemcpyDeviceToHost);
    printf("only=%d: x4 slots:", only);
    for (int l : {0, 1, 5, 63}) printf(" lane%d=[%u %u %u %u]", l, r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
    printf(" | dword slots: l0=%u l1=%u l5=%u l63=%u\n", r[256], r[257], r[261], r[319]);
  }
  return 0;
}
