// tools/lat_probe.hip — microbenchmark used while tuning the LDS tier: latency of a
// dependent chain of LDS reads / L2-hit global loads with ONE wave per workgroup at
// low occupancy (the regime a 500-gap batch runs in), and the shader clock it runs at.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lat_probe.hip -o /tmp/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void chain_lds(int iters, unsigned long long* out) {
  __shared__ unsigned int tab[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) tab[i] = (i * 37 + 11) & 1023;
  __syncthreads();
  unsigned int x = threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) x = tab[x];
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 3 + 0] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x * 3 + 2] = x; }
}

__global__ void chain_global(const unsigned int* tab, int iters, unsigned long long* out) {
  unsigned int x = (threadIdx.x + blockIdx.x * 977) & 0xFFFFF;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) x = tab[x];
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 3 + 0] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x * 3 + 2] = x; }
}

__global__ void chain_store_then_lds(unsigned long long* sink, int iters, unsigned long long* out) {
  __shared__ unsigned int tab[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) tab[i] = (i * 37 + 11) & 1023;
  __syncthreads();
  unsigned int x = threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
    x = tab[x];
    if (threadIdx.x < 2) sink[(size_t)blockIdx.x * 8192 + (i & 8191)] = x;  // fire-and-forget store per step
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 3 + 0] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x * 3 + 2] = x; }
}

int main() {
  const int iters = 20000;
  unsigned long long* d_out; unsigned int* d_tab; unsigned long long* d_sink;
  hipMalloc(&d_out, 4096 * 3 * 8);
  hipMalloc(&d_tab, (1 << 20) * 4);
  hipMalloc(&d_sink, (size_t)2048 * 8192 * 8);
  std::vector<unsigned int> h(1 << 20);
  for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned int)((i * 2654435761ull + 12345) & 0xFFFFF);
  hipMemcpy(d_tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<unsigned long long> o(4096 * 3);
  for (int rep = 0; rep < 2; rep++)
    for (int blocks : {256, 512, 2048}) {
      for (int which = 0; which < 3; which++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(chain_lds, dim3(blocks), dim3(64), 0, 0, iters, d_out);
        if (which == 1) hipLaunchKernelGGL(chain_global, dim3(blocks), dim3(64), 0, 0, d_tab, iters, d_out);
        if (which == 2) hipLaunchKernelGGL(chain_store_then_lds, dim3(blocks), dim3(64), 0, 0, d_sink, iters, d_out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(o.data(), d_out, blocks * 3 * 8, hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0;
        for (int b = 0; b < blocks; b++) { cyc += o[b * 3]; rt += o[b * 3 + 1]; }
        cyc /= blocks; rt /= blocks;
        printf("%-22s blocks %4d: kernel %.3f ms | per step: %.1f shader cycles, %.1f ns (realtime ctr) -> clock %.0f MHz, wall/iters %.1f ns\n",
               which == 0 ? "lds chain" : which == 1 ? "global chain (4MB tab)" : "lds chain + 2 stores", blocks, ms,
               cyc / iters, rt * 10.0 / iters, cyc / (rt * 10.0) * 1000.0, ms * 1e6 / iters);
      }
    }
  return 0;
}
