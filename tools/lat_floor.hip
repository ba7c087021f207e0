// tools/lat_floor.hip — the latency floor of the segment search (profiles/r03_latency_floor.txt).
// A round of phase A or B of g2s_fill_seg is one DEPENDENT load of a 32-byte record (urec[v]: where the unitig
// that begins at v ends and what follows it) from tables of ~264 MB, plus the work on what came back: the next
// round's addresses are in this round's records.  This probe measures that dependent load by itself: a few
// lanes of one wave per workgroup each chase their own chain through a 268 MB table of 32-byte records, with
// 1 .. 10 000 such waves on the chip (the launch sizes of BASELINE configs 2 and 3).
// Build: hipcc --offload-arch=gfx950 -O3 tools/lat_floor.hip -o tools/lat_floor.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void chase(const uint4* __restrict__ tab, unsigned mask, int iters, int lanes, unsigned long long* out) {
  unsigned x = (threadIdx.x * 2654435761u + blockIdx.x * 40503u) & mask;
  const bool act = (int)threadIdx.x < lanes;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (act)
    for (int i = 0; i < iters; i++) {
      const uint4 a = tab[2 * (size_t)x], b = tab[2 * (size_t)x + 1];  // the record: two 16-byte halves, as the kernel reads it
      x = (a.x ^ b.x) & mask;
    }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 3] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; }
  if (act && x == 0xFFFFFFFFu) out[blockIdx.x * 3 + 2] = x;
}

int main() {
  const unsigned recs = 1u << 23;  // 8 M records x 32 B = 268 MB
  const int iters = 4000;
  uint4* d_tab; unsigned long long* d_out;
  hipMalloc(&d_tab, (size_t)recs * 32);
  hipMalloc(&d_out, 16384 * 3 * 8);
  {
    std::vector<uint4> h((size_t)recs * 2);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); i++) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      h[i] = make_uint4((unsigned)s, (unsigned)(s >> 32), (unsigned)(s >> 16), (unsigned)(s >> 8));
    }
    hipMemcpy(d_tab, h.data(), h.size() * 16, hipMemcpyHostToDevice);
  }
  std::vector<unsigned long long> o(16384 * 3);
  printf("# dependent 32-byte record loads from a 268 MB table, one wave per workgroup, `lanes` chains per wave\n");
  printf("# waves lanes cycles_per_step ns_per_step clock_MHz\n");
  for (int rep = 0; rep < 2; rep++)
    for (int blocks : {1, 256, 500, 1250, 2500, 10000})
      for (int lanes : {1, 8}) {
        hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, d_tab, recs - 1, iters, lanes, d_out);
        hipDeviceSynchronize();
        hipMemcpy(o.data(), d_out, (size_t)blocks * 3 * 8, hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0;
        for (int b = 0; b < blocks; b++) { cyc += o[b * 3]; rt += o[b * 3 + 1]; }
        cyc /= blocks; rt /= blocks;
        if (rep) printf("%6d %2d %10.1f %9.1f %7.0f\n", blocks, lanes, cyc / iters, rt * 10.0 / iters, cyc / (rt * 10.0) * 1000.0);
      }
  return 0;
}
