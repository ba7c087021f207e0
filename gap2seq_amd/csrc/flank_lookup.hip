// gap2seq_amd/csrc/flank_lookup.hip — flank k-mers -> oriented node ids on the device.
//
// Every gap needs graph.buildNode + graph.contains for the k-mers of its flanks: the lmf+1 left
// seeds kmer_left.substr(d, k), the rmf+1 right-search seeds kmer_right.substr(len-k-j, k) and the
// rmf+1 targets kmer_right.substr(j, k) (/root/reference/src/Gap2Seq.cpp:878-884, 953-957,
// 995-1000, 1083-1086, 1113-1114): 33 look-ups per gap at -fuz 10.  On the host they were a third
// of a 500-gap step (binary searches over the sorted k-mer set, cache-missing); here one wave per
// gap does them in parallel in front of the fill kernel, on the same stream: lane i encodes its
// k-mer from the flank text (GATB codec, kmer.hpp), takes the canonical form and finds it in the
// device copy of the sorted k-mer set through the 22-bit prefix index.  The flank text and the
// per-gap descriptors are read straight from pinned host memory; the node ids go to HBM (for the
// fill kernels) and to pinned host memory (for the host half of phase D).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "fill_device.h"
#include "flank_device.h"
#include "flank_lookup.h"
#include "kmer.hpp"

namespace {

using g2s::u128;

// FLANK_WAVES waves a workgroup, a gap per wave at a time (a quarter of the dispatches of one-wave workgroups).  What
// rocprofv3 shows as this kernel's 75-80 us on a 10 000-gap list is mostly the 1.5 MB copy of descriptors and flank text
// in front of it (53 us with the look-ups switched off; the look-ups themselves ~22 us)
#define FLANK_WAVES 4
template <class KT>
__global__ __launch_bounds__(64 * FLANK_WAVES) void g2s_resolve_flanks(g2s::FlankLookup lk, const g2s::FlankDesc* __restrict__ desc,
                                                                       const char* __restrict__ text, uint32_t* __restrict__ nodes_dev,
                                                                       uint32_t* __restrict__ nodes_host, uint32_t ngaps) {
  // the gap's flank text: [left: first k+lmf chars][right: first k+rmf chars][right: last k+rmf chars] — the last part
  // left out when it is the second (a flank of exactly k+rmf characters: what GapCutter writes)
  __shared__ __attribute__((aligned(16))) uint32_t tws[FLANK_WAVES][G2S_FLANK_TEXT_MAX / 4 + 20];  // (+20: flank_encode reads sizeof(KT) + 1 words from an item's first)
  const int lane = (int)(threadIdx.x & 63u);
  const uint32_t wave = threadIdx.x >> 6;
  uint32_t* tw = tws[wave];
  const int k = lk.k;
  for (uint32_t gap = blockIdx.x * FLANK_WAVES + wave; gap < ngaps; gap += gridDim.x * FLANK_WAVES) {
    const g2s::FlankDesc d = desc[gap];
    const bool once = (d.rmf & G2S_FLANK_RIGHT_ONCE) != 0u;
    const int rmf = (int)(d.rmf & 0x7FFFu);
    const int nl = (int)d.lmf + 1, nr = rmf + 1;
    const int llen = k + (int)d.lmf, rlen = k + rmf;
    (void)nl; (void)nr;
    const int tail = once ? llen : llen + rlen;  // where the right flank's last k+rmf characters begin
    const uint32_t words = (uint32_t)(tail + rlen + 3) / 4u;
    for (uint32_t w = (uint32_t)lane; w < words; w += 64u) tw[w] = ((const uint32_t*)(text + d.text_off))[w];  // (4-byte aligned, padded)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (the wave's own words: no other wave reads them)
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < nl + 2 * nr; i += 64) {
      const uint32_t node = g2s::flank_node_of<KT>(lk, tw, g2s::flank_item_offset(i, k, (int)d.lmf, rmf, tail));
      nodes_dev[d.flank_off + (uint32_t)i] = node;
      nodes_host[d.flank_off + (uint32_t)i] = node;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the text has been read before the next gap's overwrites it)
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

namespace g2s {

hipError_t launch_resolve_flanks(hipStream_t st, const FlankLookup& lk, uint32_t ngaps, const FlankDesc* desc, const char* text,
                                 uint32_t* nodes_dev, uint32_t* nodes_host) {
  if (ngaps == 0) return hipSuccess;
  const uint32_t wgs = std::min<uint32_t>((ngaps + FLANK_WAVES - 1u) / FLANK_WAVES, 4096u);
  if (lk.wide)
    hipLaunchKernelGGL(g2s_resolve_flanks<u128>, dim3(wgs), dim3(64 * FLANK_WAVES), 0, st, lk, desc, text, nodes_dev, nodes_host, ngaps);
  else
    hipLaunchKernelGGL(g2s_resolve_flanks<uint64_t>, dim3(wgs), dim3(64 * FLANK_WAVES), 0, st, lk, desc, text, nodes_dev, nodes_host, ngaps);
  return hipGetLastError();
}

}  // namespace g2s
