// gap2seq_amd/csrc/flank_device.h — one flank k-mer -> oriented node id, on the device: graph.buildNode +
// graph.contains of /root/reference/src/Gap2Seq.cpp:878-884, 953-957, 995-1000, 1083-1086, 1113-1114 (GATB codec,
// kmer.hpp; the sorted canonical k-mer set behind a prefix index, dbg.cpp: rank_of).  Shared by the look-up kernel
// (flank_lookup.hip: one wave per gap in front of the fill kernels) and by the segment tier's kernels, whose waves
// resolve their own gap's flanks in their prologue when the list stays on the device (fill_seg.hip, round 6: the
// look-up launch and the dependency behind it were 10 us of a 500-gap step and 75 us of a 10 000-gap one).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fill_device.h"
#include "flank_lookup.h"
#include "kmer.hpp"

namespace g2s {

__device__ __forceinline__ uint64_t d_revcomp32(uint64_t x) {
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = __builtin_bswap64(x);
  return x ^ 0xAAAAAAAAAAAAAAAAULL;
}
__device__ __forceinline__ uint64_t d_revcomp(uint64_t x, int k) { return d_revcomp32(x) >> (64 - 2 * k); }
__device__ __forceinline__ u128 d_revcomp(u128 x, int k) {
  const u128 y = ((u128)d_revcomp32((uint64_t)x) << 64) | (u128)d_revcomp32((uint64_t)(x >> 64));
  return y >> (128 - 2 * k);
}

// the k characters at t (any memory the lane can read bytes of: the kernels stage the gap's flank text in LDS)
template <class KT>
__device__ __forceinline__ uint32_t flank_node_of(const FlankLookup& lk, const char* t) {
  const int k = lk.k;
  const KT* v = (const KT*)lk.kmers;
  const int shift = 2 * k - lk.bucket_bits;
  KT f = 0;
  for (int c = 0; c < k; c++) f = (f << 2) | (KT)((t[c] >> 1) & 3);  // GATB codec: A0 C1 T2 G3, any byte maps to a base
  const KT r = d_revcomp(f, k);
  const bool fwd = f < r;
  const KT canon = fwd ? f : r;
  // sorted rank through the prefix index (dbg.cpp: rank_of)
  const size_t b = (size_t)(canon >> shift);
  uint32_t lo = lk.bucket[b], hi = lk.bucket[b + 1];
  const uint32_t end = hi;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (v[mid] < canon) lo = mid + 1; else hi = mid;
  }
  // (the k-mer at the rank and the rank's node asked for together: one round trip instead of two)
  const uint32_t at = lo < end ? lo : (end ? end - 1u : 0u);
  const KT found = v[at];
  const uint32_t r2n = lk.rank2node[at];
  return (lo < end && found == canon) ? (r2n ^ (fwd ? 0u : 1u)) : G2S_DEV_INVALID;
}

// where item i of a gap's (lmf + 1) + 2 (rmf + 1) flank k-mers starts in the gap's flank text
// [left: first k+lmf chars][right: first k+rmf chars][right: last k+rmf chars] (tail: where the third part begins)
__device__ __forceinline__ int flank_item_offset(int i, int k, int lmf, int rmf, int tail) {
  const int nl = lmf + 1, nr = rmf + 1, llen = k + lmf, rlen = k + rmf;
  if (i < nl) return i;                                   // left.substr(d, k)          :995,1083
  if (i < nl + nr) return tail + (rlen - k - (i - nl));   // right.substr(len-k-j, k)    :878,954
  return llen + (i - nl - nr);                            // right.substr(j, k)         :1113
}

}  // namespace g2s
