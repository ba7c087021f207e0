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

// The k characters from byte `off` of the text at `tw` (4-byte aligned words, in LDS; sizeof(KT) + 1 words from word
// off / 4 on are read, whatever they hold) as a k-mer, GATB codec: A0 C1 T2 G3, any byte maps to a base ((c >> 1) & 3),
// first base most significant.  Four characters a word: the four 2-bit codes of a word gathered into a byte by one
// multiplication (the codes sit at bits 0, 8, 16, 24; times 2^30 + 2^20 + 2^10 + 1 they meet, in text order, in the top
// byte — no two partial products overlap), the bytes strung together, the window of 2k bits cut out.  (A byte a step —
// one LDS read, a wait and three operations per character — was 4 000 cycles of a gap's wave at k = 31.)
template <class KT>
__device__ __forceinline__ KT flank_encode(const uint32_t* tw, int off, int k) {
  constexpr int NB = (int)sizeof(KT);
  const uint32_t* w = tw + (off >> 2);
  uint32_t code[NB + 1];
#pragma unroll
  for (int j = 0; j <= NB; j++) code[j] = ((((w[j] >> 1) & 0x03030303u) * 0x40100401u) >> 24) & 0xFFu;
  KT a = 0;
#pragma unroll
  for (int j = 0; j < NB; j++) a = (a << 8) | (KT)code[j];
  const int s0 = 2 * (off & 3);
  const KT win = s0 ? (KT)((a << s0) | (KT)(code[NB] >> (8 - s0))) : a;
  return win >> (8 * NB - 2 * k);
}

// the k characters at byte `off` of the text staged at tw (see flank_encode)
template <class KT>
__device__ __forceinline__ uint32_t flank_node_of(const FlankLookup& lk, const uint32_t* tw, int off) {
  const int k = lk.k;
  const KT* v = (const KT*)lk.kmers;
  const int shift = 2 * k - lk.bucket_bits;
  const KT f = flank_encode<KT>(tw, off, k);
  const KT r = d_revcomp(f, k);
  const bool fwd = f < r;
  const KT canon = fwd ? f : r;
  // sorted rank through the prefix index (dbg.cpp: rank_of).  A bucket holds less than one k-mer on average: its first
  // four and their nodes are asked for at once — two dependent round trips per look-up (the bounds, the candidates)
  // where a binary search and the final look took four; a fuller bucket is searched.
  const size_t b = (size_t)(canon >> shift);
  uint32_t lo = lk.bucket[b];
  const uint32_t end = lk.bucket[b + 1];
  if (end - lo > 4u) {
    uint32_t hi = end;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (v[mid] < canon) lo = mid + 1; else hi = mid;
    }
    if (lo >= end) return G2S_DEV_INVALID;
    const KT found = v[lo];
    const uint32_t r2n = lk.rank2node[lo];
    return found == canon ? (r2n ^ (fwd ? 0u : 1u)) : G2S_DEV_INVALID;
  }
  KT c4[4];
  uint32_t n4[4];
#pragma unroll
  for (uint32_t q = 0; q < 4u; q++) {
    const bool in = lo + q < end;
    c4[q] = in ? v[lo + q] : (KT)0;
    n4[q] = in ? lk.rank2node[lo + q] : 0u;
  }
  uint32_t node = G2S_DEV_INVALID;
#pragma unroll
  for (uint32_t q = 0; q < 4u; q++)
    if (lo + q < end && c4[q] == canon) node = n4[q] ^ (fwd ? 0u : 1u);
  return node;
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
