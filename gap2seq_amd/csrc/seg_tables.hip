// gap2seq_amd/csrc/seg_tables.hip — the table the segment tier (fill_seg.hip) walks unitigs
// with: rem[v] = how many further unitig-internal steps the oriented node v can take in its
// walking direction (even orientation: ids grow by 2 per step, odd: shrink; dbg.hpp).  One
// 4-byte load then tells a wave how long a whole run of DP states is, where the LDS tier
// reads words of the unitig-start bitmap level by level.
//
// Built on the device from the unitig-start bitmap that is already there (bit i set = the
// edge 2(i-1) -> 2i is not unitig-internal), one wave per 64-bit word: lane b answers for
// k-mer index 64 w + b; the nearest set bits outside the word are found by the wave reading
// 64 neighbouring words per step.  Values are capped at G2S_REM_CAP: a longer unitig is
// simply walked in several segments (the cap is far above any DP depth).
//
// Replaces nothing of the reference by itself: it is the unitig-compacted view of
// Graph::successors (/root/reference/src/Gap2Seq.cpp:1043) on nodes with one way on.
#include <hip/hip_runtime.h>

#include "dbg.hpp"
#include "seg_tables.h"

namespace {

// words[-kUstartPad .. nwords + kUstartPad) are readable; the pads are all ones
__global__ __launch_bounds__(256) void k_rem(const uint64_t* __restrict__ words, uint64_t nwords, uint64_t n,
                                             uint32_t* __restrict__ rem) {
  const uint64_t w = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
  if (w >= nwords) return;
  const int lane = threadIdx.x & 63;
  const uint64_t word = words[w];
  const int64_t max_words = (int64_t)(G2S_REM_CAP / 64u) + 2;
  // nearest unitig start below this word: highest set bit of the words before it
  int64_t prev_start = (int64_t)w * 64 - (int64_t)G2S_REM_CAP - 64;  // "further away than the cap"
  for (int64_t o = 1; o <= max_words; o += 64) {
    const int64_t ww = (int64_t)w - o - lane;
    const bool in = o + lane <= max_words && ww >= -(int64_t)g2s::kUstartPad;
    const uint64_t x = in ? words[ww] : 0ull;
    const uint64_t m = __ballot(x != 0ull);
    if (m) {
      const int l = __builtin_ctzll(m);  // the nearest non-empty word
      const uint64_t xv = __shfl(x, l);
      prev_start = ((int64_t)w - o - l) * 64 + (63 - __builtin_clzll(xv));
      break;
    }
    if (__ballot(!in) != 0ull) break;
  }
  // nearest unitig start above this word
  int64_t next_start = (int64_t)w * 64 + 64 + (int64_t)G2S_REM_CAP + 64;
  for (int64_t o = 1; o <= max_words; o += 64) {
    const int64_t ww = (int64_t)w + o + lane;
    const bool in = o + lane <= max_words && ww < (int64_t)(nwords + g2s::kUstartPad);
    const uint64_t x = in ? words[ww] : 0ull;
    const uint64_t m = __ballot(x != 0ull);
    if (m) {
      const int l = __builtin_ctzll(m);
      const uint64_t xv = __shfl(x, l);
      next_start = ((int64_t)w + o + l) * 64 + __builtin_ctzll(xv);
      break;
    }
    if (__ballot(!in) != 0ull) break;
  }
  const uint64_t idx = w * 64 + (uint64_t)lane;
  if (idx >= n) return;
  const uint64_t le = word & (~0ull >> (63 - lane));                       // bits <= lane
  const uint64_t gt = lane == 63 ? 0ull : word & (~0ull << (lane + 1));    // bits > lane
  const int64_t start = le ? (int64_t)w * 64 + (63 - __builtin_clzll(le)) : prev_start;
  const int64_t nxt = gt ? (int64_t)w * 64 + __builtin_ctzll(gt) : next_start;
  const int64_t up = nxt - 1 - (int64_t)idx, dn = (int64_t)idx - start;
  uint2 out;
  out.x = (uint32_t)(up > (int64_t)G2S_REM_CAP ? (int64_t)G2S_REM_CAP : up);  // even orientation walks up
  out.y = (uint32_t)(dn > (int64_t)G2S_REM_CAP ? (int64_t)G2S_REM_CAP : dn);  // odd orientation walks down
  ((uint2*)rem)[idx] = out;
}

// urec[v] = {successor record of the LAST node of v's unitig walk (4 words), rem[v], 0, 0, 0}: what
// a segment that enters at v needs to know — how far it can go and where it leaves — in one
// 32-byte record, one memory round trip instead of two dependent ones (rem[v], then succ[end]).
__global__ __launch_bounds__(256) void k_urec(const uint32_t* __restrict__ succ, const uint32_t* __restrict__ rem,
                                              uint64_t n2, uint4* __restrict__ urec) {
  const uint64_t v = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (v >= n2) return;
  const uint32_t r = rem[v];
  const uint32_t e = (v & 1u) ? (uint32_t)v - 2u * r : (uint32_t)v + 2u * r;
  urec[2 * v] = *(const uint4*)(succ + (size_t)e * 4);
  urec[2 * v + 1] = make_uint4(r, 0u, 0u, 0u);
}

}  // namespace

namespace g2s {

hipError_t build_urec_table(const uint32_t* succ_dev, const uint32_t* rem_dev, uint64_t n, uint32_t** urec_out) {
  *urec_out = nullptr;
  if (n == 0) return hipSuccess;
  uint32_t* urec = nullptr;
  hipError_t e = hipMalloc((void**)&urec, (size_t)n * 64 + 64);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_urec, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, 0, succ_dev, rem_dev, 2 * n, (uint4*)urec);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(urec); return e; }
  *urec_out = urec;
  return hipSuccess;
}

hipError_t build_rem_table(const uint64_t* ustart_dev, uint64_t n, uint32_t** rem_out) {
  *rem_out = nullptr;
  if (n == 0) return hipSuccess;
  uint32_t* rem = nullptr;
  hipError_t e = hipMalloc((void**)&rem, (size_t)n * 8 + 64);
  if (e != hipSuccess) return e;
  const uint64_t nwords = (n + 63) / 64;
  hipLaunchKernelGGL(k_rem, dim3((unsigned)((nwords + 3) / 4)), dim3(256), 0, 0, ustart_dev, nwords, n, rem);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(rem); return e; }
  *rem_out = rem;
  return hipSuccess;
}

}  // namespace g2s
