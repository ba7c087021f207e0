// gap2seq_amd/csrc/d3_device.hip — phase D3 of fill_gap on the device, for gfx950 (CDNA4).
//
// The reference draws rand() once per gap for the path length and once per traced base, in gap order
// (/root/reference/src/Gap2Seq.cpp:178,1440,1513): gap i's traceback reads the stream at the sum of the draws
// of all gaps before it.  Until round 2 the host did this part: an in-order pass over the gaps for the offsets
// (serial: ~4 % of the gaps draw a number of values that depends on the values drawn) and the tracebacks on a
// thread pool — 1.8 ms of host work beside a 0.48 ms kernel on a 10 000-gap list, and not divided by the
// number of GPUs.  Here the whole of it runs behind the fill kernel on its stream, with no host round trip:
//
//   g2s_d3_classify / g2s_d3_scan (one launch g2s_d3_front for lists of up to 3 072 gaps: the per-gap words stay in
//                  LDS between the two)  every gap's class (no phase D / fixed draw count / draw-dependent), its
//                  (fewest) draws and their spread, the skip rule of consecutive gaps of a record (:369,402), the
//                  prefix sums over the list and the layout of the tables below; one record per draw-dependent gap
//                  (D3Var) and the first half of one per gap (D3Trace) for the kernels behind.
//   g2s_rand_fill  the glibc TYPE_3 stream (x[n] = x[n-31] + x[n-3], the linear recurrence behind rand())
//                  materialised from the session's position on: a wave computes the state in front of its
//                  4096 values from the state the host hands over with three jump polynomials
//                  (x^(2^20 a) x^(4096 b) x^(64 l) modulo x^31 - x^28 - 1 over Z/2^32, seed independent
//                  tables), then every lane runs the recurrence for its 64 values.  On a stream of its own.
//   g2s_d3_tables  A draw-dependent gap v can only start at base_v + d, d in [0, R_v], R_v = the summed spreads of
//                  the draw-dependent gaps before it: its draw count for EVERY such start, one lane per (v, d), the
//                  closure's links and the tile's window of the stream in LDS — the serial chain "offset of v+1 =
//                  offset of v + draws of v" becomes table look-ups.
//   g2s_d3_blocks  the chain through 16 consecutive tables for every deviation a block can start with;
//   g2s_d3_chain   the chain over blocks (one lane), then back into the blocks: the deviation in front of
//                  every draw-dependent gap;
//   g2s_d3_handoff every gap's first draw and draw count (second half of D3Trace), a slot in pinned memory for the
//                  gaps whose closure the host analyses.  (Short lists: one launch g2s_d3_back for the three, the
//                  tables and the chain in LDS.)
//   g2s_d3_trace   one wave per gap: the traceback over the closure segments (post.cpp: seg_traceback is
//                  the host version and the reference for every branch here) as a per-segment choice of parent
//                  (all lanes: the draw made at a depth does not depend on the path), a chase through those, and a
//                  lane-parallel pass for base sources, safe bits and case; fill text and result record
//                  written where the caller wants them (pinned host memory, or a staging buffer); host-finished
//                  gaps handed over by their own waves; the last wave copies the summary to the host and, for a
//                  list of one session, zeroes what the next list expects to find zero.
//
// Anything out of the ordinary (a gap the segment tier could not finish or analyse, tables beyond the
// budget, a walk that does not end in a left-flank k-mer) is only COUNTED here: the host then discards the
// attempt and runs the list through the host path (g2s_api.hip), which remains the authority.
#include "sync_debug.h"
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <set>

#include "../../include/g2s.h"
#include "d3_device.h"
#include "envcache.hpp"

namespace {

using g2s::D3Gap;
using g2s::D3Params;
using g2s::D3Summary;
using g2s::D3Work;

__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// ginfo: class (2 bits) | filled << 2 | skipped << 3 | bad flank << 4 | mem verdict << 5 | phase D << 6 | host << 7 | right_fuz << 8
#define GI_CLASS(x) ((x) & 3u)
#define GI_FILLED 0x4u
#define GI_SKIPPED 0x8u
#define GI_BAD 0x10u
#define GI_MEM 0x20u
#define GI_PHASE_D 0x40u
#define D3_TILE 256u   /* deviations of one gap a workgroup of g2s_d3_tables takes */
#define GI_HOST 0x80u  /* the host finishes this gap: its closure was not analysed by the fill kernel */

// lap stamps (100 MHz clock) of the first workgroup of every kernel, in the second half of the summary's 1024 bytes:
// what G2S_DEBUG prints (g2s_api.hip) — how the time of the short kernels of a short list is spent
__device__ __forceinline__ void stamp(const D3Work& W, int slot) {
  ((unsigned long long*)((char*)W.sum + 512))[slot] = wall_clock64();
}

// the closure records of gap i's group (a list filled by several sessions arrives as one region per group)
__device__ __forceinline__ const SubRec* sub_of(const D3Params& P, const SubRec* sub, uint32_t i) {
  return sub + (uint64_t)(i / P.group_size) * P.sub_region;
}

// the value pp.count of the host analysis (g2s_api.hip: analyze_gap)
__device__ __forceinline__ int gap_count(const GapOut& go, const D3Params& P, bool phase_d) {
  if (phase_d && !P.skip_confident && P.all_paths) return go.count_s;
  return go.c_count;
}

// ---------------------------------------------------------------------------------------------------------
// classify: every gap by itself — class, (fewest) draws, spread; the list's counters into the summary (zeroed
// by the launcher).  One thread per gap.
// ---------------------------------------------------------------------------------------------------------
// sums and scans over the lanes as row shifts and row broadcasts inside the vector ALU (a shuffle through the LDS
// crossbar per step costs ten times as much: 96 of them were 2 us of a 500-gap list's classify)
__device__ __forceinline__ uint32_t dpp_scan_add(uint32_t x) {  // inclusive prefix sum
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);  // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);  // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);  // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);  // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
  return x;
}
__device__ __forceinline__ uint32_t sat_add(uint32_t a, uint32_t b) { const uint32_t t = a + b; return t < a ? 0xFFFFFFFFu : t; }
__device__ __forceinline__ uint32_t dpp_scan_sat(uint32_t x) {  // the same with saturating additions (associative too)
  x = sat_add(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false));
  x = sat_add(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false));
  x = sat_add(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false));
  x = sat_add(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false));
  x = sat_add(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false));
  x = sat_add(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false));
  return x;
}
__device__ __forceinline__ uint32_t dpp_sum(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)dpp_scan_add(x), 63); }
// (64-bit values in three limbs of 24, 24 and 16 bits: no limb's sum over 64 lanes carries)
__device__ __forceinline__ unsigned long long wave_add64(unsigned long long v) {
  const unsigned long long lo = dpp_sum((uint32_t)v & 0xFFFFFFu), mid = dpp_sum((uint32_t)(v >> 24) & 0xFFFFFFu), hi = dpp_sum((uint32_t)(v >> 48));
  return lo + (mid << 24) + (hi << 48);
}

// (l_gi, l_dm, l_ds, l_sk: copies of the per-gap words in LDS when one workgroup does the scan as well, else null.
// red: [16][8] sums per wave.  Returns, in every thread, the number of gaps the segment tier did not finish.)
__device__ __forceinline__ uint32_t d3_classify_body(const D3Params& P, const D3Work& W, const GapOut* __restrict__ outs,
                                                     const D3Gap* __restrict__ dgaps, uint32_t first, uint32_t stride,
                                                     uint32_t* l_gi, uint32_t* l_dm, uint32_t* l_ds, int32_t* l_sk,
                                                     unsigned long long* red, bool one_wg) {
  const uint32_t n = P.n;
  unsigned long long xA = 0, sA = 0, xB = 0, sB = 0, xD = 0, sD = 0, segs = 0;
  uint32_t unhandled = 0, seg_gaps = 0, big = 0, traced = 0, spec = 0;
  for (uint32_t i = first; i < n; i += stride) {
    // (the descriptor comes over the link on a short list, the record from device memory: asked for together)
    const D3Gap dg = dgaps[i];
    const GapOut& go = outs[i];
    const uint32_t flags = go.flags, n_states = go.n_states, n_right = go.n_right, x_right = go.x_right, x_left = go.x_left;
    const uint32_t x_sub = go.x_sub, n_sub = go.n_sub, nseg_b = go.stat[3], dflags = go.dflags, stop0 = go.stop[0], stop1 = go.stop[1];
    const int c_count = go.c_count, n_len = go.n_len, len0 = go.len[0], len1 = go.len[1], reached_j = go.reached_j, count_s = go.count_s;
    const uint32_t n_xl = go.n_xl;
    const uint64_t sub_off = go.sub_off;
    uint32_t gi = 0, dmin = 0, spread = 0;
    if (dg.kind != 0) gi = GI_BAD;
    else {
      if (flags & (G2S_DEV_OVERFLOW_A | G2S_DEV_OVERFLOW_B)) unhandled++;
      else if ((uint64_t)n_states > P.max_states || (uint64_t)n_right > P.max_states) gi = GI_MEM;
      else {
        xA += x_right; sA += n_right; xB += x_left; sB += n_states; xD += x_sub; sD += n_sub;
        segs += nseg_b;
        seg_gaps++;
        if (flags & G2S_DEV_BIG) big++;
        if (dflags & G2S_DEVA_TRACED) traced++;
        if (dflags & G2S_DEVA_SPEC) spec++;
        const bool phase_d = c_count > 0 && n_len > 0;  // :1169
        // (a gap listed for g2s_d2_* is not the host's yet: that kernel may still be running — the hand-off decides)
        const bool by_host = phase_d && !(dflags & (G2S_DEVA_ANALYSED | G2S_DEVA_D2_PENDING));
        // (the all-paths recount — the sum of the counts of the sink states — is the kernel's for every closure,
        // also for those the host analyses: whether a gap counts as filled decides the skip rule of the next)
        const int cnt = (phase_d && !P.skip_confident && P.all_paths) ? count_s : c_count;
        if (cnt > 0 && (!P.unique_paths || cnt == 1)) gi |= GI_FILLED;
        if (n_len > 0) gi |= ((uint32_t)reached_j & 0xFFu) << 8;
        if (phase_d) {
          gi |= GI_PHASE_D;
          if (by_host) gi |= GI_HOST;
          int lo_d = 0x7FFFFFFF, hi_d = 0;
#pragma unroll
          for (int q = 0; q < 2; q++) {
            if (q >= n_len) continue;
            // stop depths behind traceback start q: lowest | highest << 16 (fill_seg.hip); a traceback stops at a
            // left-flank k-mer, i.e. at a depth in [0, lmf]
            const uint32_t sw = q ? stop1 : stop0;
            const int lq = q ? len1 : len0;
            int lo = (int)(sw & 0xFFFFu), hi = (int)(sw >> 16);
            if (lo > hi || hi > (int)dg.lmf) { lo = 0; hi = (int)dg.lmf; }
            lo_d = min(lo_d, 1 + lq - hi);
            hi_d = max(hi_d, 1 + lq - lo);
          }
          lo_d = max(lo_d, 1);
          hi_d = max(hi_d, lo_d);
          dmin = (uint32_t)lo_d;
          spread = (uint32_t)(hi_d - lo_d);
          gi |= spread ? 2u : 1u;
        }
      }
    }
    gi |= (uint32_t)dg.lmf << 16;
    W.ginfo[i] = gi;
    W.dmin[i] = dmin;
    W.dspread[i] = spread;
    W.skip[i] = dg.skip_thr;
    if (l_gi) { l_gi[i] = gi; l_dm[i] = dmin; l_ds[i] = spread; l_sk[i] = dg.skip_thr; }
    // the half of the trace kernel's record that does not depend on the offsets
    uint4 h0;
    const uint64_t ao = P.arena_base + dg.arena_off, sa = (uint64_t)(i / P.group_size) * P.sub_region + sub_off;
    h0.x = (uint32_t)ao; h0.y = (uint32_t)(ao >> 32); h0.z = (uint32_t)sa; h0.w = (uint32_t)(sa >> 32);
    ((uint4*)&W.tdesc[i])[0] = h0;
    (void)n_xl;
  }
  // ---- the list's counters: per wave, per workgroup through LDS, then one addition per workgroup (a short
  // list's single workgroup stores them: 144 atomics on two lines were most of its 10 us)
  xA = wave_add64(xA); sA = wave_add64(sA); xB = wave_add64(xB); sB = wave_add64(sB); xD = wave_add64(xD); sD = wave_add64(sD);
  segs = wave_add64(segs);
  const unsigned long long cnts = wave_add64(((unsigned long long)unhandled << 32) | seg_gaps);
  big = dpp_sum(big);
  if ((threadIdx.x & 63u) == 0u && big) atomicAdd(&W.sum->big_gaps, big);
  traced = dpp_sum(traced);
  if ((threadIdx.x & 63u) == 0u && traced) atomicAdd(&W.sum->traced_gaps, traced);
  spec = dpp_sum(spec);
  if ((threadIdx.x & 63u) == 0u && spec) atomicAdd(&W.sum->spec_gaps, spec);
  const uint32_t wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  if ((threadIdx.x & 63u) == 0u) {
    unsigned long long* r = red + wave * 8u;
    r[0] = xA; r[1] = sA; r[2] = xB; r[3] = sB; r[4] = xD; r[5] = sD; r[6] = segs; r[7] = cnts;
  }
  __syncthreads();
  unsigned long long tot = 0;
  if (threadIdx.x < 8u) for (uint32_t w = 0; w < nwaves; w++) tot += red[w * 8u + threadIdx.x];
  unsigned long long all_cnts = 0;
  for (uint32_t w = 0; w < nwaves; w++) all_cnts += red[w * 8u + 7u];
  D3Summary* S = W.sum;
  if (threadIdx.x < 7u) {
    unsigned long long* dst = (unsigned long long*)&S->xA + threadIdx.x;  // xA sA xB sB xD sD segs are consecutive
    if (one_wg) *dst = tot; else if (tot) atomicAdd(dst, tot);
  } else if (threadIdx.x == 7u) {
    if (one_wg) { S->unhandled = (uint32_t)(tot >> 32); S->seg_gaps = (uint32_t)tot; }
    else {
      if (tot >> 32) atomicAdd(&S->unhandled, (uint32_t)(tot >> 32));
      if (tot & 0xFFFFFFFFull) atomicAdd(&S->seg_gaps, (uint32_t)tot);
    }
  }
  return (uint32_t)(all_cnts >> 32);
}
__global__ __launch_bounds__(256) void g2s_d3_classify(const D3Params P, const D3Work W, const GapOut* __restrict__ outs,
                                                       const D3Gap* __restrict__ dgaps) {
  __shared__ unsigned long long red[4 * 8];
  (void)d3_classify_body(P, W, outs, dgaps, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, nullptr, nullptr, nullptr, nullptr, red, false);
}

// ---------------------------------------------------------------------------------------------------------
// scan: the skip rule, prefix sums in list order, table layout.  One workgroup of 1024 threads over the compact
// per-gap arrays; thread t owns the contiguous range [t * per, (t + 1) * per) of the list.
// ---------------------------------------------------------------------------------------------------------
// exclusive prefix sums of (a, b, c) over the 1024 threads of the workgroup; totals in tot[].  One barrier: every
// wave scans the 16 wave totals itself.  sh: [16][3] 64-bit words, not in use by a previous call still being read
// (the callers alternate between two).
__device__ __forceinline__ void block_scan3(uint64_t& a, uint64_t& b, uint32_t& c, uint64_t* sh /* [48] */, uint64_t* tot_a, uint64_t* tot_b,
                                            uint32_t* tot_c) {
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  uint64_t ia = a, ib = b;
  uint32_t ic = c;
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t ya = __shfl_up(ia, o), yb = __shfl_up(ib, o);
    const uint32_t yc = __shfl_up(ic, o);
    if (lane >= o) { ia += ya; ib += yb; ic += yc; }
  }
  if (lane == 63) { sh[wave * 3] = ia; sh[wave * 3 + 1] = ib; sh[wave * 3 + 2] = ic; }
  __syncthreads();
  uint64_t wa = lane < 16 ? sh[lane * 3] : 0, wb = lane < 16 ? sh[lane * 3 + 1] : 0, wc = lane < 16 ? sh[lane * 3 + 2] : 0;
  for (int o = 1; o < 16; o <<= 1) {
    const uint64_t ya = __shfl_up(wa, o), yb = __shfl_up(wb, o), yc = __shfl_up(wc, o);
    if (lane >= o) { wa += ya; wb += yb; wc += yc; }
  }
  *tot_a = __shfl(wa, 15); *tot_b = __shfl(wb, 15); *tot_c = (uint32_t)__shfl(wc, 15);
  const uint64_t oa = wave ? __shfl(wa, wave - 1) : 0, ob = wave ? __shfl(wb, wave - 1) : 0, oc = wave ? __shfl(wc, wave - 1) : 0;
  a = oa + ia - a; b = ob + ib - b; c = (uint32_t)oc + ic - c;
}

// the same for a short list's sums, which fit 32 bits (a and b cannot reach 2^32 with 3 072 gaps; c — table
// entries — can: it saturates, and a saturated total is over every budget)
__device__ __forceinline__ void block_scan3_short(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t* sh /* [48] */, uint32_t* tot_a, uint32_t* tot_b,
                                                  uint32_t* tot_c) {
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  const uint32_t ia = dpp_scan_add(a), ib = dpp_scan_add(b), ic = dpp_scan_sat(c);
  if (lane == 63) { sh[wave * 3] = ia; sh[wave * 3 + 1] = ib; sh[wave * 3 + 2] = ic; }
  __syncthreads();
  uint32_t wa = lane < 16 ? sh[lane * 3] : 0u, wb = lane < 16 ? sh[lane * 3 + 1] : 0u, wc = lane < 16 ? sh[lane * 3 + 2] : 0u;
  wa = dpp_scan_add(wa); wb = dpp_scan_add(wb); wc = dpp_scan_sat(wc);  // (lanes 16 and up hold the total)
  *tot_a = (uint32_t)__builtin_amdgcn_readlane((int)wa, 15); *tot_b = (uint32_t)__builtin_amdgcn_readlane((int)wb, 15);
  *tot_c = (uint32_t)__builtin_amdgcn_readlane((int)wc, 15);
  const int wp = (int)uni((uint32_t)max(wave - 1, 0));
  const uint32_t oa = wave ? (uint32_t)__builtin_amdgcn_readlane((int)wa, wp) : 0u, ob = wave ? (uint32_t)__builtin_amdgcn_readlane((int)wb, wp) : 0u;
  const uint32_t oc = wave ? (uint32_t)__builtin_amdgcn_readlane((int)wc, wp) : 0u;
  a = oa + ia - a; b = ob + ib - b;
  // (exclusive value of a saturating scan: the inclusive one without this thread's share, unless it saturated)
  const uint32_t incl_c = sat_add(oc, ic);
  c = incl_c == 0xFFFFFFFFu ? 0xFFFFFFFFu : incl_c - c;
}

// (gi_a, dm_a, ds_a, sk_a: the per-gap words — W's arrays, or their copies in LDS when this workgroup classified
// the gaps itself (`dual`: what the skip rule changes is then written to both).  unhandled: gaps the segment tier
// did not finish.)
__device__ __forceinline__ void d3_scan_body(const D3Params& P, const D3Work& W, const GapOut* __restrict__ outs, uint32_t* gi_a,
                                             uint32_t* dm_a, uint32_t* ds_a, const int32_t* sk_a, bool dual, uint32_t unhandled,
                                             uint64_t* sh /* [96] */, uint32_t* sh_f /* [1024] */, bool short_list /* <= 3 072 gaps */) {
  const uint32_t t = threadIdx.x, n = P.n;
  const uint32_t per = (n + 1023u) / 1024u;
  const uint32_t lo = min(n, t * per), hi = min(n, lo + per);
  // ---- the skip rule (:369: a gap is not attempted when the previous gap of its record was filled with a right
  // fuz beyond the distance between them): skipped_i = s_i && !skipped_(i-1), s_i from gap i-1's own result
  if (P.has_skip) {
    auto s_of = [&](uint32_t i) -> bool {
      if (i == 0) return false;
      const int thr = sk_a[i];
      if (thr < 0) return false;
      const uint32_t pg = gi_a[i - 1];
      return (pg & GI_FILLED) && !(pg & (GI_BAD | GI_MEM)) && (int)((pg >> 8) & 0xFFu) > thr;
    };
    // the range as a function of "the gap in front of it was skipped": results for both inputs
    bool o0 = false, o1 = true;
    for (uint32_t i = lo; i < hi; i++) { const bool s = s_of(i); o0 = s && !o0; o1 = s && !o1; }
    sh_f[t] = (o0 ? 1u : 0u) | (o1 ? 2u : 0u);
    __syncthreads();
    if (t < 64u) {  // 1024 one-bit functions composed by one wave: 16 each, then a scan over the lanes
      auto apply = [](uint32_t f, bool in) -> bool { return in ? (f & 2u) != 0 : (f & 1u) != 0; };
      uint32_t f = 2u;  // identity
      for (uint32_t q = 0; q < 16u; q++) { const uint32_t g = sh_f[t * 16u + q]; f = (apply(g, apply(f, false)) ? 1u : 0u) | (apply(g, apply(f, true)) ? 2u : 0u); }
      uint32_t inc = f;  // inclusive composition over lanes 0 .. t
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t pv = (uint32_t)__shfl_up((int)inc, o);
        if ((int)t >= o) inc = (apply(inc, apply(pv, false)) ? 1u : 0u) | (apply(inc, apply(pv, true)) ? 2u : 0u);
      }
      uint32_t before = (uint32_t)__shfl_up((int)inc, 1);
      if (t == 0) before = 2u;
      bool cur = apply(before, false);  // nothing in front of the list
      for (uint32_t q = 0; q < 16u; q++) { const uint32_t g = sh_f[t * 16u + q]; sh_f[t * 16u + q] = cur ? 1u : 0u; cur = apply(g, cur); }
    }
    __syncthreads();
    bool sk = sh_f[t] != 0;
    // (a thread's verdicts read the gap in front of its range, which its neighbour may be rewriting: the class bits
    // only, which s_of does not look at)
    for (uint32_t i = lo; i < hi; i++) {
      sk = s_of(i) && !sk;
      if (sk) {
        const uint32_t g = (gi_a[i] & ~3u) | GI_SKIPPED;
        gi_a[i] = g; dm_a[i] = 0; ds_a[i] = 0;
        if (dual) { W.ginfo[i] = g; W.dmin[i] = 0; W.dspread[i] = 0; }
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  // ---- prefix sums in list order: draws, draw-dependent gaps, spreads.  A thread's range is read eight gaps at a
  // time with all 24 loads in flight (one gap per iteration, load then use, made every pass a chain of memory round
  // trips: the scan took 26 us of a 10 000-gap list's 1 ms).
  auto load8 = [&](uint32_t i0, uint32_t* g8, uint32_t* m8, uint32_t* s8) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint32_t i = i0 + (uint32_t)q;
      const bool have = i < hi;
      g8[q] = have ? gi_a[i] : GI_SKIPPED;  // (beyond the range: reads as a gap that contributes nothing)
      m8[q] = have ? dm_a[i] : 0u;
      s8[q] = have ? ds_a[i] : 0u;
    }
  };
  if (t == 0) stamp(W, 20);
  uint64_t sd = 0, ss = 0;
  uint32_t nv = 0;
  for (uint32_t i0 = lo; i0 < hi; i0 += 8u) {
    uint32_t g8[8], m8[8], s8[8];
    load8(i0, g8, m8, s8);
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (g8[q] & GI_SKIPPED) continue;
      sd += m8[q];
      if (GI_CLASS(g8[q]) == 2u) { nv++; ss += s8[q]; }
    }
  }
  uint64_t tot_d, tot_s;
  uint32_t V;
  if (short_list) {
    uint32_t a = (uint32_t)sd, b = (uint32_t)ss, ta, tb;
    block_scan3_short(a, b, nv, (uint32_t*)sh, &ta, &tb, &V);
    sd = a; ss = b; tot_d = ta; tot_s = tb;
  } else block_scan3(sd, ss, nv, sh, &tot_d, &tot_s, &V);
  if (t == 0) stamp(W, 21);
  const bool too_wide = tot_d + tot_s + P.base0 + P.R0 >= 0xFFFF0000ull || tot_s + P.R0 >= (uint64_t)G2S_D3_TABLE_BUDGET;
  uint64_t my_tab = 0, my_blk = 0;
  uint32_t my_tiles = 0;
  {
    uint64_t d = sd + P.base0, r = ss + P.R0;  // (a group of a sharded list: behind the groups in front of it)
    uint32_t v = nv;
    for (uint32_t i0 = lo; i0 < hi; i0 += 8u) {
      uint32_t g8[8], m8[8], s8[8];
      load8(i0, g8, m8, s8);
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint32_t i = i0 + (uint32_t)q;
        if (i >= hi) break;
        W.base[i] = (uint32_t)d;
        W.vrank[i] = v;
        if (g8[q] & GI_SKIPPED) continue;
        d += m8[q];
        if (GI_CLASS(g8[q]) == 2u) {
          W.var_gap[v] = i;
          W.var_R[v] = (uint32_t)r;
          my_tab += r + 1u;
          my_tiles += (uint32_t)((r + D3_TILE) / D3_TILE);
          if (v % G2S_D3_BLOCK_VARS == 0u) my_blk += r + 1u;
          v++;
          r += s8[q];
        }
      }
    }
    if (t == 1023u) W.var_R[V] = (uint32_t)tot_s + P.R0;
  }
  // ---- where every table starts
  if (t == 0) stamp(W, 22);
  uint64_t to = my_tab, bo = my_blk, T, TB;
  uint32_t tl = my_tiles, tiles;
  if (short_list) {
    uint32_t b = (uint32_t)std::min<uint64_t>(bo, 0xFFFFFFFFull), c = (uint32_t)std::min<uint64_t>(to, 0xFFFFFFFFull), tb, tc;
    block_scan3_short(tl, b, c, (uint32_t*)(sh + 48), &tiles, &tb, &tc);
    bo = b; to = c; TB = tb; T = tc == 0xFFFFFFFFu ? ~0ull : tc;
  } else block_scan3(to, bo, tl, sh + 48, &T, &TB, &tiles);
  const bool over = too_wide || T > (uint64_t)G2S_D3_TABLE_BUDGET || TB > (uint64_t)(G2S_D3_TABLE_BUDGET / 4u);
  if (!over) {
    uint64_t d = sd + P.base0, r = ss + P.R0;
    uint32_t v = nv;
    for (uint32_t i0 = lo; i0 < hi; i0 += 8u) {
      uint32_t g8[8], m8[8], s8[8];
      load8(i0, g8, m8, s8);
#pragma unroll
      for (int q = 0; q < 8; q++) {
        if (g8[q] & GI_SKIPPED) continue;
        const uint32_t base_i = (uint32_t)d;
        d += m8[q];
        if (GI_CLASS(g8[q]) != 2u) continue;
        W.var_toff[v] = (uint32_t)to;
        W.var_tile[v] = tl;
        {  // the half of the gap's record for g2s_d3_tables that this pass knows (the other half: below, a thread a gap)
          const uint32_t i = i0 + (uint32_t)q;
          uint4 h0, h1;
          h0.x = i; h0.y = (uint32_t)r; h0.z = (uint32_t)to; h0.w = tl;
          h1.x = base_i; h1.y = m8[q]; h1.z = s8[q]; h1.w = 0u;
          ((uint4*)&W.vdesc[v])[0] = h0;
          ((uint4*)&W.vdesc[v])[1] = h1;
        }
        tl += (uint32_t)((r + D3_TILE) / D3_TILE);
        to += r + 1u;
        if (v % G2S_D3_BLOCK_VARS == 0u) { W.blk_toff[v / G2S_D3_BLOCK_VARS] = (uint32_t)bo; bo += r + 1u; }
        v++;
        r += s8[q];
      }
    }
    if (t == 1023u) { W.var_toff[V] = (uint32_t)T; W.var_tile[V] = tiles; W.blk_toff[(V + G2S_D3_BLOCK_VARS - 1u) / G2S_D3_BLOCK_VARS] = (uint32_t)TB; }
    // ---- the other half of every draw-dependent gap's record — what its GapOut record holds — and the gap of each of
    // its tiles: a thread a gap, all their loads in flight at once.  (Inside the pass above, the thread that owns a
    // range took its gaps one by one, a round trip to memory each: 17 of the 44 us this kernel took on a 10 000-gap list.)
    __threadfence_block();
    __syncthreads();
    for (uint32_t v = t; v < V; v += 1024u) {
      const uint4 h0 = ((const uint4*)&W.vdesc[v])[0];
      const uint32_t i = h0.x, r = h0.y, tl0 = h0.w;
      const GapOut& go = outs[i];
      uint4 h2, h3;
      const uint64_t sa = (uint64_t)(i / P.group_size) * P.sub_region + go.sub_off;
      // (D3Var: gap R toff tile0 | base dmin dspread ns | sub_at start_seg start_t | len0 len1 n_len pad)
      ((uint32_t*)&W.vdesc[v])[7] = go.n_xl;
      h2.x = (uint32_t)sa; h2.y = (uint32_t)(sa >> 32); h2.z = go.start_seg; h2.w = go.start_t;
      h3.x = (uint32_t)go.len[0]; h3.y = (uint32_t)go.len[1]; h3.z = (uint32_t)go.n_len; h3.w = 0u;
      ((uint4*)&W.vdesc[v])[2] = h2;
      ((uint4*)&W.vdesc[v])[3] = h3;
      const uint32_t nt = (r + D3_TILE) / D3_TILE;
      for (uint32_t x = 0; x < nt; x++) W.tile_var[tl0 + x] = v;
    }
  }
  if (t == 0) {
    stamp(W, 23);
    D3Summary* S = W.sum;
    S->status = (unhandled ? G2S_D3_UNHANDLED : 0u) | (over ? G2S_D3_BUDGET : 0u);
    S->n_var = V;
    S->tiles = over ? 0u : tiles;
    S->table_entries = T;
    S->block_entries = TB;
    S->draws_min = tot_d;
    S->draws_spread = tot_s;
  }
}
// (the per-gap words of a list of up to D3_SCAN_LDS_GAPS gaps come into LDS first, 1 024 consecutive words an
// instruction: the body's three passes give every thread a contiguous range of the list, and read from device memory
// that way — every lane its own cache lines — each pass was a chain of round trips: 13 of this kernel's 44 us on a
// 10 000-gap list.  Dynamic LDS: 12 bytes a gap, or none.)
#define D3_SCAN_LDS_GAPS 12288u
__global__ __launch_bounds__(1024) void g2s_d3_scan(const D3Params P, const D3Work W, const GapOut* __restrict__ outs) {
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
  __shared__ uint64_t sh[96];
  __shared__ uint32_t sh_f[1024];
  // (the 32-bit block scans: a list in this mode has at most 20 480 gaps of at most 12 000 draws each)
  if (P.n <= D3_SCAN_LDS_GAPS) {
    uint32_t *l_gi = dyn, *l_dm = dyn + P.n, *l_ds = dyn + 2u * P.n;
    for (uint32_t i = threadIdx.x; i < P.n; i += 1024u) { l_gi[i] = W.ginfo[i]; l_dm[i] = W.dmin[i]; l_ds[i] = W.dspread[i]; }
    const uint32_t unhandled = W.sum->unhandled;
    __syncthreads();
    d3_scan_body(P, W, outs, l_gi, l_dm, l_ds, W.skip, true, unhandled, sh, sh_f, true);
    return;
  }
  d3_scan_body(P, W, outs, W.ginfo, W.dmin, W.dspread, W.skip, false, W.sum->unhandled, sh, sh_f, P.n <= 262144u);
}

// ---------------------------------------------------------------------------------------------------------
// the rand() stream.  W[0 .. 31) is the generator's state in front of the next value, W[31 + k] the k-th value
// the session will draw (raw words: rand() returns word >> 1).  The host hands over W[0 .. G2S_RAND_WINDOW).
// ---------------------------------------------------------------------------------------------------------
// The window of a stream that continues another list's: W[0 .. 31) = the state that list's kernels left behind its
// last draw, W[31 + i] = W[i] + W[28 + i] (glibc TYPE_3: r[i] = r[i-31] + r[i-3]) — three values a step.
__global__ __launch_bounds__(64) void g2s_rand_window(uint32_t* __restrict__ Wd, const uint32_t* __restrict__ link) {
  __shared__ uint32_t w[G2S_RAND_WINDOW + 2];
  const uint32_t lane = threadIdx.x;
  if (lane < 31u) w[lane] = link[lane];
  __syncthreads();
  for (uint32_t i = 31u; i < G2S_RAND_WINDOW; i += 3u) {
    if (lane < 3u) w[i + lane] = w[i + lane - 31u] + w[i + lane - 3u];
    __syncthreads();
  }
  for (uint32_t i = lane; i < G2S_RAND_WINDOW; i += 64u) Wd[i] = w[i];
}
__global__ __launch_bounds__(64) void g2s_rand_fill(uint32_t* __restrict__ Wd, const g2s::RandTables rt, const D3Summary* sum,
                                                    uint64_t capacity, uint64_t first_block) {
  __shared__ uint32_t base[G2S_RAND_WINDOW], s1[96], s2[64];
  // (sum: only as far as the list can draw, known once its scan has run; nullptr: the whole capacity, for a launch
  // beside the fill kernel)
  if (sum && sum->status) return;
  const uint64_t need = sum ? min(capacity, sum->draws_min + sum->draws_spread + 64ull) : capacity;
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = lane; i < G2S_RAND_WINDOW; i += 64u) base[i] = Wd[i];
  __syncthreads();
  const uint64_t nblocks = (need + G2S_RAND_BLOCK - 1u) / G2S_RAND_BLOCK;
  // (first_block: a group of a sharded list generates the stream from where its own gaps can first draw)
  for (uint64_t B = first_block + blockIdx.x; B < nblocks; B += gridDim.x) {
    const uint32_t* ph = rt.hi + (size_t)((B >> 8) & 127u) * 31u;
    const uint32_t* pm = rt.mid + (size_t)(B & 255u) * 31u;
    const uint32_t* pl = rt.lane + (size_t)lane * 31u;
    // W[2^20 a + i], i < 91
    for (uint32_t i = lane; i < 91u; i += 64u) {
      uint32_t a = 0;
#pragma unroll
      for (int j = 0; j < 31; j++) a += ph[j] * base[i + (uint32_t)j];
      s1[i] = a;
    }
    __syncthreads();
    // W[4096 B + i], i < 61
    if (lane < 61u) {
      uint32_t a = 0;
#pragma unroll
      for (int j = 0; j < 31; j++) a += pm[j] * s1[lane + (uint32_t)j];
      s2[lane] = a;
    }
    __syncthreads();
    // the lane's state W[4096 B + 64 l + i], i < 31, then its 64 values
    uint32_t c[31], r[31];
#pragma unroll
    for (int j = 0; j < 31; j++) c[j] = pl[j];
#pragma unroll
    for (int i = 0; i < 31; i++) {
      uint32_t a = 0;
#pragma unroll
      for (int j = 0; j < 31; j++) a += c[j] * s2[i + j];
      r[i] = a;
    }
    uint32_t* out = Wd + 31u + B * G2S_RAND_BLOCK + (uint64_t)lane * 64u;
    const uint64_t room = capacity > B * G2S_RAND_BLOCK + (uint64_t)lane * 64u ? capacity - (B * G2S_RAND_BLOCK + (uint64_t)lane * 64u) : 0ull;
#pragma unroll
    for (int q = 0; q < 64; q++) {
      // ring: value q is r[q mod 31] := r[q mod 31] (31 back) + r[(q + 28) mod 31] (3 back)
      const uint32_t v = r[q % 31] + r[(q + 28) % 31];
      r[q % 31] = v;
      if ((uint64_t)q < room) out[q] = v;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// walks over the closure segments (post.cpp: seg_count_draws / seg_traceback are the host versions)
// ---------------------------------------------------------------------------------------------------------
struct SegW { uint32_t node, depth_len, cnt, ts_tt, par01, par23, flags, pad; };

// number of parents of a closure segment (GATB predecessor order: slots that hold a segment id)
__device__ __forceinline__ int seg_nparents(uint32_t p01, uint32_t p23) {
  return ((p01 & 0xFFFFu) != 0xFFFFu) + ((p01 >> 16) != 0xFFFFu) + ((p23 & 0xFFFFu) != 0xFFFFu) + ((p23 >> 16) != 0xFFFFu);
}
// its sel-th parent
__device__ __forceinline__ uint32_t seg_parent(uint32_t p01, uint32_t p23, int sel) {
  const uint32_t ps[4] = {p01 & 0xFFFFu, p01 >> 16, p23 & 0xFFFFu, p23 >> 16};
  uint32_t chosen = 0xFFFFu;
  int idx = 0;
#pragma unroll
  for (int q = 0; q < 4; q++)
    if (ps[q] != 0xFFFFu) { if (idx == sel) chosen = ps[q]; idx++; }
  return chosen;
}

// rand() % nb for the 2 to 4 parents a segment can have (a division by a variable is ~40 instructions)
__device__ __forceinline__ int pick_parent(uint32_t rv, int nb) { return nb == 2 ? (int)(rv & 1u) : nb == 3 ? (int)(rv % 3u) : (int)(rv & 3u); }

// What a walk needs of a closure segment: entry depth | length << 16, the parents, the flags.
struct SegLite { uint32_t depth_len, par01, par23, flags; };

__device__ __forceinline__ SegLite seg_lite(const SegLite* p, int i) { return p[i]; }
__device__ __forceinline__ SegLite seg_lite(const SegW* p, int i) {  // (a closure too large for LDS: from the records themselves)
  SegLite l;
  l.depth_len = p[i].depth_len; l.par01 = p[i].par01; l.par23 = p[i].par23; l.flags = p[i].flags;
  return l;
}
// rand() draws of the traceback of one gap whose draws start at rnd[0]; *bad: something the host must look at
template <typename SegT>
__device__ int d3_walk_count(int n_len, int len0, int len1, uint32_t start_seg, uint32_t start_t, int nsegs,
                             const SegT* __restrict__ segs, const uint32_t* lwin /* LDS: the values from the first draw on */,
                             uint32_t nwin /* how many of them are there */, uint64_t avail /* values that exist from the first on */,
                             bool* bad) {
  int draws = 1;
  if (avail < 1 || nwin < 1) { *bad = true; return draws; }
  const int pick = n_len > 1 ? (int)((lwin[0] >> 1) & 1u) : 0;  // :1440 (n_len <= 2)
  int d2 = pick ? len1 : len0;
  const uint32_t sg = (start_seg >> (16 * pick)) & 0xFFFFu;
  int i = sg == 0xFFFFu ? -1 : (int)sg;
  int t = (int)((start_t >> (16 * pick)) & 0xFFFFu);
  if (i < 0 || i >= nsegs) { *bad = true; return draws; }
  SegLite s = seg_lite(segs, i);
  bool ended = false;
  for (int guard = 0; d2 >= 0; guard++) {
    if (guard > 70000) { *bad = true; break; }
    const int run = min(t, d2);  // the states t .. 1 of this segment: a draw each
    draws += run; d2 -= run; t -= run;
    if (t == 0 && (s.flags & G2S_SUB_SOURCE)) { ended = true; break; }  // :1455-1462
    if (d2 > 0) {  // (then t == 0: the segment's first state, on to a parent)
      const int nb = seg_nparents(s.par01, s.par23);
      if (nb == 0 || (nb > 1 && !(s.flags & G2S_SEG_ORDERED)) || (uint64_t)draws >= avail || (uint32_t)draws >= nwin) { *bad = true; break; }
      const uint32_t rv = nb > 1 ? lwin[draws] >> 1 : 0u;  // (rand() % 1: the value does not matter)
      draws++;
      i = (int)seg_parent(s.par01, s.par23, nb == 1 ? 0 : pick_parent(rv, nb));  // :1513, GATB predecessor order
      if (i >= nsegs) { *bad = true; break; }
      s = seg_lite(segs, i);
      t = (int)(s.depth_len >> 16) - 1;
    }
    d2--;
  }
  if (!ended) *bad = true;  // (the host path decides what an unfinished walk means, :1493-1510)
  return draws;
}

// The same count for a closure whose links are in LDS, as the trace kernel's walk goes (see there): every state passed
// draws once, so the draw made at a segment's first state, at depth d0, is the (1 + len - d0)-th of the gap whatever
// the path, and a walk that ends at a source at depth d0 has drawn 1 + len - d0 times — one record and one rand()
// value per segment entered, no counters carried along.  (The depths are the records': a closure whose segments do
// not follow each other as their depths say is the trace kernel's to find — it checks every hop of the walk that is
// taken — and the host's to decide.)
__device__ int d3_walk_count_lite(int n_len, int len0, int len1, uint32_t start_seg, int nsegs, const SegLite* __restrict__ segs,
                                  const uint32_t* lwin, uint32_t nwin, uint64_t avail, bool* bad) {
  if (avail < 1 || nwin < 1) { *bad = true; return 1; }
  const int pick = n_len > 1 ? (int)((lwin[0] >> 1) & 1u) : 0;  // :1440 (n_len <= 2)
  const int len = pick ? len1 : len0;
  const uint32_t sg = (start_seg >> (16 * pick)) & 0xFFFFu;
  int i = sg == 0xFFFFu ? -1 : (int)sg;
  if (i < 0 || i >= nsegs || len < 0) { *bad = true; return 1; }
  for (int hop = 0; hop <= nsegs; hop++) {  // (a traceback descends: it enters a segment once)
    const SegLite s = segs[i];
    const int d0 = (int)(s.depth_len & 0xFFFFu);
    if (d0 > len) break;
    if (s.flags & G2S_SUB_SOURCE) return 1 + len - d0;  // :1455-1462
    if (d0 == 0) break;  // (no depth below: the walk ends without a source, :1493-1510)
    const int nb = seg_nparents(s.par01, s.par23);
    const uint32_t at = (uint32_t)(1 + len - d0);
    if (nb == 0 || (nb > 1 && !(s.flags & G2S_SEG_ORDERED)) || (uint64_t)at >= avail || at >= nwin) break;
    const uint32_t rv = nb > 1 ? lwin[at] >> 1 : 0u;  // (rand() % 1: the value does not matter)
    i = (int)seg_parent(s.par01, s.par23, nb == 1 ? 0 : pick_parent(rv, nb));  // :1513, GATB predecessor order
    if (i >= nsegs) break;
  }
  *bad = true;
  return 1;
}

// draws of draw-dependent gap v when its draws start at base + d, for every d it can meet: a workgroup takes 256
// consecutive deviations of one gap: the closure's links in LDS, one walk per thread.  What it needs to know comes
// in three round trips: the tile's gap (tile_var), that gap's record (D3Var), then the closure's links and the
// rand() values the tile's walks can read — both into LDS: a walk asks for a value at every choice between
// parents, and a round trip to memory per choice was the kernel's time (20 us for a 30-segment path).
// (the window: map_cap + 256 words of dynamic LDS — the longest path of the list, 256 starts)
#define D3_TAB_SEGS 512u
__global__ __launch_bounds__(256) void g2s_d3_tables(const D3Params P, const D3Work W, const SubRec* __restrict__ sub,
                                                     const uint32_t* __restrict__ rnd /* first upcoming value */, uint64_t capacity) {
  __shared__ SegLite lseg[D3_TAB_SEGS];
  extern __shared__ __attribute__((aligned(16))) uint32_t lwin[];
  const uint32_t win_cap = P.map_cap + D3_TILE;
  const D3Summary* S = W.sum;
  const bool st0 = blockIdx.x == 0 && threadIdx.x == 0;
  if (st0) stamp(W, 3);
  const uint32_t status = S->status, tiles = S->tiles;
  const uint32_t tid = threadIdx.x;
  if (status) return;
  if (st0) stamp(W, 4);
  bool bad_any = false;
  for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const uint32_t v = W.tile_var[tile];
    const g2s::D3Var vd = W.vdesc[v];
    const uint32_t ns = vd.ns;
    const uint32_t d0 = (tile - vd.tile0) * D3_TILE, d = d0 + tid;
    const uint64_t at0 = (uint64_t)vd.base + d0;  // the tile's first value; thread tid starts tid values further on
    const bool mine = d <= vd.R;
    // a walk from deviation d reads values [d, d + dmin + dspread) of the tile's window
    const uint32_t want = min(win_cap, min(vd.R - d0, D3_TILE - 1u) + vd.dmin + vd.dspread + 1u);
    if (st0) stamp(W, 5);
    __syncthreads();  // (the previous tile's walks are over)
    {
      const SegW* gs = (const SegW*)(sub + vd.sub_at);
      for (uint32_t q = tid; q < ns && q < D3_TAB_SEGS; q += 256u) {
        SegLite l;
        l.depth_len = gs[q].depth_len; l.par01 = gs[q].par01; l.par23 = gs[q].par23; l.flags = gs[q].flags;
        lseg[q] = l;
      }
      for (uint32_t x = tid; x < want; x += 256u) lwin[x] = at0 + x < capacity ? rnd[at0 + x] : 0u;
    }
    __syncthreads();
    if (st0) stamp(W, 6);
    if (mine) {
      // (the closures of deep searches — thousands of segments, the large variant's — are walked where they lie)
      bool bad = false;
      const uint64_t at = at0 + tid;
      const uint32_t nw = want > tid ? want - tid : 0u;
      const uint64_t av = capacity > at ? capacity - at : 0ull;
      const int draws = ns > D3_TAB_SEGS
                            ? d3_walk_count(vd.n_len, vd.len0, vd.len1, vd.start_seg, vd.start_t, (int)ns, (const SegW*)(sub + vd.sub_at), lwin + tid, nw, av, &bad)
                            : d3_walk_count_lite(vd.n_len, vd.len0, vd.len1, vd.start_seg, (int)ns, lseg, lwin + tid, nw, av, &bad);
      const int dev = draws - (int)vd.dmin;
      if (dev < 0 || dev > (int)vd.dspread) bad = true;
      W.tab[(uint64_t)vd.toff + d] = bad ? (uint16_t)0 : (uint16_t)dev;
      bad_any |= bad;
    }
  }
  if (bad_any) atomicAdd(&W.sum->anomalies, 1u);
  if (st0) stamp(W, 7);
  if (P.laps && tid == 0) {  // (G2S_DEBUG: the first workgroup's entry and the last workgroup's end, any workgroup with a tile)
    unsigned long long* laps = (unsigned long long*)((char*)W.sum + 512);
    if (blockIdx.x < tiles) atomicMax(&laps[32], wall_clock64());
    atomicMax(&laps[33], wall_clock64());
  }
}

// deviation behind block b for every deviation in front of it
__device__ __forceinline__ void d3_blocks_body(const D3Work& W, uint64_t first, uint64_t stride) {
  const D3Summary* S = W.sum;
  const uint32_t V = S->n_var;
  const uint32_t NB = (V + G2S_D3_BLOCK_VARS - 1u) / G2S_D3_BLOCK_VARS;
  const uint64_t TB = S->block_entries;
  for (uint64_t e = first; e < TB; e += stride) {
    uint32_t lo = 0, hi = NB;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)W.blk_toff[mid] <= e) lo = mid; else hi = mid; }
    const uint32_t b = lo;
    uint32_t d = (uint32_t)(e - W.blk_toff[b]);
    const uint32_t v1 = min(V, (b + 1u) * G2S_D3_BLOCK_VARS);
    for (uint32_t v = b * G2S_D3_BLOCK_VARS; v < v1; v++) d += W.tab[(uint64_t)W.var_toff[v] + d];
    W.btab[e] = d;
  }
}
__global__ __launch_bounds__(256) void g2s_d3_blocks(const D3Work W) {
  if (W.sum->status) return;
  d3_blocks_body(W, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}

// the chain over blocks, then inside every block; the list's total and the generator's state behind it
__device__ __forceinline__ void d3_chain_body(const D3Work& W, const uint32_t* __restrict__ Wd, uint32_t* total_dev /* shared */,
                                              uint32_t base0 = 0u, uint32_t d_in = 0u) {
  D3Summary* S = W.sum;
  const uint32_t V = S->n_var;
  const uint32_t NB = (V + G2S_D3_BLOCK_VARS - 1u) / G2S_D3_BLOCK_VARS;
  // (the chain over blocks is one lane's chain of dependent loads — the deviation behind a block is looked up at the
  // deviation in front of it; where the blocks' tables start does not depend on it: those words come into LDS first, all
  // at once, so that a step is ONE round trip instead of two — 19 -> ~12 us for config 3's 26 blocks)
  __shared__ uint32_t l_toff[2048], l_in[2048];
  const bool in_lds = NB <= 2048u;
  if (in_lds) for (uint32_t b = threadIdx.x; b < NB; b += blockDim.x) l_toff[b] = W.blk_toff[b];
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t d = d_in;
    if (in_lds) for (uint32_t b = 0; b < NB; b++) { l_in[b] = d; d = W.btab[(uint64_t)l_toff[b] + d]; }
    else for (uint32_t b = 0; b < NB; b++) { W.blk_in[b] = d; d = W.btab[(uint64_t)W.blk_toff[b] + d]; }
    W.dvar[V] = d;
    *total_dev = d;
  }
  __threadfence_block();
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < NB; b += blockDim.x) {
    uint32_t d = in_lds ? l_in[b] : W.blk_in[b];
    if (in_lds) W.blk_in[b] = d;
    const uint32_t v0 = b * G2S_D3_BLOCK_VARS, v1 = min(V, (b + 1u) * G2S_D3_BLOCK_VARS);
    uint32_t toff[G2S_D3_BLOCK_VARS];  // (where the block's tables start: asked for together, in front of the chain through them)
#pragma unroll
    for (uint32_t q = 0; q < G2S_D3_BLOCK_VARS; q++) toff[q] = v0 + q < v1 ? W.var_toff[v0 + q] : 0u;
#pragma unroll
    for (uint32_t q = 0; q < G2S_D3_BLOCK_VARS; q++)
      if (v0 + q < v1) { W.dvar[v0 + q] = d; d += W.tab[(uint64_t)toff[q] + d]; }
  }
  const uint64_t total = (uint64_t)base0 + S->draws_min + *total_dev;  // (draws of the list up to the end of this group)
  if (threadIdx.x == 0) S->draws_total = total;
  if (threadIdx.x < 31u) {
    const uint32_t w = Wd[total + threadIdx.x];
    S->rand_state[threadIdx.x] = w;
    if (W.link) W.link[threadIdx.x] = w;
  }
}
__global__ __launch_bounds__(1024) void g2s_d3_chain(const D3Work W, const uint32_t* __restrict__ Wd, uint32_t base0, uint32_t d_in) {
  __shared__ uint32_t total_dev;
  if (W.sum->status) return;
  d3_chain_body(W, Wd, &total_dev, base0, d_in);
}
// (sharded lists) the deviation behind this group for every deviation d = 0 .. R0 in front of it: a walk through the
// blocks' tables per d.  fn[] is pinned host memory; fn[R0 + 1] = 1 says it is complete (the stream's end does too).
__global__ __launch_bounds__(256) void g2s_d3_groupfn(const D3Work W, uint32_t R0, uint32_t* __restrict__ fn) {
  const D3Summary* S = W.sum;
  if (S->status) return;
  const uint32_t V = S->n_var;
  const uint32_t NB = (V + G2S_D3_BLOCK_VARS - 1u) / G2S_D3_BLOCK_VARS;
  for (uint32_t d0 = blockIdx.x * blockDim.x + threadIdx.x; d0 <= R0; d0 += gridDim.x * blockDim.x) {
    uint32_t d = d0;
    for (uint32_t b = 0; b < NB; b++) d = W.btab[(uint64_t)W.blk_toff[b] + d];
    fn[d0] = d;
  }
}

// Short lists: five launches cost a 2 000-gap list more than the work in them — classes and scan in one launch
// of one workgroup (the per-gap words stay in LDS between the two), blocks and chain in another; the tables keep
// their own (they want the whole chip).
#define D3_FRONT_GAPS 3072u
__global__ __launch_bounds__(1024) void g2s_d3_front(const D3Params P, const D3Work W, const GapOut* __restrict__ outs,
                                                     const D3Gap* __restrict__ dgaps) {
  __shared__ uint64_t sh[96];
  __shared__ unsigned long long red[16 * 8];
  __shared__ uint32_t sh_f[1024];
  __shared__ uint32_t l_gi[D3_FRONT_GAPS], l_dm[D3_FRONT_GAPS], l_ds[D3_FRONT_GAPS];
  __shared__ int32_t l_sk[D3_FRONT_GAPS];
  if (threadIdx.x == 0) stamp(W, 0);
  const uint32_t unhandled = d3_classify_body(P, W, outs, dgaps, threadIdx.x, 1024u, l_gi, l_dm, l_ds, l_sk, red, true);
  __syncthreads();
  if (threadIdx.x == 0) stamp(W, 1);
  d3_scan_body(P, W, outs, l_gi, l_dm, l_ds, l_sk, true, unhandled, sh, sh_f, true);
  if (threadIdx.x == 0) stamp(W, 2);
}

// ---------------------------------------------------------------------------------------------------------
// hand-off: every gap's first draw and draw count, with what else the trace kernel needs of it, in one record
// (D3Trace); the gaps whose closure the host analyses (GI_HOST) get their record, closure segments and the rand()
// values of their traceback copied into pinned memory — in front of the trace kernel, so that the host finishes
// them while that kernel runs.  One wave per 64 gaps of the list.
// ---------------------------------------------------------------------------------------------------------
// (the body: one wave, the 64 gaps from i0 on; true when it handed something over.  ldvar/ltab/ltoff: the
// deviations, tables and table offsets in LDS when the caller holds them there, else null; lcur: the three cursors
// of the side buffers in LDS when one workgroup hands the whole list over, else null — the summary's are used)
__device__ __forceinline__ bool d3_handoff_body(const D3Params& P, const D3Work& W, const GapOut* __restrict__ outs,
                                                const SubRec* __restrict__ sub, const uint32_t* __restrict__ rnd, uint64_t capacity,
                                                const g2s::D3Side& side, uint32_t i0, const uint32_t* ldvar, const uint16_t* ltab,
                                                const uint32_t* ltoff, unsigned long long* lcur) {
  D3Summary* S = W.sum;
  const int lane = (int)(threadIdx.x & 63u);
  const uint32_t mine = i0 + (uint32_t)lane;
  const bool have = mine < P.n;
  uint32_t gi = 0, my_off = 0, my_want = 0, my_ns = 0;
  if (have) {
    gi = W.ginfo[mine];
    const uint32_t vr = W.vrank[mine], base = W.base[mine], dmin = W.dmin[mine];
    const GapOut& go = outs[mine];
    my_ns = go.n_xl;
    // (g2s_d2_* runs beside these kernels: a gap it was to analyse and has given up on — beyond its capacities — is the
    // host's after all; one it has not got to yet is waited for by the gap's trace wave)
    if ((gi & GI_PHASE_D) && !P.skip_confident) {
      const uint32_t dfl = go.dflags;
      if ((dfl & G2S_DEVA_D2_PENDING) && (dfl & G2S_DEVA_D2_FAILED)) gi |= GI_HOST;
    }
    const uint32_t dv = ldvar ? ldvar[vr] : W.dvar[vr];
    my_off = base + dv;
    my_want = dmin;
    if (GI_CLASS(gi) == 2u) my_want += ltab ? (uint32_t)ltab[ltoff[vr] + dv] : (uint32_t)W.tab[(uint64_t)W.var_toff[vr] + dv];
    // the half of the trace kernel's record that depends on the offsets (the classes wrote the other)
    ((uint4*)&W.tdesc[mine])[1] = make_uint4(gi, my_off, my_want, (gi >> 16) | (min(my_ns, 0xFFFFu) << 16));
  }
  const bool host = have && (gi & GI_HOST) != 0u;
  const uint64_t hm = __ballot(host);
  for (uint64_t m = hm; m; m &= m - 1) {
    const int l = __builtin_ctzll(m);
    const uint32_t i = i0 + (uint32_t)l;
    const uint32_t off = (uint32_t)__shfl((int)my_off, l), want = (uint32_t)__shfl((int)my_want, l), ns = (uint32_t)__shfl((int)my_ns, l);
    unsigned long long it = 0, so = 0, ro = 0;
    if (lane == 0) {
      if (lcur) {
        it = atomicAdd(&lcur[0], 1ull); so = atomicAdd(&lcur[1], (unsigned long long)ns); ro = atomicAdd(&lcur[2], ((unsigned long long)want + 8ull) / 8ull);
      } else {
        it = atomicAdd(&S->host_items, 1ull);
        so = atomicAdd(&S->host_segs, (unsigned long long)ns);
        ro = atomicAdd(&S->host_rnd, ((unsigned long long)want + 8ull) / 8ull);  // (its draws go to the host four bits each: below)
      }
    }
    it = __shfl(it, 0); so = __shfl(so, 0); ro = __shfl(ro, 0);
    if (it >= side.cap_items || so + ns > side.cap_segs || ro + (want + 8ull) / 8ull > side.cap_rnd || (uint64_t)off + want + 1ull > capacity) {
      if (lane == 0) { atomicAdd(&S->anomalies, 1u); W.host_slot[i] = 0xFFFFFFFFu; }
      continue;
    }
    // (where the gap's wave of the trace kernel will put its record, closure and rand() values: it does the
    // copying — beside the other gaps' tracebacks instead of in front of them)
    if (lane == 0) {
      g2s::D3HostItem h;
      h.gap = i; h.n_segs = ns; h.seg_off = so; h.rnd_off = ro; h.draws = want; h.pad = 0;
      W.hitems[it] = h;
      W.host_slot[i] = (uint32_t)it;
      // (the host as well, without the ready word: it takes the items largest closure first)
      uint32_t* hi = (uint32_t*)&side.items[it];
      hi[0] = i; hi[1] = ns; hi[2] = (uint32_t)so; hi[3] = (uint32_t)(so >> 32); hi[4] = (uint32_t)ro; hi[5] = (uint32_t)(ro >> 32); hi[6] = want;
    }
  }
  if (hm != 0ull) __threadfence_system();  // (the items are in host memory before the count that announces them)
  (void)sub; (void)rnd; (void)outs;
  return hm != 0ull;
}
// (*side.count is ~0 until the hand-over is complete: the host polls it.  Bit 63: something did not fit, or the
// list is not finished on the device at all.)
__global__ __launch_bounds__(64) void g2s_d3_handoff(const D3Params P, const D3Work W, const GapOut* __restrict__ outs,
                                                     const SubRec* __restrict__ sub, const uint32_t* __restrict__ rnd, uint64_t capacity,
                                                     const g2s::D3Side side) {
  D3Summary* S = W.sum;
  const bool dead = S->status != 0u;
  if (!dead) (void)d3_handoff_body(P, W, outs, sub, rnd, capacity, side, blockIdx.x * 64u, nullptr, nullptr, nullptr, nullptr);
  if ((threadIdx.x & 63u) == 0u) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wave's additions to the cursors have been acknowledged)
    const unsigned int done = atomicAdd(&S->handoff_waves, 1u) + 1u;
    if (done == gridDim.x) {
      const unsigned long long items = atomicAdd(&S->host_items, 0ull);
      const uint32_t anomalies = atomicAdd(&S->anomalies, 0u);
      __hip_atomic_store(side.count, (items & 0x7FFFFFFFFFFFFFFFull) | ((unsigned long long)((anomalies || dead) ? 1u : 0u) << 63),
                         __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// Short lists: blocks, chain and the hand-off in one launch of one workgroup.  When the tables are small (a 500-gap
// list: a thousand entries) they come into LDS in one pass and one lane walks the chain there — a chain of loads
// from LDS instead of two passes of 16 dependent loads from memory.
#define D3_BACK_TAB 24576u
__global__ __launch_bounds__(1024) void g2s_d3_back(const D3Params P, const D3Work W, const GapOut* __restrict__ outs,
                                                    const SubRec* __restrict__ sub, const uint32_t* __restrict__ Wd, uint64_t capacity,
                                                    const g2s::D3Side side) {
  __shared__ uint32_t total_dev;
  __shared__ unsigned long long lcur[3];
  if (threadIdx.x < 3u) lcur[threadIdx.x] = 0ull;
  __shared__ uint16_t ltab[D3_BACK_TAB];
  __shared__ uint32_t ltoff[1025], ldvar[1025];
  D3Summary* S = W.sum;
  if (threadIdx.x == 0) stamp(W, 8);
  const uint32_t status = S->status, V = S->n_var;
  const uint64_t T = S->table_entries;
  if (status) {
    if (threadIdx.x == 0) __hip_atomic_store(side.count, 1ull << 63, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  const bool in_lds = T <= (uint64_t)D3_BACK_TAB && V <= 1024u;
  if (in_lds) {
    for (uint32_t e = threadIdx.x; e < (uint32_t)T; e += 1024u) ltab[e] = W.tab[e];
    if (threadIdx.x <= V) ltoff[threadIdx.x] = W.var_toff[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
      stamp(W, 9);
      uint32_t d = 0;
      for (uint32_t v = 0; v < V; v++) { ldvar[v] = d; d += ltab[ltoff[v] + d]; }
      ldvar[V] = d;
      total_dev = d;
    }
    __syncthreads();
    if (threadIdx.x <= V) W.dvar[threadIdx.x] = ldvar[threadIdx.x];
    const uint64_t total = S->draws_min + total_dev;
    if (threadIdx.x == 0) S->draws_total = total;
    if (threadIdx.x < 31u) {
      const uint32_t w = Wd[total + threadIdx.x];
      S->rand_state[threadIdx.x] = w;
      if (W.link) W.link[threadIdx.x] = w;
    }
  } else {
    d3_blocks_body(W, threadIdx.x, 1024u);
    __threadfence();
    __syncthreads();
    d3_chain_body(W, Wd, &total_dev);
    __threadfence();
    __syncthreads();
  }
  if (threadIdx.x == 0) stamp(W, 10);
  for (uint32_t i0 = (threadIdx.x >> 6) * 64u; i0 < P.n; i0 += 1024u)
    (void)d3_handoff_body(P, W, outs, sub, Wd + 31, capacity, side, i0, in_lds ? ldvar : nullptr, in_lds ? ltab : nullptr,
                          in_lds ? ltoff : nullptr, lcur);
  __syncthreads();
  if (threadIdx.x == 0) {
    stamp(W, 11);
    const unsigned long long items = lcur[0];
    S->host_items = items; S->host_segs = lcur[1]; S->host_rnd = lcur[2];
    const uint32_t anomalies = atomicAdd(&S->anomalies, 0u);
    __hip_atomic_store(side.count, (items & 0x7FFFFFFFFFFFFFFFull) | ((unsigned long long)(anomalies ? 1u : 0u) << 63), __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    stamp(W, 12);
  }
}

// ---------------------------------------------------------------------------------------------------------
// the tracebacks: one wave per gap.  Writes g2s_result[i] in full and the gap's fill text.
// ---------------------------------------------------------------------------------------------------------
// (its first loads — the gap's D3Trace record, its GapOut record, the list's status — do not depend on each other;
// the closure and the first rand() value follow from the D3Trace record: two round trips in front of the walk)
// NW waves per gap: one on a list that fills the chip; four on a short list, whose trace kernel is its slowest gap —
// the passes over the closure, over the stretches of the fill and over its bases are shared by the workgroup's threads,
// the chain of segments is walked by every wave for itself (LDS reads; the same words written by all)
template <int NW>
__global__ __launch_bounds__(64 * NW) void g2s_d3_trace(const D3Params P, const D3Work W, GapOut* __restrict__ outs,
                                                   const SubRec* __restrict__ sub, const char* __restrict__ chu,
                                                   const char* __restrict__ chd, const uint32_t* __restrict__ rnd,
                                                   uint64_t capacity, g2s_result* __restrict__ results,
                                                   char* __restrict__ arena, uint32_t* __restrict__ summary_host,
                                                   const g2s::D3Side side, uint32_t* __restrict__ clean_words /* or null */) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  D3Summary* S = W.sum;
  const uint32_t i = blockIdx.x;
  constexpr int NT = 64 * NW;
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const bool w0 = tid < 64;  // (the first wave: what one wave does for the gap — counters, the hand to the host, the clean-up)
#ifdef G2S_SYNC_DEBUG
  auto wg_sync = [&]() { if constexpr (NW > 1) __syncthreads(); else { g2s_sync_jitter(0xD3u); g2s_wait_all(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } };
#else
  auto wg_sync = [&]() { if constexpr (NW > 1) __syncthreads(); else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } };
#endif
  SegW* segs = (SegW*)lds;  // the gap's closure segments
  static_assert(sizeof(g2s_result) == 112, "g2s_result layout");
  unsigned long long* laps = (unsigned long long*)((char*)W.sum + 512);
  const unsigned long long tk0 = P.laps ? wall_clock64() : 0ull;
  unsigned long long tk1 = 0, tk2 = 0, tk3 = 0, hops = 0, tkm = 0;
  if (P.laps && i == 0 && tid == 0) laps[13] = tk0;
  const uint32_t status = uni(S->status);
  // (every lane loads the same words: handed to the scalar unit, so that the walk below is scalar code)
  g2s::D3Trace td;
  {
    const uint4 h0 = ((const uint4*)&W.tdesc[i])[0], h1 = ((const uint4*)&W.tdesc[i])[1];
    td.arena_off = (uint64_t)uni(h0.x) | ((uint64_t)uni(h0.y) << 32);
    td.sub_at = (uint64_t)uni(h0.z) | ((uint64_t)uni(h0.w) << 32);
    td.gi = uni(h1.x); td.off = uni(h1.y); td.want = uni(h1.z);
    td.lmf = (uint16_t)(uni(h1.w) & 0xFFFFu); td.nsegs = (uint16_t)(uni(h1.w) >> 16);
  }
  struct {
    uint32_t flags, dflags, start_seg, start_t, sub_vertices, sub_edges, top_level;
    int32_t c_count, n_len, len[2], reached_j, count_s;
  } go;
  {
    const GapOut& g = outs[i];
    go.top_level = uni(g.top_level);
    go.flags = uni(g.flags); go.dflags = uni(g.dflags); go.start_seg = uni(g.start_seg); go.start_t = uni(g.start_t);
    go.sub_vertices = uni(g.sub_vertices); go.sub_edges = uni(g.sub_edges);
    go.c_count = (int32_t)uni((uint32_t)g.c_count); go.n_len = (int32_t)uni((uint32_t)g.n_len);
    go.len[0] = (int32_t)uni((uint32_t)g.len[0]); go.len[1] = (int32_t)uni((uint32_t)g.len[1]);
    go.reached_j = (int32_t)uni((uint32_t)g.reached_j); go.count_s = (int32_t)uni((uint32_t)g.count_s);
  }
  // (the gap's closure is g2s_d2_*'s to analyse, on its own stream: until its verdict word says it has — or has given
  // up —, this wave waits; then the words that kernel wrote are read again)
  bool d2_lost = false;
  if ((go.dflags & G2S_DEVA_D2_PENDING) && !(go.dflags & (G2S_DEVA_RUNS | G2S_DEVA_D2_FAILED)) && !P.skip_confident) {
    uint32_t v = go.dflags;
    // (an acquire with every look: a relaxed look may be answered from this compute unit's view of the L2 for as long
    // as the line stays there — two of ten config-3 steps took 8 ms that way)
    for (uint32_t spin = 0; spin < (1u << 22) && !(v & (G2S_DEVA_RUNS | G2S_DEVA_D2_FAILED)); spin++) {
      __builtin_amdgcn_s_sleep(8);
      v = uni(__hip_atomic_load(&outs[i].dflags, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    go.dflags = v;
    go.sub_vertices = uni(__hip_atomic_load(&outs[i].sub_vertices, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    go.sub_edges = uni(__hip_atomic_load(&outs[i].sub_edges, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (!(v & (G2S_DEVA_RUNS | G2S_DEVA_D2_FAILED))) d2_lost = true;  // (never expected: the list goes to the host path)
  }
  // (given up after the hand-off looked: too late for a slot in the host's side buffers — counted, the host path's)
  if ((go.dflags & G2S_DEVA_D2_PENDING) && (go.dflags & G2S_DEVA_D2_FAILED) && !(td.gi & GI_HOST) && (td.gi & GI_PHASE_D)) d2_lost = true;
  // The wave that is through last copies the summary and the fill-byte counters to the host's pinned memory (a copy
  // command behind the kernel costs the stream a barrier).  Who is last: the fill-byte counter of the wave's residue
  // modulo 64 also counts its waves (bits 40 and up — one addition for both; one counter for all waves would have
  // 10 000 of them queue at one address), the wave that completes a residue adds to the summary's counter, the one
  // that completes that copies.  Counters other waves add to are read at the L2.
  uint32_t gsp_all = 0, gsp_sent = 0;  // (guessed gaps: groups of 64 bases the first wave compared / sent again — the other waves' are not counted)
  auto leave = [&](uint32_t fill_len) {
    if (!w0) return;
    uint32_t last = 0;
    if (lane == 0) {
      if (P.laps) {  // (the longest wave's laps: entry to closure in LDS, the walk, the bases, all of it; the latest end)
        const unsigned long long tk4 = wall_clock64();
        atomicMax(&laps[14], tk1 ? tk1 - tk0 : 0ull); atomicMax(&laps[15], tk2 ? tk2 - tk1 : 0ull);
        atomicMax(&laps[16], tk3 ? tk3 - tk2 : 0ull); atomicMax(&laps[17], ((tk4 - tk0) << 24) | ((td.gi & GI_HOST) ? 1ull << 23 : 0ull) | (unsigned long long)(i & 0x7FFFFFu)); atomicMax(&laps[18], tk4);
        atomicMax(&laps[19], ((tk2 ? tk2 - tk1 : 0ull) << 32) | (hops << 16) | (tkm ? tkm - tk1 : 0ull));
        if ((P.laps >> 4) == i + 1u) { laps[26] = tk0; laps[27] = tk1; laps[28] = tkm; laps[29] = tk2; laps[30] = tk3; laps[31] = tk4; }  // (G2S_DEBUG_GAP: this gap's own)
      }
      const uint32_t c = i & 63u, n = gridDim.x;
      const unsigned long long before = atomicAdd(&W.fill_bytes[c * 16u], (unsigned long long)fill_len | (1ull << 40));
      if (gsp_all) atomicAdd(&W.fill_bytes[c * 16u + 1u], (unsigned long long)gsp_all | ((unsigned long long)gsp_sent << 32));  // (the same line)
      if ((uint32_t)(before >> 40) + 1u == (n + 63u - c) / 64u) last = atomicAdd(&S->trace_waves, 1u) + 1u == min(n, 64u) ? 1u : 0u;
    }
    // (self_clean — a list that is one batch on one session: what the next list's kernels expect to find zero is
    // zeroed here instead of by three memsets behind the kernel, which cost the host 10 us between two lists: this
    // gap's record, and by the last wave the summary, the counters and the fill kernel's cursors)
    if (P.self_clean && (uint32_t)lane < sizeof(GapOut) / 4u) ((uint32_t*)&outs[i])[lane] = 0u;
    if (!uni(last)) return;
    // (g2s_d2_*'s workgroups have all left before the cursors they read are zeroed)
    if (P.d2_done && lane == 0)
      for (uint32_t spin = 0; spin < (1u << 22) && __hip_atomic_load(P.d2_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)P.d2_wgs; spin++)
        __builtin_amdgcn_s_sleep(8);
    uint32_t* src = (uint32_t*)S;
    if (P.laps && lane == 0) laps[24] = wall_clock64();
    // (what the host reads: the summary's 512 bytes, and of each of the 64 counters' lines its first two words — all
    // loads first, then the stores: as a loop over the slot's 2 304 words, a load and a store an iteration, every store
    // waited for the acknowledgement of the one before it over the link — 8-11 us at the end of every list's last kernel)
    static_assert(sizeof(D3Summary) <= 512, "two words a lane");
    const uint32_t s0 = __hip_atomic_load(src + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t s1 = __hip_atomic_load(src + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long* cl = (const unsigned long long*)(src + 256 + 32 * lane);  // (counter `lane`: a 128-byte line)
    const unsigned long long c0 = __hip_atomic_load(cl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long c1 = __hip_atomic_load(cl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    summary_host[lane] = s0;
    summary_host[64 + lane] = s1;
    ((unsigned long long*)(summary_host + 256 + 32 * lane))[0] = c0;
    ((unsigned long long*)(summary_host + 256 + 32 * lane))[1] = c1;
    if (P.self_clean) {  // (nothing else of the slot is ever written: the lap stamps at byte 512 only without self_clean)
      src[lane] = 0u; src[64 + lane] = 0u;
      ((unsigned long long*)(src + 256 + 32 * lane))[0] = 0ull;
      ((unsigned long long*)(src + 256 + 32 * lane))[1] = 0ull;
    }
    if (P.self_clean && clean_words && lane < 32) clean_words[lane] = 0u;  // (the fill kernels' and g2s_d2_*'s cursors: 16 counters)
    if (P.laps && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); laps[25] = wall_clock64(); }
  };
  if (status) { leave(0u); return; }  // (the records in front of this kernel were not written: nothing below may run)
  if (d2_lost) {
    if (tid == 0) { atomicAdd(&S->anomalies, 1u); arena[td.arena_off + td.lmf] = '\0'; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    leave(0u);
    return;
  }
  const uint32_t gi = td.gi;
  // (its fill kernel's wave traced the gap itself — one path, nothing to draw for: text and record are written; a gap the
  // skip rule or the memory verdict took out after all cannot carry the flag: fill_seg.hip asks for neither)
  if ((go.dflags & G2S_DEVA_TRACED) && !(gi & (GI_BAD | GI_SKIPPED | GI_MEM | GI_HOST)) && (gi & GI_PHASE_D)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    leave((go.top_level >> 16) - (go.top_level & 0xFFFFu));
    return;
  }
  // (its fill kernel's wave wrote a guess — first length, first parent at every choice — to the caller's buffers and to
  // device memory: what this wave finds equal there, 64 bases or the whole record at a time, it does not send again)
  const bool spec = (go.dflags & G2S_DEVA_SPEC) != 0u && W.spec_text != nullptr && !(gi & (GI_BAD | GI_SKIPPED | GI_MEM | GI_HOST)) && (gi & GI_PHASE_D);
  const int spec_stop = (int)(go.top_level & 0xFFFFu), spec_len = (int)(go.top_level >> 16);
  // (the chain of segments the guess followed, as a 64-bit hash of their ids in order: fill_seg.hip)
  const unsigned long long spec_chain = spec ? ((unsigned long long)uni(outs[i].stat[6]) | ((unsigned long long)uni(outs[i].stat[7]) << 32)) : 0ull;
  struct { uint16_t lmf; } dg = {td.lmf};
  const uint64_t abs_off = td.arena_off;
  char* buf = arena + abs_off;
  const int k = P.k;
  // the record, words 0-23 of g2s_result (count, left_fuz, right_fuz, flags, fill_off, fill_len, draws, six 64-bit
  // subgraph statistics, phaseC_count, n_lengths, lengths[2]); words 24-27 (the "Unable to backtrace!" numbers) are zero
  uint32_t rw[24];
#pragma unroll
  for (int q = 0; q < 24; q++) rw[q] = 0u;
  int left_fuz = 0;
  bool spec_rec = false;  // (set where the gap turns out to be one whose guess stands to be compared)
  uint32_t g_all = 0, g_sent = 0;  // (a guessed gap: groups of 64 bases this wave looked at / sent through the link again)
  // (the first wave writes it, a word a lane: 112 contiguous bytes in one instruction — seven 16-byte stores of one lane
  // were seven packets of the link; the words are the same in every lane)
  auto finish = [&](uint32_t fill_len) {
    if (!w0) return;
    const uint64_t fo = abs_off + (uint64_t)((int)dg.lmf - left_fuz);
    rw[1] = (uint32_t)left_fuz;
    rw[4] = (uint32_t)fo; rw[5] = (uint32_t)(fo >> 32);
    rw[6] = fill_len;
    uint32_t mine = 0u;  // (words 24-27 — backtrace_depth, backtrace_final_d: a traceback that fails is the host path's)
#pragma unroll
    for (int q = 0; q < 24; q++) mine = lane == q ? rw[q] : mine;
    if (spec_rec) {  // (the guessed record is there already: the same words are not sent again)
      const uint32_t was = lane < 28 ? W.spec_res[(size_t)i * 28u + (uint32_t)lane] : 0u;
      if (__ballot(lane < 28 && was != mine) == 0ull) return;
    }
    if (lane < 28) ((uint32_t*)&results[i])[lane] = mine;
  };
  if (gi & (GI_BAD | GI_SKIPPED | GI_MEM)) {
    rw[3] = (gi & GI_BAD) ? G2S_GAP_BAD_FLANK : (gi & GI_SKIPPED) ? G2S_GAP_SKIPPED : G2S_GAP_MEM_EXCEEDED;
    if (gi & GI_MEM) rw[0] = (uint32_t)-1;
    if (tid == 0) buf[dg.lmf] = '\0';
    finish(0u);
    leave(0u);
    return;
  }
  const bool phase_d = (gi & GI_PHASE_D) != 0;
  rw[20] = (uint32_t)go.c_count;
  rw[21] = (uint32_t)go.n_len;
  rw[22] = (uint32_t)go.len[0];
  rw[23] = (uint32_t)go.len[1];
  rw[3] = (go.flags & (G2S_DEV_Q7_A | G2S_DEV_Q7_B | G2S_DEV_Q7_D)) ? G2S_GAP_Q7 : 0u;
  rw[0] = (uint32_t)((phase_d && !P.skip_confident && P.all_paths) ? go.count_s : go.c_count);  // pp.count of the host analysis
  // (the gap's closure was analysed by g2s_d2_* — d2_device.hip: more than 192 segments, or a k-mer at several depths:
  // the statistics and the verdicts are in its record and its runs)
  const bool by_runs = phase_d && !P.skip_confident && (go.dflags & G2S_DEVA_RUNS) != 0u && W.d2out != nullptr;
  uint32_t n_runs = 0;
  const uint32_t* runs = nullptr;
  if (phase_d && !P.skip_confident) {
    rw[8] = go.sub_vertices; rw[10] = go.sub_edges;    // vertices, edges (nothing contracted: no non-trivial component)
    rw[16] = go.sub_vertices; rw[18] = go.sub_edges;   // vertices_final, edges_final
    if (by_runs) {
      const uint4 o0 = ((const uint4*)&W.d2out[i])[0], o1 = ((const uint4*)&W.d2out[i])[1];
      runs = W.d2runs + 2ull * uni(o0.x);
      n_runs = uni(o0.y);
      rw[8] = uni(o0.z); rw[10] = uni(o0.w); rw[12] = uni(o1.x); rw[14] = uni(o1.y); rw[16] = uni(o1.z); rw[18] = uni(o1.w);
    }
  }
  if (!phase_d) {
    if (tid == 0) buf[dg.lmf] = '\0';
    finish(0u);
    leave(0u);
    return;
  }
  if (gi & GI_HOST) {
    // The host analyses this gap's closure (a k-mer at two depths) and traces it, while the other gaps' waves work:
    // its record, closure segments and the rand() values of its traceback go to pinned memory, then the item's
    // ready word (the host polls it; it knows from *side.count how many items to expect).
    const uint32_t it = uni(W.host_slot[i]);
    if (it != 0xFFFFFFFFu) {
      const g2s::D3HostItem* hp = &W.hitems[it];
      const uint64_t so = (uint64_t)uni((uint32_t)hp->seg_off) | ((uint64_t)uni((uint32_t)(hp->seg_off >> 32)) << 32);
      const uint64_t ro = (uint64_t)uni((uint32_t)hp->rnd_off) | ((uint64_t)uni((uint32_t)(hp->rnd_off >> 32)) << 32);
      const uint32_t ns = uni(hp->n_segs), want = uni(hp->draws);
      const GapOut& g = outs[i];
      if ((uint32_t)tid < sizeof(GapOut) / 4u) ((uint32_t*)&side.outs[it])[tid] = ((const uint32_t*)&g)[tid];
      const uint4* src = (const uint4*)(sub + td.sub_at);
      uint4* dst = (uint4*)(side.segs + so);
      for (uint32_t w = (uint32_t)tid; w < 2u * ns; w += (uint32_t)NT) dst[w] = src[w];
      // (the rand() values of its traceback as their remainders by 12, four bits each: all a traceback asks of a value is
      // its remainder by the number of lengths or of parents — 1 .. 4; config 5's 412 host-finished gaps: 7 MB of raw
      // words through the link, half of what the kernel wrote)
      for (uint32_t w = (uint32_t)tid; w < (want + 8u) / 8u; w += (uint32_t)NT) {
        uint32_t pk8 = 0u;
#pragma unroll
        for (uint32_t x = 0; x < 8u; x++) {
          const uint32_t at = 8u * w + x;
          if (at <= want) pk8 |= ((rnd[td.off + at] >> 1) % 12u) << (4u * x);
        }
        side.rnd[ro + w] = pk8;
      }
      __threadfence_system();  // (the whole workgroup's stores are in host memory before the item says so)
      if constexpr (NW > 1) __syncthreads();
      if (tid == 0) {
        // (the item's other words: the hand-off wrote them)
        __hip_atomic_store(&side.items[it].pad, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    leave(0u);
    return;
  }
  // ---- the closure into LDS
  // (closures the fill kernel analysed have at most seg_cap segments; a longer one — analysed by g2s_d2_* — is walked
  // where it lies, the segments its traceback enters listed in device memory: `big`)
  const bool big = (uint32_t)td.nsegs > P.seg_cap && by_runs && W.hops != nullptr;
  const uint32_t nsegs = big ? (uint32_t)td.nsegs : min((uint32_t)td.nsegs, P.seg_cap);
  const SegW* gsegs = (const SegW*)(sub + td.sub_at);
  uint64_t* ghop = big ? W.hops + td.sub_at / 2u : nullptr;
  uint32_t* cmap = lds + (size_t)P.seg_cap * 8u;  // by fill-buffer index: k-mer index | orientation << 30 | lower case << 31
  // the rand() values of this traceback, into LDS together with the closure: the walk asks for one at every choice
  // between parents, and each was a round trip to memory in the middle of it (30 us for the longest of 500 walks)
  uint32_t* lwin = cmap + P.map_cap;
  const uint32_t nwin = min(td.want, P.map_cap);
  for (uint32_t x = (uint32_t)tid; x < nwin; x += (uint32_t)NT) lwin[x] = (uint64_t)td.off + x < capacity ? rnd[td.off + x] : 0u;
  if (!big) {
    const uint4* src = (const uint4*)gsegs;
    uint4* dst = (uint4*)segs;
    for (uint32_t w = (uint32_t)tid; w < 2u * nsegs; w += (uint32_t)NT) dst[w] = src[w];
  }
  __syncthreads();
  if (P.laps) tk1 = wall_clock64();
  // the branch rule's verdict for a k-mer that is not in the subgraph at this depth: at another depth, or the
  // sink's (Q5) — post.cpp: seg_safe.  (Asked by single lanes, for the few states of a traceback closure that are
  // on no path to a sink: the first segment in emission order that holds the k-mer decides.)
  const bool sink_safe = (go.dflags & G2S_DEVA_SINK_SAFE) != 0;
  auto outside_safe = [&](uint32_t x) -> bool {
    for (uint32_t q = 0; q < nsegs; q++) {
      const uint32_t onode = segs[q].node, ots = segs[q].ts_tt;
      const int ts = (ots & 0x7FFFu) == 0x7FFFu ? -1 : (int)(ots & 0x7FFFu);
      const uint32_t oidx = onode >> 1;
      const int tq = (onode & 1u) ? (int)oidx - (int)x : (int)x - (int)oidx;
      if (tq >= 0 && tq <= ts) return tq <= (int)segs[q].pad ? (ots & 0x8000u) != 0 : (ots & 0x80000000u) != 0;
    }
    return sink_safe;
  };
  // the verdict of k-mer x from the gap's runs (sorted, disjoint); *rlo, *rhi: the run's ends, or x itself when it is
  // in no run (a k-mer outside the subgraph reads branch[sink], Q5)
  const uint32_t* lruns = nullptr;  // (the gap's runs in LDS, once the walk has no more use for the rand() values there)
  auto run_verdict = [&](uint32_t x, uint32_t* rlo, uint32_t* rhi) -> bool {
    const uint32_t* rr = lruns ? lruns : runs;
    uint32_t lo = 0, hi = n_runs;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rr[2u * mid] <= x) lo = mid + 1u; else hi = mid; }
    if (lo > 0u) {
      const uint32_t f = rr[2u * (lo - 1u)], l = rr[2u * (lo - 1u) + 1u];
      if (x <= (l & 0x7FFFFFFFu)) { *rlo = f; *rhi = l & 0x7FFFFFFFu; return (l >> 31) != 0u; }
    }
    *rlo = x; *rhi = x;
    return sink_safe;
  };
  // ---- the walk, in two parts.  (i) Wave-uniform and as short as it can be: which segments the traceback enters,
  // and at which state — per segment one read of its links from LDS and, where it has several parents, of a rand()
  // value.  (ii) All lanes: where every base of the fill comes from — k-mer index, orientation, safe bit — into
  // cmap, a lane per stretch of consecutive depths; then upper or lower case (:1466-1468: lower case = not safe
  // and k or more below the nearest safe base above) as a scan over the safe bits.  (One pass that did both per
  // segment took the longest of 500 walks 25 us — 38 segments at 1 500 cycles of dependent scalar work each;
  // the bases themselves are fetched afterwards, all lanes at once and several loads in flight.)
  const uint32_t off = td.off;
  const uint64_t avail = min((uint64_t)nwin, capacity > off ? capacity - off : 0ull);  // (values the walk can read: in LDS)
  const int want = (int)td.want;
  const int n_len = go.n_len, len0 = go.len[0], len1 = go.len[1];
  int draws = 1;
  auto value = [&](uint32_t x) -> uint32_t { return uni(lwin[x]); };
  const int pick = (n_len > 1 && avail > 0) ? (int)((value(0) >> 1) & 1u) : 0;  // :1440 (n_len <= 2)
  int d2 = pick ? len1 : len0;
  const int len = d2;
  bool bad = false, ended = false;
  if ((uint32_t)len + 1u > P.map_cap || (!big && (uint32_t)td.nsegs > P.seg_cap) || td.want > P.map_cap) bad = true;
  uint2* hop = (uint2*)(lwin + P.map_cap);  // by hop: the depth at which the walk enters the segment (descending), the segment | its entry state << 16
  uint2* pk = hop + P.seg_cap;              // by segment: depth | length << 16, the parent a traceback goes on to from its first state | source << 30 | no way on << 31
  // (Every traced base draws one value — :1513 draws for a single parent too — so the draw made at depth d is the
  // (1 + len - d)-th of the gap whatever the path: the parent a traceback takes from a segment's first state is a
  // property of the segment.  All lanes work those out; the walk itself then reads three words per segment.)
  for (uint32_t q = (uint32_t)tid; q < nsegs && !big; q += (uint32_t)NT) {
    const uint32_t dl = segs[q].depth_len, p01 = segs[q].par01, p23 = segs[q].par23, fl = segs[q].flags;
    const int d0 = (int)(dl & 0xFFFFu);
    const int nb = seg_nparents(p01, p23);
    const int64_t at = 1 + (int64_t)len - d0;  // draws in front of the one made at this segment's first state
    uint32_t w;
    if (nb == 0 || (nb > 1 && !(fl & G2S_SEG_ORDERED)) || at < 1 || (uint64_t)at >= avail) w = 0x80000000u;  // (:1493-1510: the host path's business)
    else {
      const uint32_t rv = nb > 1 ? lwin[at] >> 1 : 0u;  // (rand() % 1: the value does not matter)
      w = seg_parent(p01, p23, nb == 1 ? 0 : pick_parent(rv, nb));  // :1513
      if (w >= nsegs) w = 0x80000000u;
    }
    pk[q] = make_uint2(dl, w | ((fl & G2S_SUB_SOURCE) ? 0x40000000u : 0u));
  }
  wg_sync();
  int nh = 0;
  {
    // (i) The chain of segments: from the start along the parents chosen above until a source or a segment without a
    // way on — ONE LDS word and a dozen scalar instructions per segment entered (the walk that also carried depths and
    // checked them was 80 instructions and 500 cycles a segment: 8 of the 27 us of config 2's longest wave).
    // A traceback descends: it enters a segment once, so a chain of more than nsegs segments is not one.
    const uint32_t sg = (go.start_seg >> (16 * pick)) & 0xFFFFu;
    int si = sg == 0xFFFFu ? -1 : (int)sg;
    const int t0 = (int)((go.start_t >> (16 * pick)) & 0xFFFFu);
    if (si < 0 || si >= (int)nsegs || avail < 1) bad = true;
    uint32_t last = 0x80000000u;
    const int hop_cap = (int)(big ? nsegs : min(nsegs, P.seg_cap));
    unsigned long long chain_hash = 14695981039346656037ull;
    if (!bad && !big) {
      if (spec) {  // (a guessed gap: the chain's hash in the order of the walk)
        for (;;) {
          if (nh >= hop_cap) { bad = true; break; }
          last = uni(pk[si].y);
          chain_hash = (chain_hash ^ (unsigned long long)(uint32_t)si) * 1099511628211ull;
          if (lane == 0) hop[nh].y = (uint32_t)si;
          nh++;
          if (last & 0xC0000000u) break;
          si = (int)(last & 0xFFFFu);
        }
      } else {
        // (the same chain with nothing in the step but the read it depends on: the segment entered at hop h stays in
        // lane h & 63 and goes to LDS 64 hops at a time — with the hash and a masked store a step was 350 cycles, and
        // the longest of a short list's walks, 38 segments, 6 of its trace wave's 20 us)
        uint32_t hv = 0u;
        for (;;) {
          if (nh >= hop_cap) { bad = true; break; }
          last = uni(pk[si].y);
          hv = lane == (nh & 63) ? (uint32_t)si : hv;
          nh++;
          if ((nh & 63) == 0) hop[nh - 64 + lane].y = hv;
          if (last & 0xC0000000u) break;
          si = (int)(last & 0xFFFFu);
        }
        if (!bad && (nh & 63) != 0 && lane < (nh & 63)) hop[(nh & ~63) + lane].y = hv;
      }
      if (P.laps) hops = (unsigned long long)nh;
      // (the fill kernel's wave guessed THIS chain, from this length: text, case, fuz values and draws are the guess's —
      // they are in the caller's buffers already)
      if (!bad && spec && pick == 0 && len == spec_len && (last & 0x40000000u) && chain_hash == spec_chain) {
        gsp_all = 1u;  // (counted as one group compared, none sent)
        leave((uint32_t)(spec_len - spec_stop));
        return;
      }
    }
    if (!bad && big) {  // (the same chain over the records in device memory: a round trip per segment entered)
      for (;;) {
        if (nh >= hop_cap) { bad = true; break; }
        if (P.laps) hops++;
        const uint32_t dl = uni(gsegs[si].depth_len), p01 = uni(gsegs[si].par01), p23 = uni(gsegs[si].par23), fl = uni(gsegs[si].flags);
        const int d0 = (int)(dl & 0xFFFFu);
        const int nb = seg_nparents(p01, p23);
        const int64_t at = 1 + (int64_t)len - d0;
        uint32_t w;
        if (nb == 0 || (nb > 1 && !(fl & G2S_SEG_ORDERED)) || at < 1 || (uint64_t)at >= avail) w = 0x80000000u;
        else {
          const uint32_t rv = nb > 1 ? value((uint32_t)at) >> 1 : 0u;
          w = seg_parent(p01, p23, nb == 1 ? 0 : pick_parent(rv, nb));  // :1513
          if (w >= nsegs) w = 0x80000000u;
        }
        last = w | ((fl & G2S_SUB_SOURCE) ? 0x40000000u : 0u);
        if (lane == 0) ghop[nh] = (uint64_t)(uint32_t)si << 32;
        nh++;
        if (last & 0xC0000000u) break;
        si = (int)(last & 0xFFFFu);
      }
      __threadfence_block();
    }
    if (!(last & 0x40000000u)) bad = true;  // (ended without a way on — :1493-1510 — or past the depth of a source)
    wg_sync();  // (every wave has listed the chain — the same words — before any of them rewrites an entry below)
    // (ii) All lanes, a hop each: state t of a segment that begins at depth d0 sits at depth d0 + t; the walk enters
    // the first segment at state start_t and every other at its last state, and steps from a segment's first state
    // to the depth below it: the depth at which hop h is entered is len less the states passed before it — a scan —
    // and has to be d0 + t there.  Every state passed draws once (:1513), so the walk draws 1 + len - (the depth at
    // which it ends) times.
    int carry = 0, d_end = 0;
    for (int h0 = 0; !bad && h0 < nh; h0 += 64) {
      const int h = h0 + lane;
      const bool in = h < nh;
      const uint32_t sq = in ? ((big ? (uint32_t)(ghop[h] >> 32) : hop[h].y) & 0xFFFFu) : 0u;  // (another wave may have completed the entry already)
      const uint32_t dl = in ? (big ? gsegs[sq].depth_len : pk[sq].x) : 0u;
      const int d0 = (int)(dl & 0xFFFFu), t = h == 0 ? t0 : (int)(dl >> 16) - 1;  // (a child in the closure puts the whole parent there)
      int inc = in ? t + 1 : 0;
      for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(inc, o); if (lane >= o) inc += y; }
      const int at = len - carry - (inc - (in ? t + 1 : 0));  // the depth at which this hop is entered
      // (a hop that is not the last leaves its segment above depth 0: the step down exists)
      if (__ballot(in && (d0 + t != at || t < 0 || (h + 1 < nh && d0 < 1))) != 0ull) { bad = true; break; }
      if (in) { if (big) ghop[h] = (uint64_t)(uint32_t)at | ((uint64_t)(sq | ((uint32_t)t << 16)) << 32); else hop[h] = make_uint2((uint32_t)at, sq | ((uint32_t)t << 16)); }
      if (h + 1 == nh) d_end = d0;
      carry += __shfl(inc, 63);
    }
    if (!bad) {
      d2 = uni((uint32_t)__shfl(d_end, (nh - 1) & 63));
      draws = 1 + len - d2;
      left_fuz = (int)dg.lmf - d2;  // :1455-1462
      ended = true;
    }
  }
  if (!ended || draws != want) bad = true;
  if (P.laps) tkm = wall_clock64();
  if (!bad) {
    const int stop0 = (int)dg.lmf - left_fuz;  // the fill takes depths stop0 + 1 .. len (cmap index = depth - 1)
    wg_sync();
    // (a closure g2s_d2_* analysed: its runs — a binary search wherever the fill leaves one — come into LDS over the
    // rand() values, which the chain above was the last to read: eight dependent reads of device memory per search
    // made these gaps' waves the kernel's last, config 3: 213 instead of 196 us)
    if (by_runs && 2u * n_runs <= P.map_cap) {
      for (uint32_t w = (uint32_t)tid; w < 2u * n_runs; w += (uint32_t)NT) lwin[w] = runs[w];
      wg_sync();
      lruns = lwin;
    }
    const int npos = max(0, len - stop0), per = (npos + NT - 1) / NT;
    // thread t: depths (hi_d - cnt, hi_d], from the top of the fill downwards in thread order
    const int hi_d = len - tid * per, cnt = max(0, min(per, hi_d - stop0));
    int lowest_safe = 0x7FFFFFFF;
    if (cnt > 0) {
      // (hop h: the depth at which it is entered, the segment | its entry state << 16 — from LDS, or, a closure too
      // large for it, from where the chain above listed them)
      auto hop_at = [&](int h) -> uint2 {
        if (!big) return hop[h];
        const uint64_t w = ghop[h];
        return make_uint2((uint32_t)w, (uint32_t)(w >> 32));
      };
      auto seg_at = [&](uint32_t q) -> SegW { return big ? gsegs[q] : segs[q]; };
      int lo = 0, hi = nh;  // the last hop entered at or above hi_d
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)hop_at(mid).x >= hi_d) lo = mid; else hi = mid; }
      int h = lo;
      uint2 hr = hop_at(h);
      uint32_t rec = hr.y;
      int top = (int)hr.x, d0 = top - (int)(rec >> 16);
      SegW sg = seg_at(rec & 0xFFFFu);
      uint32_t rlo = 1u, rhi = 0u;  // the run the last k-mer lay in (none yet)
      bool rsafe = false;
      for (int c = 0; c < cnt; c++) {
        const int p = hi_d - c;
        if (p < d0) {  // the next hop begins right below
          h++;
          hr = hop_at(h);
          rec = hr.y;
          top = (int)hr.x; d0 = top - (int)(rec >> 16);
          sg = seg_at(rec & 0xFFFFu);
        }
        const int q = p - d0;
        const uint32_t idx0 = sg.node >> 1;
        const bool up = (sg.node & 1u) == 0u;
        const uint32_t x = up ? idx0 + (uint32_t)q : idx0 - (uint32_t)q;
        const int ts = (sg.ts_tt & 0x7FFFu) == 0x7FFFu ? -1 : (int)(sg.ts_tt & 0x7FFFu);
        bool sf;
        if (P.skip_confident) sf = true;
        else if (by_runs) {  // (consecutive bases are consecutive k-mers: a search only where a run ends)
          if (x < rlo || x > rhi) rsafe = run_verdict(x, &rlo, &rhi);
          sf = rsafe;
        }
        else if (q > ts) sf = outside_safe(x);
        else if (q > (int)sg.pad) sf = (sg.ts_tt & 0x80000000u) != 0;  // beyond the split: safe bit b
        else sf = (sg.ts_tt & 0x8000u) != 0;
        if (sf) lowest_safe = p;
        cmap[p - 1] = x | (up ? 0u : 0x40000000u) | (sf ? 0x20000000u : 0u);
      }
    }
    // the nearest safe depth above each lane's stretch (the walk begins with the top of the fill counting as safe)
    int above = lowest_safe;
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(above, o); if (lane >= o) above = min(above, y); }
    int waves_above = 0x7FFFFFFF;  // (the stretches of the waves in front of this one)
    if constexpr (NW > 1) {
      int* xw = (int*)(lds + (size_t)P.seg_cap * 12u + (size_t)P.map_cap * 2u);  // (the four spare words behind everything)
      if (lane == 63) xw[tid >> 6] = above;
      __syncthreads();
      for (int w = 0; w < (tid >> 6); w++) waves_above = min(waves_above, xw[w]);
    }
    above = __shfl_up(above, 1);
    if (lane == 0) above = 0x7FFFFFFF;
    above = min(above, waves_above);
    int last_solid = min(above, len);
    for (int c = 0; c < cnt; c++) {
      const int p = hi_d - c;
      const uint32_t e = cmap[p - 1];
      if (e & 0x20000000u) last_solid = p;
      else if (p <= last_solid - k) cmap[p - 1] = e | 0x80000000u;
    }
  }
  if (P.laps) tk2 = wall_clock64();
  if (bad) {  // (acknowledged before this wave is counted as through)
    if (tid == 0) atomicAdd(&S->anomalies, 1u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const int stop = (int)dg.lmf - left_fuz;  // the fill begins at this index of the buffer
  const uint32_t fill_len = (!bad && len >= stop && stop >= 0) ? (uint32_t)(len - stop) : 0u;
  if (!bad) {
    wg_sync();
    // ---- the bases: eight loads in flight per lane (a 1 000-base fill: two round trips on one wave).  (Measured and
    // not kept: the text in 4-byte words, 256 bytes an instruction — config 3's kernel 207 us against 205 with a byte a
    // lane, config 4's 57 against 54: the link takes what the kernel writes at ~38 GB/s either way.)
    const char* sbuf = spec ? W.spec_text + abs_off : nullptr;
    for (int p0 = stop; p0 < len; p0 += 8 * NT) {
      uint32_t e[8];
      char c[8], was[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int p = p0 + NT * u + tid; e[u] = p < len ? cmap[p] : 0u; }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t x = e[u] & 0x0FFFFFFFu;
        c[u] = (e[u] & 0x40000000u) ? chd[x] : chu[x];
        const int p = p0 + NT * u + tid;
        was[u] = (spec && p >= spec_stop && p < spec_len && p < len) ? sbuf[p] : (char)0;  // (0: no base — nothing was guessed there)
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int p = p0 + NT * u + tid;
        const char ch = (e[u] >> 31) ? (char)(c[u] | 0x20) : c[u];
        // (a wave's lanes hold 64 consecutive bases: sent when any of them differs from the guess)
        const uint64_t diff = __ballot(p < len && was[u] != ch), any = __ballot(p < len);
        if (any && spec) { g_all++; if (diff) g_sent++; }
        if (diff != 0ull && p < len) buf[p] = ch;
      }
    }
    if (tid == 0 && !(spec && spec_len == len)) buf[len] = '\0';

  } else if (tid == 0) buf[dg.lmf] = '\0';
  if (P.laps) tk3 = wall_clock64();
  rw[2] = (uint32_t)go.reached_j;  // :1171
  rw[3] |= G2S_GAP_PHASE_D;
  rw[7] = (uint32_t)draws;
  spec_rec = spec && !bad && W.spec_res != nullptr;
  gsp_all = g_all; gsp_sent = g_sent;
  // (one counter for the whole list made 10 000 waves queue at one address of the L2: 64 counters, a cache line each)
  finish(fill_len);
  leave(fill_len);
}

}  // namespace

namespace g2s {

void rand_tables_host(uint32_t* hi, uint32_t* mid, uint32_t* lane) {
  auto mulmod = [](const uint32_t* a, const uint32_t* b, uint32_t* out) {
    uint32_t c[61];
    for (int i = 0; i < 61; i++) c[i] = 0;
    for (int i = 0; i < 31; i++) {
      if (!a[i]) continue;
      for (int j = 0; j < 31; j++) c[i + j] += a[i] * b[j];
    }
    for (int d = 60; d >= 31; d--) { c[d - 3] += c[d]; c[d - 31] += c[d]; }  // x^d = x^(d-3) + x^(d-31)
    for (int i = 0; i < 31; i++) out[i] = c[i];
  };
  auto power = [&](uint64_t N, uint32_t* q) {  // x^N
    uint32_t r[31] = {0}, b[31] = {0}, t[31];
    r[0] = 1; b[1] = 1;
    for (; N; N >>= 1) {
      if (N & 1) { mulmod(r, b, t); for (int i = 0; i < 31; i++) r[i] = t[i]; }
      mulmod(b, b, t);
      for (int i = 0; i < 31; i++) b[i] = t[i];
    }
    for (int i = 0; i < 31; i++) q[i] = r[i];
  };
  auto series = [&](uint64_t step, int count, uint32_t* out) {  // x^(step * a), a < count
    uint32_t s[31];
    power(step, s);
    for (int i = 0; i < 31; i++) out[i] = i == 0 ? 1u : 0u;
    for (int a = 1; a < count; a++) mulmod(out + (size_t)(a - 1) * 31, s, out + (size_t)a * 31);
  };
  series(1ull << 20, 128, hi);
  series(4096, 256, mid);
  series(64, 64, lane);
}

size_t d3_work_bytes(uint32_t n) {
  const size_t per = (((size_t)n + 64) * 4 + 63) & ~(size_t)63;
  return 14 * per + 64 * 128 + (size_t)G2S_D3_TABLE_BUDGET * 2 + (size_t)(G2S_D3_TABLE_BUDGET / 4u) * 4 + 4096 +
         ((size_t)n + 64) * (sizeof(D3Var) + sizeof(D3Trace) + sizeof(D3HostItem)) + ((size_t)n + G2S_D3_TABLE_BUDGET / 256u + 64) * 4;
}

void d3_work_carve(void* p, uint32_t n, D3Work* w) {
  char* c = (char*)p;
  const size_t per = (((size_t)n + 64) * 4 + 63) & ~(size_t)63;
  w->sum = (D3Summary*)c; c += 1024;
  w->fill_bytes = (unsigned long long*)c; c += 64 * 128;
  w->ginfo = (uint32_t*)c; c += per;
  w->dmin = (uint32_t*)c; c += per;
  w->dspread = (uint32_t*)c; c += per;
  w->base = (uint32_t*)c; c += per;
  w->vrank = (uint32_t*)c; c += per;
  w->var_gap = (uint32_t*)c; c += per;
  w->var_R = (uint32_t*)c; c += per;
  w->var_toff = (uint32_t*)c; c += per;
  w->var_tile = (uint32_t*)c; c += per;
  w->blk_toff = (uint32_t*)c; c += per;
  w->blk_in = (uint32_t*)c; c += per;
  w->dvar = (uint32_t*)c; c += per;
  w->skip = (int32_t*)c; c += per;
  w->host_slot = (uint32_t*)c; c += per;
  w->vdesc = (D3Var*)c; c += ((size_t)n + 64) * sizeof(D3Var);     // (c is a multiple of 64 here)
  w->tdesc = (D3Trace*)c; c += ((size_t)n + 64) * sizeof(D3Trace);
  w->hitems = (D3HostItem*)c; c += ((size_t)n + 64) * sizeof(D3HostItem);
  w->tile_var = (uint32_t*)c; c += ((size_t)n + G2S_D3_TABLE_BUDGET / 256u + 64) * 4;
  w->btab = (uint32_t*)c; c += (size_t)(G2S_D3_TABLE_BUDGET / 4u) * 4;
  w->tab = (uint16_t*)c;
}

hipError_t launch_rand_fill(hipStream_t st, uint32_t* rnd_all, const RandTables& rt, const D3Summary* sum_dev, uint64_t capacity,
                            uint64_t first_value) {
  const uint64_t first_block = first_value / G2S_RAND_BLOCK;
  const uint64_t all = (capacity + G2S_RAND_BLOCK - 1) / G2S_RAND_BLOCK;
  const uint32_t rblocks = (uint32_t)std::min<uint64_t>(all > first_block ? all - first_block : 1, 4096);
  hipLaunchKernelGGL(g2s_rand_fill, dim3(std::max(1u, rblocks)), dim3(64), 0, st, rnd_all, rt, sum_dev, capacity, first_block);
  return hipGetLastError();
}

// g2s_d3_scan with the per-gap words in (dynamic) LDS when the list fits
static void launch_scan(hipStream_t st, const D3Params& P, const D3Work& W, const GapOut* outs) {
  static std::mutex mu;
  static std::set<int> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(mu);
    if (done.insert(dev).second)
      (void)hipFuncSetAttribute((const void*)g2s_d3_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(12u * D3_SCAN_LDS_GAPS));
  }
  const size_t lds = P.n <= D3_SCAN_LDS_GAPS ? 12u * (size_t)P.n : 0u;
  hipLaunchKernelGGL(g2s_d3_scan, dim3(1), dim3(1024), lds, st, P, W, outs);
}

hipError_t launch_rand_window(hipStream_t st, uint32_t* rnd_all, const uint32_t* link) {
  hipLaunchKernelGGL(g2s_rand_window, dim3(1), dim3(64), 0, st, rnd_all, link);
  return hipGetLastError();
}

hipError_t launch_d3_sharded_classes(hipStream_t st, const D3Params& P, const D3Work& W, const GapOut* outs, const D3Gap* dgaps,
                                     bool summary_is_clean) {
  if (P.n == 0) return hipSuccess;
  hipError_t e = summary_is_clean ? hipSuccess : hipMemsetAsync(W.sum, 0, 1024 + 64 * 128, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(g2s_d3_classify, dim3((P.n + 255u) / 256u), dim3(256), 0, st, P, W, outs, dgaps);
  launch_scan(st, P, W, outs);  // (P.base0 = P.R0 = 0: the group's totals)
  return hipGetLastError();
}
hipError_t launch_d3_sharded_tables(hipStream_t st, const D3Params& P, const D3Work& W, const GapOut* outs, const SubRec* sub,
                                    uint32_t* rnd_all, uint64_t rnd_capacity, uint32_t* group_fn) {
  if (P.n == 0) return hipSuccess;
  launch_scan(st, P, W, outs);  // (the layout again, behind the groups in front)
  const uint32_t tgrid = std::min(8192u, std::max(128u, P.n / 2u));
  const size_t win = ((size_t)P.map_cap + 256) * 4;
  hipError_t e = hipFuncSetAttribute((const void*)g2s_d3_tables, hipFuncAttributeMaxDynamicSharedMemorySize, (int)win);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(g2s_d3_tables, dim3(tgrid), dim3(256), win, st, P, W, sub, rnd_all + 31, rnd_capacity);
  hipLaunchKernelGGL(g2s_d3_blocks, dim3(256), dim3(256), 0, st, W);
  hipLaunchKernelGGL(g2s_d3_groupfn, dim3(std::min(256u, (P.R0 + 256u) / 256u)), dim3(256), 0, st, W, P.R0, group_fn);
  return hipGetLastError();
}
hipError_t launch_d3_sharded_trace(hipStream_t st, const D3Params& P, const D3Work& W, const GapOut* outs, const SubRec* sub,
                                   const char* lastch_up, const char* lastch_dn, uint32_t* rnd_all, uint64_t rnd_capacity,
                                   void* results, char* arena, const D3Side& side, void* summary_host, hipEvent_t ev_d2) {
  if (P.n == 0) return hipSuccess;
  hipLaunchKernelGGL(g2s_d3_chain, dim3(1), dim3(1024), 0, st, W, rnd_all, P.base0, P.d_in);
  if (ev_d2) { const hipError_t e2 = hipStreamWaitEvent(st, ev_d2, 0); if (e2 != hipSuccess) return e2; }
  hipLaunchKernelGGL(g2s_d3_handoff, dim3((P.n + 63u) / 64u), dim3(64), 0, st, P, W, outs, sub, rnd_all + 31, rnd_capacity, side);
  const size_t lds = (size_t)P.seg_cap * (sizeof(SegRec) + 16) + (size_t)P.map_cap * 8 + 16;
  hipError_t e = hipFuncSetAttribute((const void*)g2s_d3_trace<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(g2s_d3_trace<1>, dim3(P.n), dim3(64), lds, st, P, W, const_cast<GapOut*>(outs), sub, lastch_up, lastch_dn, rnd_all + 31, rnd_capacity,
                     (g2s_result*)results, arena, (uint32_t*)summary_host, side, (uint32_t*)nullptr);
  return hipGetLastError();
}

hipError_t launch_d3(hipStream_t st, const D3Params& P, const D3Work& W, const GapDev* gaps, const GapOut* outs,
                     const D3Gap* dgaps, const SubRec* sub, const char* lastch_up, const char* lastch_dn,
                     const RandTables& rt, uint32_t* rnd_all, uint64_t rnd_capacity, void* results, char* arena,
                     const D3Side& side, void* summary_host, bool summary_is_clean, uint32_t* clean_words, hipEvent_t ev_chain,
                     hipEvent_t ev_d2, hipEvent_t ev_stop) {
  if (P.n == 0) return hipSuccess;
  (void)gaps;
  (void)rt;
  static_assert(sizeof(D3Summary) <= 1024, "summary slot");
  // (the summary and, behind it, the 64 fill-byte counters of the trace kernel: the caller zeroes them behind the
  // previous list, off this one's critical path — a memset in front of the first kernel costs the stream 10-15 us)
  hipError_t e = summary_is_clean ? hipSuccess : hipMemsetAsync(W.sum, 0, 1024 + 64 * 128, st);
  if (e != hipSuccess) return e;
  const bool short_list = P.n <= D3_FRONT_GAPS;
  if (short_list) hipLaunchKernelGGL(g2s_d3_front, dim3(1), dim3(1024), 0, st, P, W, outs, dgaps);
  else {
    hipLaunchKernelGGL(g2s_d3_classify, dim3((P.n + 255u) / 256u), dim3(256), 0, st, P, W, outs, dgaps);
    launch_scan(st, P, W, outs);
  }
  // (a tile of 256 deviations per workgroup, grid-stride: a short list has a few dozen tiles)
  const uint32_t tgrid = std::min(8192u, std::max(128u, P.n / 2u));
  const size_t win = ((size_t)P.map_cap + 256) * 4;
  e = hipFuncSetAttribute((const void*)g2s_d3_tables, hipFuncAttributeMaxDynamicSharedMemorySize, (int)win);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(g2s_d3_tables, dim3(tgrid), dim3(256), win, st, P, W, sub, rnd_all + 31, rnd_capacity);
  // (g2s_d2_* runs on a stream of its own beside the kernels above; the hand-off reads what it decided)
  if (ev_d2 && short_list) { e = hipStreamWaitEvent(st, ev_d2, 0); if (e != hipSuccess) return e; }
  if (short_list) {
    hipLaunchKernelGGL(g2s_d3_back, dim3(1), dim3(1024), 0, st, P, W, outs, sub, rnd_all, rnd_capacity, side);
    if (ev_chain) { e = hipEventRecord(ev_chain, st); if (e != hipSuccess) return e; }
  } else {
    hipLaunchKernelGGL(g2s_d3_blocks, dim3(256), dim3(256), 0, st, W);
    hipLaunchKernelGGL(g2s_d3_chain, dim3(1), dim3(1024), 0, st, W, rnd_all, 0u, 0u);
    if (ev_chain) { e = hipEventRecord(ev_chain, st); if (e != hipSuccess) return e; }
    if (ev_d2) { e = hipStreamWaitEvent(st, ev_d2, 0); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(g2s_d3_handoff, dim3((P.n + 63u) / 64u), dim3(64), 0, st, P, W, outs, sub, rnd_all + 31, rnd_capacity, side);
  }
  const size_t lds = (size_t)P.seg_cap * (sizeof(SegRec) + 16) + (size_t)P.map_cap * 8 + 16;  // closure, base map, rand() values, the walk's segments
  // (a short list: four waves per gap — the kernel is its slowest gap, and the chip has the wave slots)
  const char* waves_s = GENV("G2S_TRACE_WAVES");  // (read per launch: the tests switch it within a process)
  const int waves_env = waves_s ? atoi(waves_s) : 0;
  const bool four = waves_env ? waves_env == 4 : P.n <= 768u;  // (2 000 gaps on four waves each: 1-2 % slower than on one)
  e = hipFuncSetAttribute(four ? (const void*)g2s_d3_trace<4> : (const void*)g2s_d3_trace<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  if (ev_stop != nullptr) {  // (a timed list: the dispatch's own stop time — an event recorded behind the kernel is a packet of its own)
    if (four)
      hipExtLaunchKernelGGL(g2s_d3_trace<4>, dim3(P.n), dim3(256), (uint32_t)lds, st, nullptr, ev_stop, 0u, P, W, const_cast<GapOut*>(outs), sub, lastch_up, lastch_dn,
                            rnd_all + 31, rnd_capacity, (g2s_result*)results, arena, (uint32_t*)summary_host, side, clean_words);
    else
      hipExtLaunchKernelGGL(g2s_d3_trace<1>, dim3(P.n), dim3(64), (uint32_t)lds, st, nullptr, ev_stop, 0u, P, W, const_cast<GapOut*>(outs), sub, lastch_up, lastch_dn,
                            rnd_all + 31, rnd_capacity, (g2s_result*)results, arena, (uint32_t*)summary_host, side, clean_words);
    return hipGetLastError();
  }
  if (four)
    hipLaunchKernelGGL(g2s_d3_trace<4>, dim3(P.n), dim3(256), lds, st, P, W, const_cast<GapOut*>(outs), sub, lastch_up, lastch_dn, rnd_all + 31, rnd_capacity,
                       (g2s_result*)results, arena, (uint32_t*)summary_host, side, clean_words);
  else
    hipLaunchKernelGGL(g2s_d3_trace<1>, dim3(P.n), dim3(64), lds, st, P, W, const_cast<GapOut*>(outs), sub, lastch_up, lastch_dn, rnd_all + 31, rnd_capacity,
                       (g2s_result*)results, arena, (uint32_t*)summary_host, side, clean_words);
  return hipGetLastError();
}

}  // namespace g2s
