// gap2seq_amd/csrc/d3_device.hip — phase D3 of fill_gap on the device, for gfx950 (CDNA4).
//
// The reference draws rand() once per gap for the path length and once per traced base, in gap order
// (/root/reference/src/Gap2Seq.cpp:178,1440,1513): gap i's traceback reads the stream at the sum of the draws
// of all gaps before it.  Until round 2 the host did this part: an in-order pass over the gaps for the offsets
// (serial: ~4 % of the gaps draw a number of values that depends on the values drawn) and the tracebacks on a
// thread pool — 1.8 ms of host work beside a 0.48 ms kernel on a 10 000-gap list, and not divided by the
// number of GPUs.  Here the whole of it runs behind the fill kernel on its stream, with no host round trip:
//
//   g2s_d3_scan    one workgroup: every gap's class (no phase D / fixed draw count / draw-dependent), its
//                  (fewest) draws and their spread, the skip rule of consecutive gaps of a record (:369,402),
//                  the prefix sums over the list and the layout of the tables below.
//   g2s_rand_fill  the glibc TYPE_3 stream (x[n] = x[n-31] + x[n-3], the linear recurrence behind rand())
//                  materialised from the session's position on: a wave computes the state in front of its
//                  4096 values from the state the host hands over with three jump polynomials
//                  (x^(2^20 a) x^(4096 b) x^(64 l) modulo x^31 - x^28 - 1 over Z/2^32, seed independent
//                  tables), then every lane runs the recurrence for its 64 values.
//   g2s_d3_tables  A draw-dependent gap v can only start at base_v + d, d in [0, R_v], R_v = the summed spreads of
//                  the draw-dependent gaps before it: its draw count for EVERY such start, one lane per (v, d) —
//                  the serial chain "offset of v+1 = offset of v + draws of v" becomes table look-ups.
//   g2s_d3_blocks  the chain through 16 consecutive tables for every deviation a block can start with;
//   g2s_d3_chain   the chain over blocks (one lane), then back into the blocks: the deviation in front of
//                  every draw-dependent gap, hence every gap's offset.
//   g2s_d3_trace   one wave per gap: the traceback over the closure segments (post.cpp: seg_traceback is
//                  the host version and the reference for every branch here), fill text and result record
//                  written where the caller wants them (pinned host memory, or a staging buffer).
//
// Anything out of the ordinary (a gap the segment tier could not finish or analyse, tables beyond the
// budget, a walk that does not end in a left-flank k-mer) is only COUNTED here: the host then discards the
// attempt and runs the list through the host path (g2s_api.hip), which remains the authority.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/g2s.h"
#include "d3_device.h"

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
__device__ __forceinline__ unsigned long long wave_add64(unsigned long long v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ void d3_classify_body(const D3Params& P, const D3Work& W, const GapOut* __restrict__ outs,
                                                 const D3Gap* __restrict__ dgaps, uint32_t first, uint32_t stride) {
  const uint32_t n = P.n;
  unsigned long long xA = 0, sA = 0, xB = 0, sB = 0, xD = 0, sD = 0, segs = 0;
  uint32_t unhandled = 0, seg_gaps = 0;
  for (uint32_t i = first; i < n; i += stride) {
    const D3Gap dg = dgaps[i];
    uint32_t gi = 0, dmin = 0, spread = 0;
    if (dg.kind != 0) gi = GI_BAD;
    else {
      const GapOut& go = outs[i];
      if (go.flags & (G2S_DEV_OVERFLOW_A | G2S_DEV_OVERFLOW_B)) unhandled++;
      else if ((uint64_t)go.n_states > P.max_states || (uint64_t)go.n_right > P.max_states) gi = GI_MEM;
      else {
        xA += go.x_right; sA += go.n_right; xB += go.x_left; sB += go.n_states; xD += go.x_sub; sD += go.n_sub;
        segs += go.stat[3];
        seg_gaps++;
        const bool phase_d = go.c_count > 0 && go.n_len > 0;  // :1169
        const bool by_host = phase_d && !(go.dflags & G2S_DEVA_ANALYSED);
        // (the all-paths recount — the sum of the counts of the sink states — is the kernel's for every closure,
        // also for those the host analyses: whether a gap counts as filled decides the skip rule of the next)
        const int cnt = gap_count(go, P, phase_d);
        if (cnt > 0 && (!P.unique_paths || cnt == 1)) gi |= GI_FILLED;
        if (go.n_len > 0) gi |= ((uint32_t)go.reached_j & 0xFFu) << 8;
        if (phase_d) {
          gi |= GI_PHASE_D;
          if (by_host) gi |= GI_HOST;
          int lo_d = 0x7FFFFFFF, hi_d = 0;
#pragma unroll
          for (int q = 0; q < 2; q++) {
            if (q >= go.n_len) continue;
            // stop depths behind traceback start q: lowest | highest << 16 (fill_seg.hip); a traceback stops at a
            // left-flank k-mer, i.e. at a depth in [0, lmf]
            const uint32_t sw = q ? go.stop[1] : go.stop[0];
            const int lq = q ? go.len[1] : go.len[0];
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
    W.ginfo[i] = gi;
    W.dmin[i] = dmin;
    W.dspread[i] = spread;
  }
  xA = wave_add64(xA); sA = wave_add64(sA); xB = wave_add64(xB); sB = wave_add64(sB); xD = wave_add64(xD); sD = wave_add64(sD);
  segs = wave_add64(segs);
  const unsigned long long cnts = wave_add64(((unsigned long long)unhandled << 32) | seg_gaps);
  if ((threadIdx.x & 63u) == 0u) {
    D3Summary* S = W.sum;
    atomicAdd((unsigned long long*)&S->xA, xA); atomicAdd((unsigned long long*)&S->sA, sA);
    atomicAdd((unsigned long long*)&S->xB, xB); atomicAdd((unsigned long long*)&S->sB, sB);
    atomicAdd((unsigned long long*)&S->xD, xD); atomicAdd((unsigned long long*)&S->sD, sD);
    atomicAdd((unsigned long long*)&S->segs, segs);
    if (cnts >> 32) atomicAdd(&S->unhandled, (uint32_t)(cnts >> 32));
    if (cnts & 0xFFFFFFFFull) atomicAdd(&S->seg_gaps, (uint32_t)cnts);
  }
}
__global__ __launch_bounds__(256) void g2s_d3_classify(const D3Params P, const D3Work W, const GapOut* __restrict__ outs,
                                                       const D3Gap* __restrict__ dgaps) {
  d3_classify_body(P, W, outs, dgaps, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// ---------------------------------------------------------------------------------------------------------
// scan: the skip rule, prefix sums in list order, table layout.  One workgroup of 1024 threads over the compact
// per-gap arrays; thread t owns the contiguous range [t * per, (t + 1) * per) of the list.
// ---------------------------------------------------------------------------------------------------------
// exclusive prefix sums of (a, b, c) over the 1024 threads of the workgroup; totals in tot[]
__device__ __forceinline__ void block_scan3(uint64_t& a, uint64_t& b, uint32_t& c, uint64_t* sh64 /* [34] */, uint32_t* sh32 /* [17] */,
                                            uint64_t* tot_a, uint64_t* tot_b, uint32_t* tot_c) {
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  uint64_t ia = a, ib = b;
  uint32_t ic = c;
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t ya = __shfl_up(ia, o), yb = __shfl_up(ib, o);
    const uint32_t yc = __shfl_up(ic, o);
    if (lane >= o) { ia += ya; ib += yb; ic += yc; }
  }
  if (lane == 63) { sh64[wave] = ia; sh64[17 + wave] = ib; sh32[wave] = ic; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t xa = 0, xb = 0;
    uint32_t xc = 0;
    for (int w = 0; w < 16; w++) {
      const uint64_t ta = sh64[w], tb = sh64[17 + w];
      const uint32_t tc = sh32[w];
      sh64[w] = xa; sh64[17 + w] = xb; sh32[w] = xc;
      xa += ta; xb += tb; xc += tc;
    }
    sh64[16] = xa; sh64[33] = xb; sh32[16] = xc;
  }
  __syncthreads();
  const uint64_t wa = sh64[wave], wb = sh64[17 + wave];
  const uint32_t wc = sh32[wave];
  *tot_a = sh64[16]; *tot_b = sh64[33]; *tot_c = sh32[16];
  a = wa + ia - a; b = wb + ib - b; c = wc + ic - c;
  __syncthreads();
}

__device__ __forceinline__ void d3_scan_body(const D3Params& P, const D3Work& W, const D3Gap* __restrict__ dgaps, uint64_t* sh64 /* [34] */,
                                             uint32_t* sh32 /* [17] */, uint32_t* sh_f /* [1024] */) {
  const uint32_t t = threadIdx.x, n = P.n;
  const uint32_t per = (n + 1023u) / 1024u;
  const uint32_t lo = min(n, t * per), hi = min(n, lo + per);
  // ---- the skip rule (:369: a gap is not attempted when the previous gap of its record was filled with a right
  // fuz beyond the distance between them): skipped_i = s_i && !skipped_(i-1), s_i from gap i-1's own result
  if (P.has_skip) {
    auto s_of = [&](uint32_t i) -> bool {
      if (i == 0) return false;
      const int thr = dgaps[i].skip_thr;
      if (thr < 0) return false;
      const uint32_t pg = W.ginfo[i - 1];
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
    for (uint32_t i = lo; i < hi; i++) {
      sk = s_of(i) && !sk;
      if (sk) { W.ginfo[i] = (W.ginfo[i] & ~3u) | GI_SKIPPED; W.dmin[i] = 0; W.dspread[i] = 0; }
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
      g8[q] = have ? W.ginfo[i] : GI_SKIPPED;  // (beyond the range: reads as a gap that contributes nothing)
      m8[q] = have ? W.dmin[i] : 0u;
      s8[q] = have ? W.dspread[i] : 0u;
    }
  };
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
  block_scan3(sd, ss, nv, sh64, sh32, &tot_d, &tot_s, &V);
  const bool too_wide = tot_d + tot_s >= 0xFFFF0000ull || tot_s >= (uint64_t)G2S_D3_TABLE_BUDGET;
  uint64_t my_tab = 0, my_blk = 0;
  uint32_t my_tiles = 0;
  {
    uint64_t d = sd, r = ss;
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
    if (t == 1023u) W.var_R[V] = (uint32_t)tot_s;
  }
  // ---- where every table starts
  uint64_t to = my_tab, bo = my_blk, T, TB;
  uint32_t tl = my_tiles, tiles;
  block_scan3(to, bo, tl, sh64, sh32, &T, &TB, &tiles);
  const bool over = too_wide || T > (uint64_t)G2S_D3_TABLE_BUDGET || TB > (uint64_t)(G2S_D3_TABLE_BUDGET / 4u);
  if (!over) {
    uint64_t r = ss;
    uint32_t v = nv;
    for (uint32_t i0 = lo; i0 < hi; i0 += 8u) {
      uint32_t g8[8], m8[8], s8[8];
      load8(i0, g8, m8, s8);
#pragma unroll
      for (int q = 0; q < 8; q++) {
        if ((g8[q] & GI_SKIPPED) || GI_CLASS(g8[q]) != 2u) continue;
        W.var_toff[v] = (uint32_t)to;
        W.var_tile[v] = tl;
        tl += (uint32_t)((r + D3_TILE) / D3_TILE);
        to += r + 1u;
        if (v % G2S_D3_BLOCK_VARS == 0u) { W.blk_toff[v / G2S_D3_BLOCK_VARS] = (uint32_t)bo; bo += r + 1u; }
        v++;
        r += s8[q];
      }
    }
    if (t == 1023u) { W.var_toff[V] = (uint32_t)T; W.var_tile[V] = tiles; W.blk_toff[(V + G2S_D3_BLOCK_VARS - 1u) / G2S_D3_BLOCK_VARS] = (uint32_t)TB; }
  }
  if (t == 0) {
    D3Summary* S = W.sum;
    S->status = (S->unhandled ? G2S_D3_UNHANDLED : 0u) | (over ? G2S_D3_BUDGET : 0u);
    S->n_var = V;
    S->table_entries = T;
    S->block_entries = TB;
    S->draws_min = tot_d;
    S->draws_spread = tot_s;
  }
}
__global__ __launch_bounds__(1024) void g2s_d3_scan(const D3Params P, const D3Work W, const D3Gap* __restrict__ dgaps) {
  __shared__ uint64_t sh64[34];
  __shared__ uint32_t sh32[17];
  __shared__ uint32_t sh_f[1024];
  d3_scan_body(P, W, dgaps, sh64, sh32, sh_f);
}

// ---------------------------------------------------------------------------------------------------------
// the rand() stream.  W[0 .. 31) is the generator's state in front of the next value, W[31 + k] the k-th value
// the session will draw (raw words: rand() returns word >> 1).  The host hands over W[0 .. G2S_RAND_WINDOW).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void g2s_rand_fill(uint32_t* __restrict__ Wd, const g2s::RandTables rt, const D3Summary* sum,
                                                    uint64_t capacity) {
  __shared__ uint32_t base[G2S_RAND_WINDOW], s1[96], s2[64];
  // (sum: only as far as the list can draw, known once its scan has run; nullptr: the whole capacity, for a launch
  // beside the fill kernel)
  if (sum && sum->status) return;
  const uint64_t need = sum ? min(capacity, sum->draws_min + sum->draws_spread + 64ull) : capacity;
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = lane; i < G2S_RAND_WINDOW; i += 64u) base[i] = Wd[i];
  __syncthreads();
  const uint64_t nblocks = (need + G2S_RAND_BLOCK - 1u) / G2S_RAND_BLOCK;
  for (uint64_t B = blockIdx.x; B < nblocks; B += gridDim.x) {
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

// What a walk needs of a closure segment: entry depth | length << 16, the parents, the flags.
struct SegLite { uint32_t depth_len, par01, par23, flags; };

// rand() draws of the traceback of one gap whose draws start at rnd[0]; *bad: something the host must look at
__device__ int d3_walk_count(int n_len, int len0, int len1, uint32_t start_seg, uint32_t start_t, int nsegs,
                             const SegLite* __restrict__ segs, const uint32_t* __restrict__ rnd,
                             uint64_t avail /* values that exist from rnd[0] on */, bool* bad) {
  int draws = 1;
  if (avail < 1) { *bad = true; return draws; }
  const int pick = n_len > 1 ? (int)((rnd[0] >> 1) % (uint32_t)n_len) : 0;  // :1440
  int d2 = pick ? len1 : len0;
  const uint32_t sg = (start_seg >> (16 * pick)) & 0xFFFFu;
  int i = sg == 0xFFFFu ? -1 : (int)sg;
  int t = (int)((start_t >> (16 * pick)) & 0xFFFFu);
  if (i < 0 || i >= nsegs) { *bad = true; return draws; }
  SegLite s = segs[i];
  bool ended = false;
  for (int guard = 0; d2 >= 0; guard++) {
    if (guard > 70000) { *bad = true; break; }
    if (t == 0 && (s.flags & G2S_SUB_SOURCE)) { ended = true; break; }  // :1455-1462
    if (d2 > 0) {
      if (t > 0) { const int run = min(t, d2); draws += run; d2 -= run; t -= run; continue; }
      const int nb = seg_nparents(s.par01, s.par23);
      if (nb == 0 || (nb > 1 && !(s.flags & G2S_SEG_ORDERED)) || (uint64_t)draws >= avail) { *bad = true; break; }
      const uint32_t rv = nb > 1 ? rnd[draws] >> 1 : 0u;  // (rand() % 1: the value does not matter)
      draws++;
      i = (int)seg_parent(s.par01, s.par23, nb == 1 ? 0 : (int)(rv % (uint32_t)nb));  // :1513, GATB predecessor order
      if (i >= nsegs) { *bad = true; break; }
      s = segs[i];
      t = (int)(s.depth_len >> 16) - 1;
    }
    d2--;
  }
  if (!ended) *bad = true;  // (the host path decides what an unfinished walk means, :1493-1510)
  return draws;
}

// draws of draw-dependent gap v when its draws start at base + d, for every d it can meet: a workgroup takes 256
// consecutive deviations of one gap: the closure's links in LDS, one walk per thread
#define D3_TAB_SEGS 512u
// (the body: sub-block `sb` of `nsb` sub-blocks of 256 threads takes tiles sb, sb + nsb, ...; every thread of the
// workgroup passes the same number of barriers)
__device__ __forceinline__ void d3_tables_body(const D3Params& P, const D3Work& W, const GapOut* __restrict__ outs,
                                               const SubRec* __restrict__ sub, const uint32_t* __restrict__ rnd, uint64_t capacity,
                                               SegLite* lseg /* this sub-block's [D3_TAB_SEGS] */, uint32_t sb, uint32_t nsb, uint32_t tid /* 0 .. 255 */) {
  const D3Summary* S = W.sum;
  const uint32_t V = S->n_var;
  const uint32_t tiles = V ? W.var_tile[V] : 0u;
  bool bad_any = false;
  for (uint32_t t0 = 0; t0 < tiles; t0 += nsb) {
    const uint32_t tile = t0 + sb;
    const bool have = tile < tiles;
    uint32_t v = 0, i = 0, ns = 0;
    if (have) {
      uint32_t lo = 0, hi = V;  // last v with var_tile[v] <= tile
      while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (W.var_tile[mid] <= tile) lo = mid; else hi = mid; }
      v = lo;
      i = W.var_gap[v];
      ns = outs[i].n_xl;
    }
    __syncthreads();
    if (have) {
      const SegW* gs = (const SegW*)(sub_of(P, sub, i) + outs[i].sub_off);
      for (uint32_t q = tid; q < ns && q < D3_TAB_SEGS; q += 256u) {
        SegLite l;
        l.depth_len = gs[q].depth_len; l.par01 = gs[q].par01; l.par23 = gs[q].par23; l.flags = gs[q].flags;
        lseg[q] = l;
      }
    }
    __syncthreads();
    if (!have) continue;
    const GapOut& go = outs[i];
    const uint32_t R = W.var_R[v];
    const uint32_t d = (tile - W.var_tile[v]) * D3_TILE + tid;
    if (d <= R) {
      bool bad = ns > D3_TAB_SEGS;
      const uint64_t at = (uint64_t)W.base[i] + d;
      const int draws = bad ? 0 : d3_walk_count(go.n_len, go.len[0], go.len[1], go.start_seg, go.start_t, (int)ns, lseg, rnd + at,
                                                capacity > at ? capacity - at : 0ull, &bad);
      const int dev = draws - (int)W.dmin[i];
      if (dev < 0 || dev > (int)W.dspread[i]) bad = true;
      W.tab[(uint64_t)W.var_toff[v] + d] = bad ? (uint16_t)0 : (uint16_t)dev;
      bad_any |= bad;
    }
  }
  if (bad_any) atomicAdd(&W.sum->anomalies, 1u);
}
__global__ __launch_bounds__(256) void g2s_d3_tables(const D3Params P, const D3Work W, const GapOut* __restrict__ outs,
                                                     const SubRec* __restrict__ sub,
                                                     const uint32_t* __restrict__ rnd /* first upcoming value */, uint64_t capacity) {
  __shared__ SegLite lseg[D3_TAB_SEGS];
  if (W.sum->status) return;
  d3_tables_body(P, W, outs, sub, rnd, capacity, lseg, blockIdx.x, gridDim.x, threadIdx.x);
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
__device__ __forceinline__ void d3_chain_body(const D3Work& W, const uint32_t* __restrict__ Wd, uint32_t* total_dev /* shared */) {
  D3Summary* S = W.sum;
  const uint32_t V = S->n_var;
  const uint32_t NB = (V + G2S_D3_BLOCK_VARS - 1u) / G2S_D3_BLOCK_VARS;
  if (threadIdx.x == 0) {
    uint32_t d = 0;
    for (uint32_t b = 0; b < NB; b++) { W.blk_in[b] = d; d = W.btab[(uint64_t)W.blk_toff[b] + d]; }
    W.dvar[V] = d;
    *total_dev = d;
  }
  __threadfence_block();
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < NB; b += blockDim.x) {
    uint32_t d = W.blk_in[b];
    const uint32_t v1 = min(V, (b + 1u) * G2S_D3_BLOCK_VARS);
    for (uint32_t v = b * G2S_D3_BLOCK_VARS; v < v1; v++) { W.dvar[v] = d; d += W.tab[(uint64_t)W.var_toff[v] + d]; }
  }
  const uint64_t total = S->draws_min + *total_dev;
  if (threadIdx.x == 0) S->draws_total = total;
  if (threadIdx.x < 31u) S->rand_state[threadIdx.x] = Wd[total + threadIdx.x];
}
__global__ __launch_bounds__(1024) void g2s_d3_chain(const D3Work W, const uint32_t* __restrict__ Wd) {
  __shared__ uint32_t total_dev;
  if (W.sum->status) return;
  d3_chain_body(W, Wd, &total_dev);
}

// Short lists: five launches cost a 2 000-gap list more than the work in them — classes and scan in one launch
// of one workgroup, blocks and chain in another; the tables keep their own (they want the whole chip).
__global__ __launch_bounds__(1024) void g2s_d3_front(const D3Params P, const D3Work W, const GapOut* __restrict__ outs,
                                                     const D3Gap* __restrict__ dgaps) {
  __shared__ uint64_t sh64[34];
  __shared__ uint32_t sh32[17];
  __shared__ uint32_t sh_f[1024];
  d3_classify_body(P, W, outs, dgaps, threadIdx.x, 1024u);
  __threadfence();
  __syncthreads();
  d3_scan_body(P, W, dgaps, sh64, sh32, sh_f);
}

// ---------------------------------------------------------------------------------------------------------
// hand-off: the gaps whose closure the host analyses (GI_HOST) get their record, closure segments and the rand()
// values of their traceback copied into pinned memory — in front of the trace kernel, so that the host
// finishes them while that kernel runs.  One wave per 64 gaps of the list.
// ---------------------------------------------------------------------------------------------------------
// (the body: one wave, the 64 gaps from i0 on; true when it handed something over)
__device__ __forceinline__ bool d3_handoff_body(const D3Params& P, const D3Work& W, const GapOut* __restrict__ outs,
                                                const SubRec* __restrict__ sub, const uint32_t* __restrict__ rnd, uint64_t capacity,
                                                const g2s::D3Side& side, uint32_t i0) {
  D3Summary* S = W.sum;
  const int lane = (int)(threadIdx.x & 63u);
  const uint32_t mine = i0 + (uint32_t)lane;
  const bool host = mine < P.n && (W.ginfo[mine] & GI_HOST) != 0u;
  for (uint64_t m = __ballot(host); m; m &= m - 1) {
    const uint32_t i = i0 + (uint32_t)__builtin_ctzll(m);
    const uint32_t gi = W.ginfo[i];
    const GapOut& go = outs[i];
    const uint32_t vr = W.vrank[i];
    const uint32_t off = W.base[i] + W.dvar[vr];
    const uint32_t want = W.dmin[i] + (GI_CLASS(gi) == 2u ? (uint32_t)W.tab[(uint64_t)W.var_toff[vr] + W.dvar[vr]] : 0u);
    const uint32_t ns = go.n_xl;
    unsigned long long it = 0, so = 0, ro = 0;
    if (lane == 0) {
      it = atomicAdd(&S->host_items, 1ull);
      so = atomicAdd(&S->host_segs, (unsigned long long)ns);
      ro = atomicAdd(&S->host_rnd, (unsigned long long)want + 1ull);
    }
    it = __shfl(it, 0); so = __shfl(so, 0); ro = __shfl(ro, 0);
    if (it >= side.cap_items || so + ns > side.cap_segs || ro + want + 1ull > side.cap_rnd || (uint64_t)off + want + 1ull > capacity) {
      if (lane == 0) atomicAdd(&S->anomalies, 1u);
      continue;
    }
    if ((uint32_t)lane < sizeof(GapOut) / 4u) ((uint32_t*)&side.outs[it])[lane] = ((const uint32_t*)&go)[lane];
    const uint4* src = (const uint4*)(sub_of(P, sub, i) + go.sub_off);
    uint4* dst = (uint4*)(side.segs + so);
    for (uint32_t w = (uint32_t)lane; w < 2u * ns; w += 64u) dst[w] = src[w];
    for (uint32_t w = (uint32_t)lane; w < want + 1u; w += 64u) side.rnd[ro + w] = rnd[off + w];
    if (lane == 0) {
      g2s::D3HostItem h;
      h.gap = i; h.n_segs = ns; h.seg_off = so; h.rnd_off = ro; h.draws = want; h.pad = 0;
      side.items[it] = h;
    }
  }
  return __ballot(host) != 0ull;
}
__global__ __launch_bounds__(64) void g2s_d3_handoff(const D3Params P, const D3Work W, const GapOut* __restrict__ outs,
                                                     const SubRec* __restrict__ sub, const uint32_t* __restrict__ rnd,
                                                     uint64_t capacity, const g2s::D3Side side) {
  D3Summary* S = W.sum;
  if (S->status) return;
  // the number of items, where the host reads it once this kernel's event has fired
  if (d3_handoff_body(P, W, outs, sub, rnd, capacity, side, blockIdx.x * 64u)) __threadfence_system();
  if ((threadIdx.x & 63u) == 0u) {
    const unsigned int done = atomicAdd(&S->handoff_waves, 1u) + 1u;
    if (done == gridDim.x) {
      __threadfence_system();
      *side.count = (unsigned long long)S->host_items | ((unsigned long long)(S->anomalies ? 1u : 0u) << 63);
    }
  }
}
// (short lists: blocks, chain and the hand-off in one launch of one workgroup)
__global__ __launch_bounds__(1024) void g2s_d3_back(const D3Params P, const D3Work W, const GapOut* __restrict__ outs,
                                                    const SubRec* __restrict__ sub, const uint32_t* __restrict__ Wd, uint64_t capacity,
                                                    const g2s::D3Side side) {
  __shared__ uint32_t total_dev;
  if (W.sum->status) return;
  d3_blocks_body(W, threadIdx.x, 1024u);
  __threadfence();
  __syncthreads();
  d3_chain_body(W, Wd, &total_dev);
  __threadfence();
  __syncthreads();
  bool any = false;
  for (uint32_t i0 = (threadIdx.x >> 6) * 64u; i0 < P.n; i0 += 1024u) any |= d3_handoff_body(P, W, outs, sub, Wd + 31, capacity, side, i0);
  if (any) __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    *side.count = (unsigned long long)W.sum->host_items | ((unsigned long long)(W.sum->anomalies ? 1u : 0u) << 63);
  }
}

// ---------------------------------------------------------------------------------------------------------
// the tracebacks: one wave per gap.  Writes g2s_result[i] in full and the gap's fill text.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void g2s_d3_trace(const D3Params P, const D3Work W, const GapDev* __restrict__ gaps,
                                                   const GapOut* __restrict__ outs, const D3Gap* __restrict__ dgaps,
                                                   const SubRec* __restrict__ sub, const char* __restrict__ chu,
                                                   const char* __restrict__ chd, const uint32_t* __restrict__ rnd,
                                                   uint64_t capacity, g2s_result* __restrict__ results,
                                                   char* __restrict__ arena) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  D3Summary* S = W.sum;
  if (S->status) return;
  const uint32_t i = blockIdx.x;
  const int lane = (int)threadIdx.x;
  SegW* segs = (SegW*)lds;  // the gap's closure segments
  static_assert(sizeof(g2s_result) == 192, "g2s_result layout");
  const D3Gap dg = dgaps[i];
  const uint32_t gi = W.ginfo[i];
  const uint64_t abs_off = P.arena_base + dg.arena_off;
  char* buf = arena + abs_off;
  const int k = P.k;
  (void)gaps;
  // the record, words 0-23 of g2s_result (count, left_fuz, right_fuz, flags, fill_off, fill_len, draws, six 64-bit
  // subgraph statistics, phaseC_count, n_lengths, lengths[2]); backtrace_msg (words 24-47) becomes the empty string
  uint32_t rw[24];
#pragma unroll
  for (int q = 0; q < 24; q++) rw[q] = 0u;
  int left_fuz = 0;
  auto finish = [&](uint32_t fill_len) {
    if (lane == 0) {
      const uint64_t fo = abs_off + (uint64_t)((int)dg.lmf - left_fuz);
      rw[1] = (uint32_t)left_fuz;
      rw[4] = (uint32_t)fo; rw[5] = (uint32_t)(fo >> 32);
      rw[6] = fill_len;
      uint4* dst = (uint4*)&results[i];
#pragma unroll
      for (int q = 0; q < 6; q++) dst[q] = make_uint4(rw[4 * q], rw[4 * q + 1], rw[4 * q + 2], rw[4 * q + 3]);
      dst[6] = make_uint4(0u, 0u, 0u, 0u);  // backtrace_msg = "" (the rest of its 96 bytes is not touched)
    }
  };
  if (gi & (GI_BAD | GI_SKIPPED | GI_MEM)) {
    rw[3] = (gi & GI_BAD) ? G2S_GAP_BAD_FLANK : (gi & GI_SKIPPED) ? G2S_GAP_SKIPPED : G2S_GAP_MEM_EXCEEDED;
    if (gi & GI_MEM) rw[0] = (uint32_t)-1;
    if (lane == 0) buf[dg.lmf] = '\0';
    finish(0u);
    return;
  }
  const GapOut& go = outs[i];
  const bool phase_d = (gi & GI_PHASE_D) != 0;
  rw[20] = (uint32_t)go.c_count;
  rw[21] = (uint32_t)go.n_len;
  rw[22] = (uint32_t)go.len[0];
  rw[23] = (uint32_t)go.len[1];
  rw[3] = (go.flags & (G2S_DEV_Q7_A | G2S_DEV_Q7_B | G2S_DEV_Q7_D)) ? G2S_GAP_Q7 : 0u;
  rw[0] = (uint32_t)gap_count(go, P, phase_d);
  if (phase_d && !P.skip_confident) {
    rw[8] = go.sub_vertices; rw[10] = go.sub_edges;    // vertices, edges (nothing contracted: no non-trivial component)
    rw[16] = go.sub_vertices; rw[18] = go.sub_edges;   // vertices_final, edges_final
  }
  if (!phase_d) {
    if (lane == 0) buf[dg.lmf] = '\0';
    finish(0u);
    return;
  }
  if (gi & GI_HOST) return;  // (g2s_d3_handoff gave it to the host, which writes its record and text)
  // ---- the closure into LDS
  const uint32_t nsegs = go.n_xl;
  const bool in_lds = nsegs <= P.seg_cap;
  const SegW* gsegs = (const SegW*)(sub_of(P, sub, i) + go.sub_off);
  uint32_t* cmap = lds + (size_t)P.seg_cap * 8u;  // by fill-buffer index: k-mer index | orientation << 30 | lower case << 31
  if (in_lds) {
    const uint4* src = (const uint4*)gsegs;
    uint4* dst = (uint4*)segs;
    for (uint32_t w = (uint32_t)lane; w < 2u * nsegs; w += 64u) dst[w] = src[w];
  }
  __syncthreads();
  auto seg_at = [&](uint32_t q) -> SegW { return in_lds ? segs[q] : gsegs[q]; };
  auto seg_uni = [&](uint32_t q) -> SegW {  // (the walk is the same in every lane: scalar registers)
    const SegW x = seg_at(q);
    SegW u;
    u.node = uni(x.node); u.depth_len = uni(x.depth_len); u.cnt = 0u; u.ts_tt = uni(x.ts_tt);
    u.par01 = uni(x.par01); u.par23 = uni(x.par23); u.flags = uni(x.flags); u.pad = uni(x.pad);
    return u;
  };
  // the branch rule's verdict for a k-mer that is not in the subgraph at this depth: at another depth, or the
  // sink's (Q5) — post.cpp: seg_safe
  const bool sink_safe = (go.dflags & G2S_DEVA_SINK_SAFE) != 0;
  auto outside_safe = [&](uint32_t x) -> bool {
    for (uint32_t q0 = 0; q0 < nsegs; q0 += 64u) {
      const uint32_t q = q0 + (uint32_t)lane;
      bool hit = false, verdict = false;
      if (q < nsegs) {
        const SegW o = seg_at(q);
        const int ts = (o.ts_tt & 0x7FFFu) == 0x7FFFu ? -1 : (int)(o.ts_tt & 0x7FFFu);
        const uint32_t oidx = o.node >> 1;
        const int tq = (o.node & 1u) ? (int)oidx - (int)x : (int)x - (int)oidx;
        hit = tq >= 0 && tq <= ts;
        verdict = tq <= (int)o.pad ? (o.ts_tt & 0x8000u) != 0 : (o.ts_tt & 0x80000000u) != 0;
      }
      const uint64_t hm = __ballot(hit);
      if (hm) return ((__ballot(hit && verdict) >> __builtin_ctzll(hm)) & 1ull) != 0;
    }
    return sink_safe;
  };
  // ---- the walk (wave-uniform).  It touches LDS only (and the rand() values at the choices): where every base of
  // the fill comes from — k-mer index, orientation, case — goes into cmap; the bases themselves are fetched
  // afterwards, all lanes at once and several loads in flight (a load and a dependent store per run of bases
  // inside the walk made the wave wait a memory round trip per segment).
  const uint32_t vr = W.vrank[i];
  const uint32_t off = W.base[i] + W.dvar[vr];
  const uint32_t* rd = rnd + off;
  const uint64_t avail = capacity > off ? capacity - off : 0ull;
  const int want = (int)W.dmin[i] + (GI_CLASS(gi) == 2u ? (int)W.tab[(uint64_t)W.var_toff[vr] + W.dvar[vr]] : 0);
  const int n_len = go.n_len, len0 = go.len[0], len1 = go.len[1];
  int draws = 1;
  const int pick = (n_len > 1 && avail > 0) ? (int)((uni(rd[0]) >> 1) % (uint32_t)n_len) : 0;  // :1440
  int d2 = pick ? len1 : len0;
  const int len = d2;
  int last_solid = d2;
  bool bad = false, ended = false;
  if ((uint32_t)len + 1u > P.map_cap) bad = true;
  {
    const uint32_t sg = (go.start_seg >> (16 * pick)) & 0xFFFFu;
    int si = sg == 0xFFFFu ? -1 : (int)sg;
    int t = (int)((go.start_t >> (16 * pick)) & 0xFFFFu);
    if (si < 0 || si >= (int)nsegs || avail < 1) bad = true;
    for (int guard = 0; !bad && d2 >= 0; guard++) {
      if (guard > 70000) { bad = true; break; }
      const SegW s = seg_uni((uint32_t)si);
      const int d0 = (int)(s.depth_len & 0xFFFFu);
      const uint32_t idx0 = s.node >> 1;
      const bool up = (s.node & 1u) == 0u;
      const uint32_t obit = up ? 0u : 0x40000000u;
      if (d0 + t != d2) { bad = true; break; }  // (state t of a segment that begins at depth d0 sits at depth d0 + t)
      if (t > 0) {
        // the states t, t-1, ..., 1 of this segment, in runs that share their safe bit (:1466-1468)
        const int ts = (s.ts_tt & 0x7FFFu) == 0x7FFFu ? -1 : (int)(s.ts_tt & 0x7FFFu);
        const int split = P.skip_confident ? t : (int)s.pad;
        const bool sfa = (s.ts_tt & 0x8000u) != 0, sfb = (s.ts_tt & 0x80000000u) != 0;
        int pos = t;
        while (pos > 0) {
          int lo_run;
          bool sf;
          if (P.skip_confident) { lo_run = 1; sf = true; }
          else if (pos > ts) { lo_run = pos; sf = outside_safe(up ? idx0 + (uint32_t)pos : idx0 - (uint32_t)pos); }
          else if (pos > split) { lo_run = max(1, split + 1); sf = sfb; }
          else { lo_run = 1; sf = sfa; }
          const int cnt = pos - lo_run + 1;
          const int qmax = sf ? -1 : min(pos, last_solid - k - d0);  // lower case unless within k of the last safe base
          for (int c = lane; c < cnt; c += 64) {
            const int q = lo_run + c;
            cmap[d0 + q - 1] = (up ? idx0 + (uint32_t)q : idx0 - (uint32_t)q) | obit | (q <= qmax ? 0x80000000u : 0u);
          }
          if (sf) last_solid = d0 + lo_run;
          pos = lo_run - 1;
        }
        draws += t;
        d2 -= t;
        t = 0;
      }
      if (s.flags & G2S_SUB_SOURCE) { left_fuz = (int)dg.lmf - d2; ended = true; break; }  // :1455-1462
      if (d2 > 0) {
        bool safe0;
        if (P.skip_confident) safe0 = true;
        else if ((s.ts_tt & 0x7FFFu) != 0x7FFFu) safe0 = (s.ts_tt & 0x8000u) != 0;  // state 0 of an S segment: safe bit a
        else safe0 = outside_safe(idx0);
        if (safe0) last_solid = d2;
        if (lane == 0) cmap[d2 - 1] = idx0 | obit | (d2 > last_solid - k ? 0u : 0x80000000u);
        const int nb = seg_nparents(s.par01, s.par23);
        if (nb == 0 || (nb > 1 && !(s.flags & G2S_SEG_ORDERED)) || (uint64_t)draws >= avail) { bad = true; break; }  // (:1493-1510: the host path's business)
        const uint32_t rv = nb > 1 ? uni(rd[draws]) >> 1 : 0u;  // (rand() % 1: the value does not matter)
        draws++;
        si = (int)seg_parent(s.par01, s.par23, nb == 1 ? 0 : (int)(rv % (uint32_t)nb));  // :1513
        if (si >= (int)nsegs) { bad = true; break; }
        t = (int)(uni(seg_at((uint32_t)si).depth_len) >> 16) - 1;  // a child in the closure puts the whole parent there
      }
      d2--;
    }
  }
  if (!ended || draws != want) bad = true;
  if (bad) { if (lane == 0) atomicAdd(&S->anomalies, 1u); }
  const int stop = (int)dg.lmf - left_fuz;  // the fill begins at this index of the buffer
  const uint32_t fill_len = (!bad && len >= stop && stop >= 0) ? (uint32_t)(len - stop) : 0u;
  if (!bad) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- the bases: four loads in flight per lane
    for (int p0 = stop; p0 < len; p0 += 256) {
      uint32_t e[4];
      char c[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int p = p0 + 64 * u + lane; e[u] = p < len ? cmap[p] : 0u; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t x = e[u] & 0x0FFFFFFFu;
        c[u] = (e[u] & 0x40000000u) ? chd[x] : chu[x];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int p = p0 + 64 * u + lane;
        if (p < len) buf[p] = (e[u] >> 31) ? (char)(c[u] | 0x20) : c[u];
      }
    }
    if (lane == 0) buf[len] = '\0';
  } else if (lane == 0) buf[dg.lmf] = '\0';
  rw[2] = (uint32_t)go.reached_j;  // :1171
  rw[3] |= G2S_GAP_PHASE_D;
  rw[7] = (uint32_t)draws;
  // (one counter for the whole list made 10 000 waves queue at one address of the L2: 64 counters, a cache line each)
  if (lane == 0 && fill_len) atomicAdd((unsigned long long*)&W.fill_bytes[(i & 63u) * 16u], (unsigned long long)fill_len);
  finish(fill_len);
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
  const size_t per = ((size_t)n + 64) * 4;
  return 12 * per + 64 * 128 + (size_t)G2S_D3_TABLE_BUDGET * 2 + (size_t)(G2S_D3_TABLE_BUDGET / 4u) * 4 + 4096;
}

void d3_work_carve(void* p, uint32_t n, D3Work* w) {
  char* c = (char*)p;
  const size_t per = ((size_t)n + 64) * 4;
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
  w->btab = (uint32_t*)c; c += (size_t)(G2S_D3_TABLE_BUDGET / 4u) * 4;
  w->tab = (uint16_t*)c;
}

hipError_t launch_rand_fill(hipStream_t st, uint32_t* rnd_all, const RandTables& rt, const D3Summary* sum_dev, uint64_t capacity) {
  const uint32_t rblocks = (uint32_t)std::min<uint64_t>((capacity + G2S_RAND_BLOCK - 1) / G2S_RAND_BLOCK, 4096);
  hipLaunchKernelGGL(g2s_rand_fill, dim3(std::max(1u, rblocks)), dim3(64), 0, st, rnd_all, rt, sum_dev, capacity);
  return hipGetLastError();
}

hipError_t launch_d3(hipStream_t st, const D3Params& P, const D3Work& W, const GapDev* gaps, const GapOut* outs,
                     const D3Gap* dgaps, const SubRec* sub, const char* lastch_up, const char* lastch_dn,
                     const RandTables& rt, uint32_t* rnd_all, uint64_t rnd_capacity, void* results, char* arena,
                     const D3Side& side, hipEvent_t handed_over) {
  if (P.n == 0) return hipSuccess;
  static_assert(sizeof(D3Summary) <= 1024, "summary slot");
  // (the summary and, behind it, the 64 fill-byte counters of the trace kernel)
  hipError_t e = hipMemsetAsync(W.sum, 0, 1024 + 64 * 128, st);
  if (e != hipSuccess) return e;
  const bool short_list = P.n <= 3072u;
  if (short_list) hipLaunchKernelGGL(g2s_d3_front, dim3(1), dim3(1024), 0, st, P, W, outs, dgaps);
  else {
    hipLaunchKernelGGL(g2s_d3_classify, dim3((P.n + 255u) / 256u), dim3(256), 0, st, P, W, outs, dgaps);
    hipLaunchKernelGGL(g2s_d3_scan, dim3(1), dim3(1024), 0, st, P, W, dgaps);
  }
  // (a tile of 256 deviations per workgroup, grid-stride: a short list has a few dozen tiles)
  const uint32_t tgrid = std::min(8192u, std::max(128u, P.n / 2u));
  hipLaunchKernelGGL(g2s_d3_tables, dim3(tgrid), dim3(256), 0, st, P, W, outs, sub, rnd_all + 31, rnd_capacity);
  if (short_list) hipLaunchKernelGGL(g2s_d3_back, dim3(1), dim3(1024), 0, st, P, W, outs, sub, rnd_all, rnd_capacity, side);
  else {
    hipLaunchKernelGGL(g2s_d3_blocks, dim3(256), dim3(256), 0, st, W);
    hipLaunchKernelGGL(g2s_d3_chain, dim3(1), dim3(1024), 0, st, W, rnd_all);
    hipLaunchKernelGGL(g2s_d3_handoff, dim3((P.n + 63u) / 64u), dim3(64), 0, st, P, W, outs, sub, rnd_all + 31, rnd_capacity, side);
  }
  if (handed_over) {
    e = hipEventRecord(handed_over, st);
    if (e != hipSuccess) return e;
  }
  const size_t lds = (size_t)P.seg_cap * sizeof(SegRec) + (size_t)P.map_cap * 4 + 16;
  e = hipFuncSetAttribute((const void*)g2s_d3_trace, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(g2s_d3_trace, dim3(P.n), dim3(64), lds, st, P, W, gaps, outs, dgaps, sub, lastch_up, lastch_dn,
                     rnd_all + 31, rnd_capacity, (g2s_result*)results, arena);
  return hipGetLastError();
}

}  // namespace g2s
