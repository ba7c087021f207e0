// gap2seq_amd/csrc/seg_device.h — device helpers shared by the kernels of the segment tier (fill_seg.hip: one or two
// waves per gap; fill_segw.hip: the large variant, a workgroup of several waves per gap): wave reductions and scans
// inside the vector ALU, segment arithmetic, the bitonic sort and interval merge in LDS, the launch arguments.
#pragma once
#include "sync_debug.h"
#include <hip/hip_runtime.h>

#include "fill_device.h"
#include "fill_seg.h"
#include "flank_lookup.h"

// The HIP headers' __ballot(p) compares an int with zero: a predicate is first turned into 0 / 1 in a vector register
// and then compared again — three instructions where the compare that made the predicate had already left the mask in
// a scalar register pair.  The segment kernels hold hundreds of ballots in their inner loops.
__device__ __forceinline__ unsigned long long g2s_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
#define __ballot(p) g2s_ballot(p)
// the lanes in which every one of the conditions holds: a ballot of each (one comparison = one instruction that leaves
// the mask in scalar registers), combined as scalars — a ballot of the conjunction goes through a vector register
template <class... B>
__device__ __forceinline__ unsigned long long ballot_and(bool a, B... b) { return (g2s_ballot(a) & ... & g2s_ballot(b)); }

#define SEG_INF 0x7FFFFFFFu
// -DG2S_SEG_PROFILE: cycles of the sections of a phase B round, summed per gap into the last words of the
// gap's diagnostics row (G2S_SEG_DUMP; tools only)
#ifdef G2S_SEG_PROFILE
#define SEG_PROF_T(i) prof_t[i] = __builtin_amdgcn_s_memtime()
#define SEG_PROF_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define SEG_PROF_ACC() do { for (int pi = 0; pi < 4; pi++) prof_acc[pi] += (uint32_t)(prof_t[pi + 1] - prof_t[pi]); } while (0)
#define SEG_PROF_TAIL(i) prof_tail[i] = __builtin_amdgcn_s_memtime()
#else
#define SEG_PROF_TAIL(i) do {} while (0)
#define SEG_PROF_T(i) do {} while (0)
#define SEG_PROF_WAIT() do {} while (0)
#define SEG_PROF_ACC() do {} while (0)
#endif
#define SEG_NOPAR 0xFFFFu

namespace {

#ifdef G2S_SYNC_DEBUG  /* sync_debug.h: the race-hunting builds */
__device__ __forceinline__ void lds_sync_at(uint32_t site) {
  g2s_sync_jitter(site);
#ifdef G2S_PARANOID_SYNC
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}
#define lds_sync() lds_sync_at((uint32_t)__LINE__)
#else
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif
__device__ __forceinline__ uint32_t rl(uint32_t x, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, l); }
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint64_t below(int lane) { return (1ull << lane) - 1ull; }
// state t of the segment that starts at v0
__device__ __forceinline__ uint32_t seg_node(uint32_t v0, uint32_t t) { return (v0 & 1u) ? v0 - 2u * t : v0 + 2u * t; }
// t with seg_node(v0, t) == node and t < len, else -1
__device__ __forceinline__ int seg_pos(uint32_t v0, uint32_t len, uint32_t node) {
  if (node == G2S_DEV_INVALID || ((node ^ v0) & 1u)) return -1;
  const int dt = (int)(node - v0) >> 1;  // node ids are below 2^31
  const int t = (v0 & 1u) ? -dt : dt;
  return (t >= 0 && (uint32_t)t < len) ? t : -1;
}
// ts / tt of a segment are kept in 15 bits each (0x7FFF = none); bits 15 and 31 carry the safe bits
__device__ __forceinline__ int dec15(uint32_t x) { return (x & 0x7FFFu) == 0x7FFFu ? -1 : (int)(x & 0x7FFFu); }
__device__ __forceinline__ uint32_t enc15(int t) { return t < 0 ? 0x7FFFu : (uint32_t)t; }
// sum over the lanes, wave-uniform (row shifts and row broadcasts inside the vector ALU, see wave_min)
__device__ __forceinline__ uint32_t wave_sum(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);  // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);  // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);  // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);  // row_shr:8: lane 15 of a row = its sum
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
// minimum over the lanes, wave-uniform: row shifts and row broadcasts inside the vector ALU (a shuffle
// through the LDS crossbar per step costs ten times as much)
__device__ __forceinline__ uint32_t wave_min(uint32_t x) {
  const int inf = (int)0xFFFFFFFFu;
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(inf, (int)x, 0x111, 0xF, 0xF, false));  // row_shr:1
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(inf, (int)x, 0x112, 0xF, 0xF, false));  // row_shr:2
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(inf, (int)x, 0x114, 0xF, 0xF, false));  // row_shr:4
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(inf, (int)x, 0x118, 0xF, 0xF, false));  // row_shr:8
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(inf, (int)x, 0x142, 0xA, 0xF, false));  // row_bcast:15
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(inf, (int)x, 0x143, 0xC, 0xF, false));  // row_bcast:31
  return rl(x, 63);
}
// inclusive prefix sum over the lanes (the same six steps: a scan inside every row of 16, then the row totals)
__device__ __forceinline__ uint32_t wave_scan(uint32_t x, int lane) {
  (void)lane;
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
  return x;
}

// inclusive prefix maximum over the lanes (unsigned values; the same six steps)
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t x) {
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false));
  return x;
}
// the value of the lane below (lane 0: zero) — wave_shr:1
__device__ __forceinline__ uint32_t wave_shr1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xF, 0xF, false); }

// ascending bitonic sort of n2 (a power of two >= 2) 64-bit keys in LDS by one wave
__device__ __forceinline__ void lds_sort64(uint64_t* a, uint32_t n2, int lane) {
  for (uint32_t k = 2; k <= n2; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = (uint32_t)lane; t < (n2 >> 1); t += 64u) {
        const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));  // the t-th index with bit j clear
        const uint32_t l = i | j;
        const bool up = (i & k) == 0u;
        const uint64_t p = a[i], q = a[l];
        if ((p > q) == up) { a[i] = q; a[l] = p; }
      }
      lds_sync();
    }
}
// a[0 .. n): keys (first index << 32 | orientation << 31 | last index), sorted.  Merges overlapping and
// adjacent intervals in place: a[g] = first << 32 | last for g < M (returned), sorted and disjoint with a
// hole between any two.  *cross: an interval of each orientation overlap; *overlap (optional): any two overlap.
__device__ __forceinline__ uint32_t lds_merge_intervals(uint64_t* a, uint32_t n, int lane, bool* cross, bool* overlap = nullptr) {
  uint32_t* w = (uint32_t*)a;
  uint32_t M = 0, pm = 0, pm_e = 0, pm_o = 0;  // highest (last index + 1) so far: all, even, odd entries
  bool cr = false, ov = false;
  for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
    const uint32_t i = i0 + (uint32_t)lane;
    const bool h = i < n;
    const uint64_t key = h ? a[i] : 0ull;
    const uint32_t lo = (uint32_t)(key >> 32), hi = (uint32_t)key & 0x7FFFFFFFu, odd = ((uint32_t)key >> 31) & 1u;
    // inclusive prefix maxima (inside the vector ALU: a shuffle through the LDS crossbar per step and value was most
    // of this function), then the value of the lane below
    const uint32_t sa = wave_scan_max(h ? hi + 1u : 0u), se = wave_scan_max((h && !odd) ? hi + 1u : 0u),
                   so = wave_scan_max((h && odd) ? hi + 1u : 0u);
    uint32_t xa = wave_shr1(sa), xe = wave_shr1(se), xo = wave_shr1(so);
    xa = max(xa, pm); xe = max(xe, pm_e); xo = max(xo, pm_o);   // over everything before element i
    if (__ballot(h && (odd ? xe : xo) > lo)) cr = true;          // an earlier interval of the other orientation ends at or after lo
    if (__ballot(h && xa > lo)) ov = true;                       // an earlier interval ends at or after lo
    const bool start = h && (xa == 0u || lo > xa);               // a hole in front of this element
    const uint64_t sm = __ballot(start);
    const uint32_t g = M + (uint32_t)__popcll(sm & below(lane));
    lds_sync();  // (every lane has read its element: the writes below land at or in front of this chunk)
    if (start) {
      w[2u * g + 1u] = lo;
      if (g > 0u) w[2u * (g - 1u)] = xa - 1u;
    }
    M += (uint32_t)__popcll(sm);
    pm = max(pm, rl(sa, 63)); pm_e = max(pm_e, rl(se, 63)); pm_o = max(pm_o, rl(so, 63));
    lds_sync();
  }
  if (M > 0u && lane == 0) w[2u * (M - 1u)] = pm - 1u;
  lds_sync();
  *cross = cr;
  if (overlap) *overlap = ov;
  return M;
}

}  // namespace

// What one launch works with (both kernels).
struct SegArgs {
  const uint32_t* succ;
  const uint32_t* urec;
  GapSrc gaps;
  const uint32_t* gap_ids;
  const uint32_t* flank_nodes;
  SubRec* sub_out;
  unsigned long long out_cap;
  unsigned long long* out_counter;
  GapOut* outs;
  GapOut* outs_host;
  uint32_t* done_list;
  int skip_confident;
  uint32_t* dbg;
  uint32_t dbg_words;
  // batched announcements (g2s_fill_seg / g2s_fill_seg2; see `publish`): per XCD a ticket counter and a list of the
  // gaps finished there; pub_batch 1 = every gap announces itself
  unsigned long long* xcd_tickets;  // 8 counters, zero before the launch
  uint32_t* xcd_list;               // 8 lists of xcd_stride entries, 0xFFFFFFFF before the launch
  uint32_t xcd_stride;
  uint32_t pub_batch;               // a power of two <= 64
  // the results stay on the device (sub_out and outs are device memory, phase D3 follows on the stream:
  // d3_device.hip): nothing is announced to the host, no write-back of the L2 per gap
  uint32_t resident;
  uint32_t d2_ticks;  // (tools, G2S_D2_LOG) where in d2_list the gaps' listing times go, by gap; 0: nowhere
  // resident mode: a gap that outgrows the regular tier's capacities enters itself here (count: out_counter[1]); the
  // large variant follows on the stream and takes the list (fill_segw.hip) — the gap's results stay on the device too
  uint32_t* ovf_list;
  // resident mode, deep lists (g2s_fill_segw): a closure the host will analyse (more than 192 segments, or a k-mer at
  // two depths) also goes to pinned host memory the moment its gap ends — the host analyses it under the rest of the
  // launch instead of behind phase D3's hand-over.  early_items: eight words per item {gap, segments, offset (two
  // words), ready, -, -, -}, zero before the launch; early_ctr (device memory, zero before the launch): items and
  // segments handed out.  All null: no early hand-over.
  SegRec* early_segs;
  uint32_t* early_items;
  GapOut* early_outs;
  unsigned long long* early_ctr;
  uint32_t early_cap_items, early_cap_segs;
  // resident mode: a gap whose closure this kernel leaves unanalysed (more than 192 segments, a k-mer at several
  // depths) enters itself here, counted in out_counter[4]: g2s_d2_* (d2_device.hip) follows on the stream.  Null: such
  // gaps are the host's (post.cpp).
  uint32_t* d2_list;
  // ... as gap | d2_tag (0x80 | the list's number mod 128, in bits 24-31): g2s_d2_small may be polling the list while
  // this kernel runs, and takes an entry only when it carries this list's tag (the gap's record and closure are
  // written, and fenced, in front of it).  out_counter[32 + (gap & 63)] counts the gaps that are through.
  uint32_t d2_tag;
  // resident mode, g2s_fill_seg / g2s_fill_seg2: every gap's wave(s) resolve the gap's flank k-mers themselves
  // (flank_device.h) — no look-up kernel in front.  inl_text: the list's flank text (GapDev.rs_mask = the gap's offset
  // in it: [left k+lmf][right first k+rmf][right last k+rmf], 4-byte aligned), null: the ids are in flank_nodes.  The
  // ids are also stored at inl_nodes_dev / inl_nodes_host + GapDev.flank_off (the table the later kernels and the
  // host's half read).
  const char* inl_text;
  uint32_t* inl_nodes_dev;
  uint32_t* inl_nodes_host;
  // (not 0: gap i has its text at inl_text + i * inl_stride — a list without a bad flank; a multiple of 4, at most 512)
  uint32_t inl_stride;
  g2s::FlankLookup lk;
  // resident mode, g2s_fill_seg / g2s_fill_seg2: a gap whose traceback has no choice to make (one path length, no entry
  // of the traceback closure with several parents) is traced by its own wave — fill text and g2s_result record written
  // where phase D3's trace kernel would write them, GapOut.dflags |= G2S_DEVA_TRACED, GapOut.top_level = the fill's
  // length (fill_seg.hip).  tr_results: g2s_result[n] as 28 words each, device-writable; null: every gap is the trace
  // kernel's.  GapDev.rlog_off = the gap's offset in the arena behind tr_arena_base; GapDev.rlog_cap != 0: the gap
  // carries a skip rule (its result depends on the gap in front: not traced here).
  uint32_t* tr_results;
  char* tr_arena;
  unsigned long long tr_arena_base;
  const char* tr_chu;   // last base of every k-mer by index, walked upwards / downwards (the trace kernel's tables)
  const char* tr_chd;
  unsigned long long tr_max_states;  // -max-mem / 64: a gap beyond it gets the memory verdict (phase D3's)
  int tr_k;
  // ... and a traceback WITH choices is written as a guess (first length, first parent), text and record also at the same
  // offsets of these device buffers: GapOut.dflags |= G2S_DEVA_SPEC; the trace kernel sends what differs.  Null: no guesses.
  char* tr_spec_text;
  uint32_t* tr_spec_res;
  // (two waves per gap — short lists, whose launch ends with its slowest gap) not 0: only the first tr_guess_until gaps
  // to end their search write a guess; the gaps the launch waits for do not add their guess's cycles to it
  uint32_t tr_guess_until;
};

// LDS of the large variant (words): see the layout notes at each phase
#define SEGX_LDS_WORDS 39936u
// global scratch of one workgroup of the large variant (words): six segment arrays + two queues
#define SEGX_SCR_WORDS (6u * G2S_SEGX_CAP + 2u * G2S_SEGX_QCAP)
#define SEGX_EMPTY64 0xFFFFFFFFFFFFFFFFull
#define SEGX_TOMB64 0xFFFFFFFFFFFFFFFEull
