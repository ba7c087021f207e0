// gap2seq_amd/csrc/post.hpp — host part of fill_gap's phase D, driven by the
// packed backward closure (SubState array) that the g2s_extract kernel produced
// for one gap:
//   D2 SCC contraction + the "vertex on all paths" rule   /root/reference/src/Gap2Seq.cpp:1314-1435
//   D3 random traceback                                                               :1437-1522
// (D1, the subgraph extraction :1169-1312, runs on the GPU.)
// Everything except the assignment of rand() stream offsets is independent per
// gap and runs on a thread pool; offsets are assigned in gap order by a pass that
// costs O(1) per gap whenever the number of draws does not depend on the choices.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/g2s.h"
#include "dbg.hpp"
#include "fill_device.h"
#include "glibc_rand.hpp"

namespace g2s {

// One gap as the host keeps it.
struct GapJob {
  int g = 0, lmf = 0, rmf = 0;
  int skip_if_prev_right_fuz_gt = -1;
  bool bad_flank = false;
  uint32_t text_off = 0;  // (batches) where the gap's flank text starts in the batch's pinned text (flank look-ups on the device)
  // oriented node (or kInvalidNode) of: left.substr(d,k) d=0..lmf | right-BFS
  // seeds right.substr(len-k-j,k) j=0..rmf | targets right.substr(j,k) j=0..rmf.
  // Not owned: the batch's pinned buffer (written by the flank look-up kernel, flank_lookup.hip)
  // or, in the test hooks, the caller's vector (resolve_on_host).
  // (No member with a destructor: a batch holds one GapJob per gap, built and dropped with every list.)
  const uint32_t* nodes = nullptr;
  const uint32_t* lseeds() const { return nodes; }
  const uint32_t* rseeds() const { return nodes + (lmf + 1); }
  const uint32_t* targets() const { return nodes + (lmf + 1) + (rmf + 1); }
  // (hooks) resolve the flank k-mers of left / right on the host, into *store (which must outlive the job)
  void resolve_on_host(const Graph& gr, const char* left, size_t left_len, const char* right, size_t right_len,
                       std::vector<uint32_t>* store) {
    const int k = gr.k;
    (void)left_len;
    store->clear();
    for (int d = 0; d <= lmf; d++) store->push_back(gr.node_of(left + d));                        // :995,1083
    for (int d = 0; d <= rmf; d++) store->push_back(gr.node_of(right + (right_len - k - d)));     // :878,954
    for (int d = 0; d <= rmf; d++) store->push_back(gr.node_of(right + d));                       // :1113
    nodes = store->data();
  }
  size_t buf_bytes(int k, int d_err) const { return (size_t)(g + k + d_err + lmf + rmf + 1 + 2); }
};

struct FillParams {
  int k = 31, d_err = 500;
  bool skip_confident = false, all_paths = true, unique_paths = false;
};

// Device results of one gap as seen on the host.
struct SubView {
  const GapOut* out = nullptr;
  const SubRec* st = nullptr;
  uint32_t n = 0;
  const uint64_t* xp = nullptr;  // parents beyond the first: (state << 32 | parent), sorted by state
  uint32_t n_xp = 0;
  // segment tier: the closure as it came from the device (SegRec); st / xp are filled by seg_expand
  const SegRec* segs = nullptr;
  uint32_t n_segs = 0;
};
// Closure segments (segment tier) -> per-state records + side list, in the order the other tiers
// emit them: children before parents, inside a segment from its last closure state down to its
// entry.  out needs go.n_sub records, xp go.n_xp entries.
void seg_expand(const FillParams& p, const GapJob& job, const GapOut& go, const SegRec* segs, uint32_t n_segs,
                SubRec* out, uint64_t* xp);
// ts / tt of a closure segment: 15 bits each, 0x7FFF = none (bits 15 / 31 are the device's safe bits)
inline int seg_ts(const SegRec& s) { return (s.ts_tt & 0x7FFFu) == 0x7FFFu ? -1 : (int)(s.ts_tt & 0x7FFFu); }
inline int seg_tt(const SegRec& s) { return ((s.ts_tt >> 16) & 0x7FFFu) == 0x7FFFu ? -1 : (int)((s.ts_tt >> 16) & 0x7FFFu); }
inline uint32_t sub_depth(const SubRec& s) { return s.meta & G2S_SUB_META_DEPTH_MASK; }
inline uint32_t sub_flags(const SubRec& s) { return s.meta >> G2S_SUB_META_FLAG_SHIFT; }
// the parents of state i (a set, at most 4); returns how many
inline int sub_preds(const SubView& v, uint32_t i, int32_t out[4]) {
  const int32_t p = v.st[i].pred;
  if (p < 0) return 0;
  out[0] = p & G2S_SUB_PRED_MASK;
  int n = 1;
  if (p & G2S_SUB_MORE) {
    uint32_t lo = 0, hi = v.n_xp;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if ((uint32_t)(v.xp[mid] >> 32) < i) lo = mid + 1; else hi = mid;
    }
    for (; lo < v.n_xp && (uint32_t)(v.xp[lo] >> 32) == i && n < 4; lo++) out[n++] = (int32_t)(uint32_t)v.xp[lo];
  }
  return n;
}
// HBM-tier closure (SubState, parents by GATB slot) -> SubRec + side list, appended to the vectors
void sub_convert(const SubState* in, uint32_t n, std::vector<SubRec>* recs, std::vector<uint64_t>* xp);

// Per-segment results of seg_analyze (segment tier, closures without a repeated k-mer).
struct SegInfo {
  int32_t lo, hi;      // depths at which tracebacks through this segment's entry stop (-1 / 1<<30: not fixed)
  int16_t split;       // S part: states t <= split carry safe_a, the others safe_b (a sink inside the segment)
  uint8_t safe_a, safe_b;
  uint8_t npar;
};

// A stretch of consecutive k-mer indices of the S closure whose vertices share the verdict of the branch rule
// (seg_analyze on closures in which a k-mer occurs at several depths): sorted by first index, disjoint.
struct SegRun {
  uint32_t lo, hi;
  uint8_t safe;
};

struct SubPrep {
  bool seg_mode = false;  // analysed on the closure SEGMENTS (seg_analyze); st / safe are not used then
  SegInfo* seg = nullptr;                                // per closure segment
  std::pair<uint32_t, uint32_t>* s_iv = nullptr;         // S closure as k-mer index intervals (lo, segment), sorted
  uint32_t n_iv = 0;
  std::vector<uint64_t> own_seg;                         // backing store of the two when the caller gives no scratch
  bool sink_safe = false, has_choice = false;
  // a k-mer at several depths of the closure: the vertices of :1314-1435 are k-mers, not states, and the safe
  // bit of a state is that of its k-mer's run (run_mode; seg[].split / safe_a / safe_b are unused then)
  bool run_mode = false;
  std::vector<SegRun> runs;
  // closures analysed on the device whose draw count depends on the draws: (lowest, highest) stop depth
  // behind every segment's entry (seg_stop_depths), so that the draw-count walk ends where the two meet
  const int32_t* stop = nullptr;
  int start_seg[2] = {-1, -1}, start_t[2] = {0, 0};
  bool phase_d = false;   // count > 0 && pathLengths non-empty (:1169)
  int count = 0;          // value fill_gap returns (before a backtrace failure / memory verdict)
  uint32_t flags = 0;     // G2S_GAP_* bits found on the host
  uint64_t sub[6] = {0, 0, 0, 0, 0, 0};
  std::vector<uint8_t> safe;  // per state: branch[vertex of its k-mer] == 1 (Q5 default = sink)
  std::vector<SubRec> own;    // segment tier: this gap's expanded closure when the launch's shared buffer is full
  int start_idx[2] = {-1, -1};  // state index of (reachedTarget, pathLengths[i])
  int stop_depth[2] = {-1, -1}; // depth every traceback from start i stops at, or -1 when it depends on the draws
  // back to the state of a new object, keeping the vectors' storage (the per-gap array of a batch is recycled)
  void reset() {
    seg_mode = false; seg = nullptr; s_iv = nullptr; n_iv = 0; own_seg.clear();
    sink_safe = has_choice = run_mode = false; runs.clear(); stop = nullptr;
    start_seg[0] = start_seg[1] = -1; start_t[0] = start_t[1] = 0;
    phase_d = false; count = 0; flags = 0;
    for (int q = 0; q < 6; q++) sub[q] = 0;
    safe.clear(); own.clear();
    start_idx[0] = start_idx[1] = -1; stop_depth[0] = stop_depth[1] = -1;
  }
};

// SCC / branch rule / stop-depth analysis of one gap; thread safe.
void sub_analyze(const FillParams& p, const GapJob& job, const SubView& v, SubPrep* out);
// Number of rand() draws the traceback will consume when pathLengths[pick] is chosen,
// or -1 when that depends on later draws.
inline int sub_fixed_draws(const SubView& v, const SubPrep& prep, int pick) {
  return prep.stop_depth[pick] < 0 ? -1 : 1 + (v.out->len[pick] - prep.stop_depth[pick]);
}
// The same analysis on the closure SEGMENTS of the segment tier.  When no k-mer occurs at two depths of the S
// closure the subgraph is a DAG whose vertices are the states: nothing to contract, :1314-1435 reduces to the
// branch rule, which is constant along a segment: O(segments).  Otherwise the vertices are k-mers: the closure's
// index intervals are cut at every segment end and every edge that is not unitig-internal, which leaves RUNS of
// k-mers that are chains; strong components, contraction and the branch rule then work on runs, not on states
// (a deep gap of config 5: ~10 k runs for ~150 k states).  Always returns true (the closure has been analysed);
// G2S_STATE_D2=1 makes it return false where a k-mer repeats, and the caller then expands the segments
// (seg_expand) and takes sub_analyze.  seg_traceback / seg_count_draws are sub_traceback / sub_count_draws on
// segments.
// `scratch`: 24 bytes per segment, 8-byte aligned, alive as long as *out is used (nullptr: *out allocates).
bool seg_analyze(const FillParams& p, const GapJob& job, const SubView& v, SubPrep* out, void* scratch = nullptr);
// (packed12: `rands` holds (value % 12) in four bits per draw, eight draws a word — all a traceback asks of a value is
// its remainder by the number of lengths (<= 2) or of parents (<= 4): what the device hands the host for the gaps it
// finishes, an eighth of the words over the link)
void seg_traceback(const Graph& g, const FillParams& p, const GapJob& job, const SubView& v, const SubPrep& prep,
                   const uint32_t* rands, char* buf, g2s_result* res, bool packed12 = false);
int seg_count_draws(const Graph& g, const SubView& v, const SubPrep& prep, const uint32_t* rands);
// (lowest, highest) depth at which a traceback that passes through the entry of each closure segment stops
// (:1455-1462), (-1, 1 << 30) where that is not fixed by the subgraph; out: two values per segment
void seg_stop_depths(const SubView& v, int32_t* out);

// D3.  `rands` points at the raw word of this gap's first draw (rand() value = word >> 1);
// returns through res (count, fuz, draws, flags).
// Writes the reference's `fill` buffer into buf (size job.buf_bytes).
void sub_traceback(const Graph& g, const FillParams& p, const GapJob& job, const SubView& v, const SubPrep& prep,
                   const uint32_t* rands, char* buf, g2s_result* res);

// TEST HOOK support: the closure the g2s_extract kernel computes, derived on the host
// from a full DP table (states sorted per level).  Not used by the product path.
int sub_count_draws(const Graph& g, const SubView& v, const SubPrep& prep, const uint32_t* rands);

struct HostTable {
  std::vector<uint32_t> lvl;     // D+2 offsets
  std::vector<uint64_t> states;  // (node << 32 | count), sorted inside each level
  int D = 0;
  uint32_t find(int depth, uint32_t node) const;
};
void host_closure(const Graph& g, const FillParams& p, const GapJob& job, const HostTable& t, const GapOut& go,
                  std::vector<SubState>* out, uint32_t* q7);

// (G2S_DEBUG) time stamps of the last seg_analyze_runs on this thread, microseconds
extern thread_local double g2s_post_laps[12];

}  // namespace g2s
