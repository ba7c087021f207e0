// gap2seq_amd/csrc/d2_device.hip — phase D2 of fill_gap (/root/reference/src/Gap2Seq.cpp:1314-1435: strong
// components, contraction, topological sweep with the branch counter, branch[v] == 1) on the device, for the closures
// the fill kernels leave unanalysed: more than 192 segments, or a k-mer at several depths.  One workgroup per gap — one
// wave for closures of up to 256 segments (g2s_d2_small), sixteen for the rest (g2s_d2_big) —, the gap's graph in LDS;
// post.cpp (seg_analyze, seg_analyze_runs) is the host version and the reference for every step.
//
// A closure arrives as SegRec records (fill_device.h): unitig segments whose states t <= ts lie on a path to a sink.
//  * Every k-mer at one depth (the segments' index intervals are disjoint: a sort tells): the subgraph is a DAG whose
//    vertices are the states, and the branch rule is a prefix sum over the segments in creation order (fill_seg.hip
//    does the same for closures of up to 192 segments, with a pairwise test in place of the sort).
//  * Otherwise the reference's vertices are K-MERS (node2boost) and its edges the de-duplicated state transitions.
//    Inside a unitig those are index +-1 steps: cut the closure's index intervals at every segment end and at every
//    k-mer that carries another edge (a parent's last k-mer, an entry, a sink position); between two cuts lies a RUN
//    of k-mers whose only edges are the chain's own.  The run graph (a few hundred to a few thousand nodes) goes
//    into LDS as two adjacency tables.  Runs with one edge in and one out are contracted out of it.  Strong
//    components: nodes without a live edge in or out are peeled off (each its own component; a lane follows a chain
//    of freed nodes itself), what is left is cycles and the paths between them — a forward and a backward search
//    from a pivot give one component, and the peeling goes on.  Then the components in a topological order (Kahn;
//    any topological order gives the same verdicts), the branch counter as a prefix sum over that order,
//    and the verdict of every run.
// What leaves: per gap the subgraph statistics and a sorted list of runs {first index, last index | safe << 31};
// g2s_d3_trace looks the safe bit of a traced base up there (a k-mer in no run reads branch[sink], Q5).
// A closure beyond an instantiation's capacities is passed on (small -> large) or left to the host (post.cpp).
//
// Several waves: the passes over the segments, the sorts, the runs, the edges, the adjacency, the chains, the peeling
// and the searches are shared by all threads (positions by atomic counters where the order does not matter, two-level
// prefix sums where it does); merging sorted intervals and the scans over per-chunk totals stay on the first wave.
#include "sync_debug.h"
#include <hip/hip_runtime.h>

#include "d2_device.h"
#include "seg_device.h"

namespace {

using g2s::D2Args;
using g2s::D2Out;

template <uint32_t NT_, uint32_t NS_, uint32_t NREC_, uint32_t BP_, uint32_t NV_, uint32_t E_>
struct D2Caps {
  static constexpr uint32_t NT = NT_, NS = NS_, NREC = NREC_, BP = BP_, NV = NV_, E = E_;
  // LDS, bytes: the sort buffer (and, DAG closures, the per-segment words behind its first NS keys) ...
  static constexpr uint32_t K_BYTES = BP * 8u;
  // ... or the graph: offsets and adjacency in both directions (u16), live degrees (u16, packed), component / state
  // (u32), two work lists (u16)
  static constexpr uint32_t OFF_BYTES = ((NV + 2u) * 2u + 15u) & ~15u;
  static constexpr uint32_t G_BYTES = 2u * OFF_BYTES + 2u * E * 2u + 2u * NV * 2u + NV * 4u + 2u * NV * 2u;
  static constexpr uint32_t MAIN_BYTES = K_BYTES > G_BYTES ? K_BYTES : G_BYTES;
  // ... and behind either: per 64 cuts a mask and a count of the runs between cuts (or: per 64 segments a total), two
  // bits per node (pass-through; sink position), counters
  static constexpr uint32_t CHUNKS = (BP > NREC ? BP : NREC) / 64u;
  static constexpr uint32_t HDR_WORDS = 64u;
  static constexpr uint32_t AUX_BYTES = CHUNKS * 12u + 2u * (NV / 32u) * 4u + HDR_WORDS * 4u;
  static constexpr uint32_t LDS_BYTES = MAIN_BYTES + AUX_BYTES;
  // global scratch of a workgroup, words
  static constexpr uint32_t SCR_WORDS = 7u * NS + 3u * BP + 5u * NV + E + 64u;
  static_assert(BP >= 4u * NS, "the sort buffer holds the look-up tables: cuts, vertex intervals, chain intervals");
  static_assert(2u * NS + NREC <= K_BYTES / 4u, "the per-segment words fit behind the keys");
  static_assert(E >= 2u * NV, "the reverse adjacency's space holds two packed degree tables");
};
using D2Small = D2Caps<64u, G2S_D2_SMALL_NS, G2S_D2_SMALL_NREC, G2S_D2_SMALL_BP, G2S_D2_SMALL_NV, G2S_D2_SMALL_E>;
// (round 6) the small capacities on FOUR waves: a list's launch of g2s_d2_small ends with its largest closure — 240 us on
// one wave for config 3's, longer than phase D3's front kernels and half the trace kernel that hide the launch on a
// list of 5 000 gaps — and the passes over segments, cuts, edges and nodes are strided over the workgroup's waves
using D2Small4 = D2Caps<256u, G2S_D2_SMALL_NS, G2S_D2_SMALL_NREC, G2S_D2_SMALL_BP, G2S_D2_SMALL_NV, G2S_D2_SMALL_E>;
using D2Big = D2Caps<G2S_D2_BIG_NT, G2S_D2_BIG_NS, G2S_D2_BIG_NREC, G2S_D2_BIG_BP, G2S_D2_BIG_NV, G2S_D2_BIG_E>;

// state of a node while the strong components are found; afterwards: its component (the id of one of its nodes)
#define D2_ALIVE 0xFFFFFFFFu
#define D2_FW 0xFFFFFFFDu
#define D2_FWBW 0xFFFFFFFCu
// a pass-through node (one edge in, one out, not covered in both directions): 0x80000000 | edge slot << 16 | the node
// its chain leaves, D2_PASS until the chain has been walked
#define D2_PASS 0xFFFFFF00u
__device__ __forceinline__ bool d2_is_pass(uint32_t st) { return st >= 0x80000000u && st <= D2_PASS; }
#define D2_NONE 0xFFFFFFFFu

// header words (LDS)
enum {
  H_NEXT = 0, H_ORDER, H_BIGS, H_N0, H_N1, H_N2, H_FLAG, H_PIVOT, H_R0, H_R1, H_R2, H_R3, H_M0, H_M1, H_M2, H_BASE_LO, H_BASE_HI,
  H_A64 = 18 /* three 64-bit accumulators: words 18-23 */, H_BIGS_A = 24 /* 12 u16 */, H_BIGS_B = 30 /* 12 u16 */, H_GAP = 36, H_SLOT = 37
};

// all threads of the workgroup (one wave: nothing but the LDS queue to wait for)
template <uint32_t NT>
__device__ __forceinline__ void bsync() {
  if constexpr (NT == 64u) lds_sync(); else __syncthreads();
}
// what the workgroup wrote to global scratch is read back by other threads of it
template <uint32_t NT>
__device__ __forceinline__ void gsync() {
  __threadfence_block();
  if constexpr (NT == 64u) __builtin_amdgcn_wave_barrier(); else __syncthreads();
}
// 16-bit counters, two a word: atomics on the word (a counter never leaves 0 .. 65535)
__device__ __forceinline__ uint32_t pk_get(const uint32_t* a, uint32_t i) { return (a[i >> 1] >> (16u * (i & 1u))) & 0xFFFFu; }
__device__ __forceinline__ uint32_t pk_add(uint32_t* a, uint32_t i, uint32_t d) {
  const uint32_t sh = 16u * (i & 1u);
  return (atomicAdd(&a[i >> 1], d << sh) >> sh) & 0xFFFFu;
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t* a, uint32_t i, uint32_t d) {
  const uint32_t sh = 16u * (i & 1u);
  return (atomicSub(&a[i >> 1], d << sh) >> sh) & 0xFFFFu;
}
__device__ __forceinline__ uint32_t pow2_at_least(uint32_t n) {
  uint32_t p = 2u;
  while (p < n) p <<= 1;
  return p;
}
// intervals {first, last} sorted by first, disjoint: is x in one of them?  *hi_out: the last index of that one
__device__ __forceinline__ bool iv_has(const uint32_t* iv, uint32_t m, uint32_t x, uint32_t* hi_out = nullptr) {
  uint32_t lo = 0, hi = m;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (iv[2u * mid] <= x) lo = mid + 1u; else hi = mid; }
  if (lo == 0u) return false;
  const uint32_t h = iv[2u * (lo - 1u) + 1u];
  if (hi_out) *hi_out = h;
  return x <= h;
}
// position of x in a sorted list of distinct values (x is in it)
__device__ __forceinline__ uint32_t cut_index(const uint32_t* cut, uint32_t n, uint32_t x) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cut[mid] < x) lo = mid + 1u; else hi = mid; }
  return lo < n ? lo : n - 1u;
}
// places behind *ctr for the lanes of this wave that ask (m: their ballot): the lane's own place
__device__ __forceinline__ uint32_t wave_reserve(uint32_t* ctr, uint64_t m, int lane) {
  uint32_t base = 0;
  if (lane == 0 && m) base = atomicAdd(ctr, (uint32_t)__popcll(m));
  return rl(base, 0) + (uint32_t)__popcll(m & below(lane));
}

struct SegD {  // a closure segment as this kernel reads it
  uint32_t node, dl, ts_tt, p01, p23, flags;
  __device__ __forceinline__ int ts() const { return dec15(ts_tt); }
  __device__ __forceinline__ int len() const { return (int)(dl >> 16); }
  __device__ __forceinline__ int d0() const { return (int)(dl & 0xFFFFu); }
  __device__ __forceinline__ bool up() const { return (node & 1u) == 0u; }
  __device__ __forceinline__ bool source() const { return (flags & G2S_SUB_SOURCE) != 0u; }
  __device__ __forceinline__ uint32_t idx(int t) const { return up() ? (node >> 1) + (uint32_t)t : (node >> 1) - (uint32_t)t; }
  __device__ __forceinline__ uint32_t par(int q) const { return q == 0 ? (p01 & 0xFFFFu) : q == 1 ? (p01 >> 16) : q == 2 ? (p23 & 0xFFFFu) : (p23 >> 16); }
  __device__ __forceinline__ int npar() const {
    if (source()) return 0;  // (post.cpp: seg_parents)
    return (int)(((p01 & 0xFFFFu) != 0xFFFFu) + ((p01 >> 16) != 0xFFFFu) + ((p23 & 0xFFFFu) != 0xFFFFu) + ((p23 >> 16) != 0xFFFFu));
  }
};
__device__ __forceinline__ SegD seg_load(const SegRec* segs, uint32_t q) {
  const uint4 a = ((const uint4*)(segs + q))[0];
  const uint2 b = ((const uint2*)(segs + q))[2];
  SegD s;
  s.node = a.x; s.dl = a.y; s.ts_tt = a.w; s.p01 = b.x; s.p23 = b.y;
  s.flags = ((const uint32_t*)(segs + q))[6];
  return s;
}

// (FIRST WAVE) K[0 .. n): sorted keys.  PACKED: first << 32 | up << 31 | length - 1 << 16 | segment (the vertex
// intervals); else first << 32 | last (bit 63: a direction some callers sort by).  Writes the merged intervals {first,
// last} to out and returns how many; merges overlapping intervals, and adjacent ones when asked.  *overlap: two overlap.
template <bool PACKED>
__device__ __forceinline__ uint32_t merge_sorted(const uint64_t* K, uint32_t n, bool adjacent, uint32_t* out, int lane, bool* overlap) {
  uint32_t M = 0, pm = 0;  // highest (last + 1) so far
  bool ov = false;
  for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
    const uint32_t i = i0 + (uint32_t)lane;
    const bool h = i < n;
    const uint64_t key = h ? K[i] : 0ull;
    const uint32_t lo = (uint32_t)(key >> 32) & 0x7FFFFFFFu;
    const uint32_t hi = PACKED ? lo + (((uint32_t)key >> 16) & 0x7FFFu) : (uint32_t)key;
    const uint32_t sa = wave_scan_max(h ? hi + 1u : 0u);
    uint32_t xa = wave_shr1(sa);
    xa = max(xa, pm);  // over everything in front of element i
    if (__ballot(h && xa > lo)) ov = true;
    const bool start = h && (xa == 0u || (adjacent ? lo > xa : lo >= xa));
    const uint64_t sm = __ballot(start);
    const uint32_t g = M + (uint32_t)__popcll(sm & below(lane));
    if (start) {
      out[2u * g] = lo;
      if (g > 0u) out[2u * (g - 1u) + 1u] = xa - 1u;
    }
    M += (uint32_t)__popcll(sm);
    pm = max(pm, rl(sa, 63));
  }
  if (M > 0u && lane == 0) out[2u * (M - 1u) + 1u] = pm - 1u;
  *overlap = ov;
  return M;
}

// ascending bitonic sort of n2 (a power of two >= 2) 64-bit keys in LDS by the workgroup: four independent
// compare-exchanges of a stage in flight per thread (the pairs of a stage are disjoint; one at a time the loop waits out
// an LDS round trip per pair)
template <uint32_t NT>
__device__ __forceinline__ void blk_sort64(uint64_t* a, uint32_t n2, uint32_t tid) {
  const uint32_t half = n2 >> 1;
  for (uint32_t k = 2; k <= n2; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      uint32_t t = tid;
      for (; t + 3u * NT < half; t += 4u * NT) {
        uint32_t i[4], l[4];
        uint64_t p[4], q[4];
#pragma unroll
        for (int x = 0; x < 4; x++) {
          const uint32_t tx = t + NT * (uint32_t)x;
          i[x] = ((tx & ~(j - 1u)) << 1) | (tx & (j - 1u));
          l[x] = i[x] | j;
          p[x] = a[i[x]]; q[x] = a[l[x]];
        }
#pragma unroll
        for (int x = 0; x < 4; x++)
          if ((p[x] > q[x]) == ((i[x] & k) == 0u)) { a[i[x]] = q[x]; a[l[x]] = p[x]; }
      }
      for (; t < half; t += NT) {
        const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));
        const uint32_t l = i | j;
        const uint64_t p = a[i], q = a[l];
        if ((p > q) == ((i & k) == 0u)) { a[i] = q; a[l] = p; }
      }
      bsync<NT>();
    }
}
// K[0 .. n) padded with the largest key to a power of two and sorted (all threads)
template <uint32_t NT>
__device__ __forceinline__ void blk_sort_padded(uint64_t* K, uint32_t n, uint32_t tid) {
  if (n < 2u) return;
  const uint32_t n2 = pow2_at_least(n);
  for (uint32_t i = n + tid; i < n2; i += NT) K[i] = ~0ull;
  bsync<NT>();
  blk_sort64<NT>(K, n2, tid);
}
// (FIRST WAVE) drops repeated keys of the sorted K[0 .. n) in place; returns how many are left
__device__ __forceinline__ uint32_t wave_unique(uint64_t* K, uint32_t n, int lane) {
  uint32_t w = 0;
  for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
    const uint32_t i = i0 + (uint32_t)lane;
    const bool h = i < n;
    const uint64_t key = h ? K[i] : 0ull;
    const bool keep = h && (i == 0u || K[i - 1u] != key);
    const uint64_t m = __ballot(keep);
    lds_sync();  // (every lane has read its element and the one in front: the writes land at or in front of the chunk)
    if (keep) K[w + (uint32_t)__popcll(m & below(lane))] = key;
    w += (uint32_t)__popcll(m);
    lds_sync();
  }
  return w;
}

#define D2_LAP(i) do { if (A.prof) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); if (tid == 0u) { atomicAdd(A.prof + (i), t_ - t_lap); if (plog) plog[3 + (i)] = t_ - t_lap; } t_lap = t_; } } while (0)
// the gap's verdict word, behind everything the workgroup wrote for the gap: its trace wave (g2s_d3_trace) may be waiting
template <uint32_t NT>
__device__ __forceinline__ void d2_publish(GapOut* go, uint32_t dflags, bool behind_the_launch) {
  if (behind_the_launch) {  // (nobody reads the gap's results before the launch has ended: no release per closure — on this
    // chip a release at device scope writes the workgroup's whole L2 back, and an acquire empties it for everybody)
    if (threadIdx.x == 0) go->dflags = dflags;
    return;
  }
  __threadfence();
  bsync<NT>();
  if (threadIdx.x == 0) __hip_atomic_store(&go->dflags, dflags, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// returns 0: analysed; 1: beyond this instantiation's capacities; 2: no room for the runs / given up
template <class C>
__device__ int d2_one(uint32_t* lds, const D2Args& A, uint32_t gap, uint32_t* scr) {
  constexpr uint32_t NT = C::NT;
  const uint32_t tid = threadIdx.x;
  const int lane = (int)(tid & 63u);
  const uint32_t wave = tid >> 6;
  unsigned long long t_lap = A.prof ? __builtin_amdgcn_s_memrealtime() : 0ull;
  GapOut* go = A.outs + gap;
  const uint32_t nrec = uni(go->n_xl);
  const SegRec* segs = (const SegRec*)(A.sub + uni((uint32_t)go->sub_off) + ((uint64_t)uni((uint32_t)(go->sub_off >> 32)) << 32));
  SegRec* segs_w = const_cast<SegRec*>(segs);
  const GapDev gd = GapSrc{A.gaps, A.lite, A.lite_e, A.all_paths}.load(gap);
  const int lmf = (int)uni((uint32_t)gd.lmf), rmf = (int)uni((uint32_t)gd.rmf);
  const uint32_t* targets = A.flank_nodes + uni(gd.flank_off) + (uint32_t)(lmf + 1) + (uint32_t)(rmf + 1);
  const bool all_paths = A.all_paths != 0;
  const uint32_t sinknode = (all_paths && rmf >= 1) ? uni(targets[rmf - 1]) : G2S_DEV_INVALID;  // Q3 / Q4
  const int lo_sink = max(0, lmf + (int)uni((uint32_t)gd.g) - (int)uni((uint32_t)gd.e));         // :1196
  const uint32_t reached = uni(targets[uni((uint32_t)go->reached_j)]);
  const bool t_is_s = !all_paths;  // -best-only: the traceback starts are the sinks (:1245-1259)
  const int n_len = (int)uni((uint32_t)go->n_len), len0 = (int)uni((uint32_t)go->len[0]), len1 = (int)uni((uint32_t)go->len[1]);
  const uint32_t dflags0 = uni(go->dflags);
  if (nrec == 0u || nrec > C::NREC) return 1;
  // position of the sink state inside a segment of the S closure, or -1 (post.cpp: sinkpos)
  auto sink_pos = [&](const SegD& s) -> int {
    const int ts = s.ts();
    if (ts < 0) return -1;
    int sp = -1;
    const int pk = seg_pos(s.node, (uint32_t)s.len(), sinknode);
    if (pk >= 0 && s.d0() + pk >= lo_sink) sp = pk;
    if (t_is_s) {
      const int pt = seg_pos(s.node, (uint32_t)s.len(), reached);
      if (pt >= 0 && (s.d0() + pt == len0 || (n_len > 1 && s.d0() + pt == len1))) sp = pt;
    }
    return sp > ts ? -1 : sp;
  };
  uint64_t* K = (uint64_t*)lds;
  uint64_t* GM = (uint64_t*)(lds + C::MAIN_BYTES / 4u);   // per 64 cuts: which of them have a run behind them ...
  uint32_t* GP = (uint32_t*)(GM + C::CHUNKS);             // ... and how many such runs lie in front of the 64 (or: chunk totals)
  uint32_t* PM = GP + C::CHUNKS;                          // a bit per node: pass-through (before that: a source's entry)
  uint32_t* KM = PM + C::NV / 32u;                        // a bit per node: a sink position
  uint32_t* hdr = KM + C::NV / 32u;                       // counters
  unsigned long long* acc64 = (unsigned long long*)(hdr + H_A64);
  // scratch
  uint32_t* VM = scr;                      // merged vertex intervals
  uint32_t* UI = VM + 2u * C::NS;          // chain edges of upward segments, by the lower of their two k-mers
  uint32_t* DI = UI + 2u * C::NS;          // ... of downward segments
  uint32_t* LOOPS = DI + 2u * C::NS;
  uint32_t* CUT = LOOPS + C::NS;
  uint64_t* SP = (uint64_t*)(CUT + C::BP); // the distinct (parent's last k-mer, entry) pairs, sorted
  uint32_t* RLO = (uint32_t*)(SP + C::BP); // by node
  uint32_t* RHI = RLO + C::NV;
  uint32_t* NW = RHI + C::NV;              // k-mers of the node | covered upwards << 30 | downwards << 31
  uint32_t* HEAD = NW + C::NV;             // pass-through nodes outside the components: the component their chain leaves
  uint32_t* ORDER = HEAD + C::NV;
  uint32_t* EDGE = ORDER + C::NV;          // from << 16 | to

  unsigned long long* plog = (A.prof && A.log && hdr[H_SLOT] < A.log_cap) ? A.log + 16ull * hdr[H_SLOT] : nullptr;
  bsync<NT>();
  for (uint32_t i = tid; i < C::HDR_WORDS; i += NT) if (i != (uint32_t)H_GAP && i != (uint32_t)H_SLOT) hdr[i] = 0u;
  bsync<NT>();
  // ---- the S closure's index intervals, sorted
  for (uint32_t q0 = wave * 64u; q0 < nrec; q0 += NT) {
    const uint32_t q = q0 + (uint32_t)lane;
    const bool h = q < nrec;
    SegD s;
    if (h) s = seg_load(segs, q);
    const int ts = h ? s.ts() : -1;
    const bool in_s = ts >= 0;
    const uint64_t m = __ballot(in_s);
    const uint32_t pos = wave_reserve(&hdr[H_N0], m, lane);
    if (in_s && pos < C::NS) {
      const uint32_t lo = s.up() ? (s.node >> 1) : (s.node >> 1) - (uint32_t)ts;
      K[pos] = ((uint64_t)lo << 32) | (s.up() ? 0x80000000u : 0u) | ((uint32_t)ts << 16) | q;
    }
  }
  bsync<NT>();
  const uint32_t ns = hdr[H_N0];
  if (ns > C::NS) return 1;
  if (ns == 0u) {  // (phase D ran without a sink state: nothing to analyse; the recount is the fill kernel's)
    if (tid == 0u) {
      D2Out o; o.run_off = 0; o.n_runs = 0; o.sub[0] = 2; o.sub[1] = 0; o.sub[2] = 0; o.sub[3] = 0; o.sub[4] = 2; o.sub[5] = 0;
      A.d2out[gap] = o;
      go->sub_vertices = 2; go->sub_edges = 0;
    }
    d2_publish<NT>(go, (dflags0 & ~G2S_DEVA_SINK_SAFE) | G2S_DEVA_ANALYSED | G2S_DEVA_RUNS, A.behind != 0u);
    return 0;
  }
  blk_sort_padded<NT>(K, ns, tid);
  bsync<NT>();
  if (wave == 0u) {
    bool ov = false;
    const uint32_t mv = merge_sorted<true>(K, ns, false, VM, lane, &ov);
    if (lane == 0) { hdr[H_M0] = mv; hdr[H_FLAG] = ov ? 1u : 0u; }
  }
  gsync<NT>();
  const uint32_t MV = hdr[H_M0];
  const bool overlap = hdr[H_FLAG] != 0u;
  for (uint32_t i0 = wave * 64u; i0 < MV; i0 += NT) {
    const uint32_t i = i0 + (uint32_t)lane;
    const uint32_t len = wave_sum(i < MV ? VM[2u * i + 1u] - VM[2u * i] + 1u : 0u);
    if (lane == 0) atomicAdd(&hdr[H_R0], len);
  }
  bsync<NT>();
  const uint32_t V = 2u + hdr[H_R0];
  bsync<NT>();
  D2_LAP(0);

  if (!overlap) {
    // ================= every k-mer at one depth: the branch rule as a prefix sum (post.cpp: seg_analyze) ==========
    // by segment, behind the keys: safe_a | safe_b << 1 | split << 2 | children on paths to a sink << 24
    uint32_t* sv = lds + 2u * C::NS;
    int* T = (int*)GP;  // per 64 segments (in the order of the sweep): the counter's change, then the counter in front
    for (uint32_t q = tid; q < nrec; q += NT) sv[q] = 0u;
    // totals; children per parent.  H_R0: states, H_R1: edges, H_R2: segments the source leads into, H_R3: sink states
    if (tid == 0u) { hdr[H_R0] = 0u; hdr[H_R1] = 0u; hdr[H_R2] = 0u; hdr[H_R3] = 0u; }
    bsync<NT>();
    for (uint32_t q0 = wave * 64u; q0 < nrec; q0 += NT) {
      const uint32_t q = q0 + (uint32_t)lane;
      const bool h = q < nrec;
      SegD s;
      if (h) s = seg_load(segs, q);
      const int ts = h ? s.ts() : -1;
      const bool in_s = ts >= 0;
      const int np = in_s ? s.npar() : 0;
      const int sp = in_s ? sink_pos(s) : -1;
      const bool src = in_s && s.source();
      const uint32_t a0 = wave_sum(in_s ? (uint32_t)ts + 1u : 0u);
      const uint32_t a1 = wave_sum(in_s ? (uint32_t)ts + (src ? 1u : (uint32_t)np) + (sp >= 0 ? 1u : 0u) : 0u);
      const uint32_t a2 = (uint32_t)__popcll(__ballot(src)), a3 = (uint32_t)__popcll(__ballot(sp >= 0));
      if (lane == 0) { atomicAdd(&hdr[H_R0], a0); atomicAdd(&hdr[H_R1], a1); atomicAdd(&hdr[H_R2], a2); atomicAdd(&hdr[H_R3], a3); }
      if (in_s && !src)
        for (int x = 0; x < 4; x++) { const uint32_t p = s.par(x); if (p != 0xFFFFu && p < nrec) atomicAdd(&sv[p], 1u << 24); }
    }
    bsync<NT>();
    const uint32_t n_s = hdr[H_R0], edges = hdr[H_R1], src_out = hdr[H_R2], sink_in = hdr[H_R3];
    bsync<NT>();
    // what a segment does to the counter (parents first = descending record index): -(in - 1) at its entry, +1 at a
    // sink inside it, +(out - 1) behind its last state
    auto seg_delta = [&](uint32_t i, SegD* s_out, int* ts_out, int* sp_out, int* d_in_out, int* d_mid_out) -> int {
      const bool h = i < nrec;
      const uint32_t q = h ? nrec - 1u - i : 0u;
      SegD s;
      if (h) s = seg_load(segs, q); else { s.node = 0; s.dl = 0; s.ts_tt = 0x7FFF7FFFu; s.p01 = s.p23 = 0xFFFFFFFFu; s.flags = 0; }
      const int ts = h ? s.ts() : -1;
      const bool in_s = ts >= 0;
      const int sp = in_s ? sink_pos(s) : -1;
      const int din = in_s ? (s.source() ? 1 : s.npar()) : 0;
      const int outs = in_s ? (int)(sv[q] >> 24) : 0;
      const int d_in = (in_s && din > 1) ? -(din - 1) : 0;
      const int d_mid = (in_s && sp >= 0 && sp < ts) ? 1 : 0;  // out-degree 2: the next state and the sink
      const int dout = in_s ? (ts == s.len() - 1 ? outs : 0) + (sp == ts ? 1 : 0) : 0;
      const int d_out = dout > 1 ? dout - 1 : 0;
      *s_out = s; *ts_out = ts; *sp_out = sp; *d_in_out = d_in; *d_mid_out = d_mid;
      return d_in + d_mid + d_out;
    };
    const uint32_t nchunks = (nrec + 63u) / 64u;
    for (uint32_t c = wave; c < nchunks; c += NT / 64u) {
      SegD s; int ts, sp, d_in, d_mid;
      const int total = seg_delta(c * 64u + (uint32_t)lane, &s, &ts, &sp, &d_in, &d_mid);
      const int sum = (int)wave_sum((uint32_t)total);
      if (lane == 0) T[c] = sum;
    }
    bsync<NT>();
    if (wave == 0u) {  // the counter in front of every chunk
      int bc = 1 + (src_out > 1u ? (int)src_out - 1 : 0);  // the source pseudo-vertex comes first
      for (uint32_t c0 = 0; c0 < nchunks; c0 += 64u) {
        const uint32_t c = c0 + (uint32_t)lane;
        const int mine = c < nchunks ? T[c] : 0;
        const int incl = (int)wave_scan((uint32_t)mine, lane);
        lds_sync();
        if (c < nchunks) T[c] = bc + incl - mine;
        bc += (int)rl((uint32_t)incl, 63);
      }
      if (lane == 0) hdr[H_R0] = (uint32_t)bc;  // (behind the last segment)
    }
    bsync<NT>();
    for (uint32_t c = wave; c < nchunks; c += NT / 64u) {
      SegD s; int ts, sp, d_in, d_mid;
      const uint32_t i = c * 64u + (uint32_t)lane;
      const int total = seg_delta(i, &s, &ts, &sp, &d_in, &d_mid);
      const int incl = (int)wave_scan((uint32_t)total, lane);
      const int at_entry = T[c] + incl - total + d_in;
      if (ts >= 0) {
        const uint32_t q = nrec - 1u - i;
        const bool sa = at_entry == 1, sb = at_entry + d_mid == 1;
        const int split = d_mid ? sp : ts;
        sv[q] = (sv[q] & 0xFF000000u) | (sa ? 1u : 0u) | (sb ? 2u : 0u) | ((uint32_t)split << 2);
        // (the records carry the verdicts as the fill kernel's own analysis leaves them)
        segs_w[q].ts_tt = (s.ts_tt & 0x7FFF7FFFu) | (sa ? 0x8000u : 0u) | (sb ? 0x80000000u : 0u);
        segs_w[q].pad = (uint32_t)split;
      }
    }
    bsync<NT>();
    bool sink_safe = false;
    { int bc = (int)hdr[H_R0]; if (sink_in >= 1u) { if (sink_in > 1u) bc -= (int)sink_in - 1; sink_safe = bc == 1; } }
    // runs, in the order of the sorted intervals: one per segment, two where a sink splits it
    const uint32_t schunks = (ns + 63u) / 64u;
    for (uint32_t c = wave; c < schunks; c += NT / 64u) {
      const uint32_t j = c * 64u + (uint32_t)lane;
      const bool h = j < ns;
      const uint32_t low = h ? (uint32_t)K[j] : 0u;
      const uint32_t ts = (low >> 16) & 0x7FFFu, q = low & 0xFFFFu;
      const uint32_t cnt = wave_sum(h ? (((sv[q] >> 2) & 0x7FFFu) < ts ? 2u : 1u) : 0u);
      if (lane == 0) GP[c] = cnt;
    }
    bsync<NT>();
    if (wave == 0u) {
      uint32_t run = 0;
      for (uint32_t c0 = 0; c0 < schunks; c0 += 64u) {
        const uint32_t c = c0 + (uint32_t)lane;
        const uint32_t mine = c < schunks ? GP[c] : 0u;
        const uint32_t incl = wave_scan(mine, lane);
        lds_sync();
        if (c < schunks) GP[c] = run + incl - mine;
        run += rl(incl, 63);
      }
      if (lane == 0) {
        const unsigned long long base = atomicAdd(A.run_cursor, (unsigned long long)run);
        hdr[H_R1] = run; hdr[H_BASE_LO] = (uint32_t)base; hdr[H_BASE_HI] = (uint32_t)(base >> 32);
      }
    }
    bsync<NT>();
    const uint32_t total_runs = hdr[H_R1];
    const unsigned long long base = ((unsigned long long)hdr[H_BASE_HI] << 32) | hdr[H_BASE_LO];
    if (base + total_runs > A.run_cap) return 2;
    uint32_t* runs = A.runs + 2ull * base;
    for (uint32_t c = wave; c < schunks; c += NT / 64u) {
      const uint32_t j = c * 64u + (uint32_t)lane;
      const bool h = j < ns;
      const uint64_t key = h ? K[j] : 0ull;
      const uint32_t low = (uint32_t)key, lo = (uint32_t)(key >> 32);
      const uint32_t ts = (low >> 16) & 0x7FFFu, q = low & 0xFFFFu;
      const bool up = (low & 0x80000000u) != 0u;
      const uint32_t v = h ? sv[q] : 0u;
      const uint32_t split = (v >> 2) & 0x7FFFu;
      const uint32_t sa = v & 1u, sb = (v >> 1) & 1u;
      const uint32_t mine = h ? (split < ts ? 2u : 1u) : 0u;
      const uint32_t incl = wave_scan(mine, lane);
      const uint32_t at = GP[c] + incl - mine;
      if (h) {
        const uint32_t hi = lo + ts;
        if (mine == 1u) { runs[2u * at] = lo; runs[2u * at + 1u] = hi | (sa << 31); }
        else if (up) {  // states 0 .. split, then the rest
          runs[2u * at] = lo; runs[2u * at + 1u] = (lo + split) | (sa << 31);
          runs[2u * at + 2u] = lo + split + 1u; runs[2u * at + 3u] = hi | (sb << 31);
        } else {        // (a downward segment: state t is k-mer hi - t)
          runs[2u * at] = lo; runs[2u * at + 1u] = (hi - split - 1u) | (sb << 31);
          runs[2u * at + 2u] = hi - split; runs[2u * at + 3u] = hi | (sa << 31);
        }
      }
    }
    if (tid == 0u) {
      D2Out o;
      o.run_off = (uint32_t)base; o.n_runs = total_runs;
      o.sub[0] = n_s + 2u; o.sub[1] = edges; o.sub[2] = 0u; o.sub[3] = 0u; o.sub[4] = n_s + 2u; o.sub[5] = edges;
      A.d2out[gap] = o;
      go->sub_vertices = n_s + 2u; go->sub_edges = edges;
    }
    d2_publish<NT>(go, (dflags0 & ~G2S_DEVA_SINK_SAFE) | G2S_DEVA_ANALYSED | G2S_DEVA_RUNS | (sink_safe ? G2S_DEVA_SINK_SAFE : 0u), A.behind != 0u);
    D2_LAP(1);
    if (A.prof && tid == 0u) atomicAdd(A.prof + 19, 1ull);
    return 0;
  }

  // ================= a k-mer at several depths: the graph of runs (post.cpp: seg_analyze_runs) ======================
  // (all the sorting first; the sort buffer then holds the tables the rest looks things up in)
  // ---- chain edges of the upward and of the downward segments as merged intervals (adjacent ones merge): one pass
  // over the segments, one sort (the direction is the key's top bit)
  if (tid == 0u) { hdr[H_N0] = 0u; hdr[H_N1] = 0u; hdr[H_N2] = 0u; hdr[H_M1] = 0u; hdr[H_M2] = 0u; }
  bsync<NT>();
  for (uint32_t q0 = wave * 64u; q0 < nrec; q0 += NT) {
    const uint32_t q = q0 + (uint32_t)lane;
    const bool h = q < nrec;
    SegD s;
    if (h) s = seg_load(segs, q); else s.node = 0;
    const int ts = h ? s.ts() : -1;
    const bool take = ts > 0;
    const uint64_t m = __ballot(take);
    const uint32_t pos = wave_reserve(&hdr[H_N0], m, lane);
    if (take && pos < C::NS) {
      const uint32_t lo = s.up() ? (s.node >> 1) : (s.node >> 1) - (uint32_t)ts;
      K[pos] = (s.up() ? 0ull : 1ull << 63) | ((uint64_t)lo << 32) | (lo + (uint32_t)ts - 1u);
    }
    const uint32_t ups = (uint32_t)__popcll(__ballot(take && s.up()));
    if (lane == 0 && ups) atomicAdd(&hdr[H_N1], ups);
  }
  bsync<NT>();
  {
    const uint32_t n = hdr[H_N0], n_up = hdr[H_N1];  // (n <= ns <= NS)
    blk_sort_padded<NT>(K, n, tid);
    bsync<NT>();
    if (wave == 0u) {
      bool ov;
      uint32_t mu = 0, md = 0;
      if (n_up > 0u) mu = merge_sorted<false>(K, n_up, true, UI, lane, &ov);
      if (n > n_up) md = merge_sorted<false>(K + n_up, n - n_up, true, DI, lane, &ov);
      if (lane == 0) { hdr[H_M1] = mu; hdr[H_M2] = md; }
    }
    gsync<NT>();
  }
  const uint32_t MU = hdr[H_M1], MD = hdr[H_M2];
  D2_LAP(2);
  // ---- the cuts: ends of every interval, sink positions; and how many parent -> entry pairs there are
  if (tid == 0u) { hdr[H_N0] = 0u; hdr[H_N1] = 0u; }
  bsync<NT>();
  for (uint32_t q0 = wave * 64u; q0 < nrec; q0 += NT) {
    const uint32_t q = q0 + (uint32_t)lane;
    const bool h = q < nrec;
    SegD s;
    if (h) s = seg_load(segs, q); else { s.node = 0; s.flags = 0; s.p01 = s.p23 = 0xFFFFFFFFu; }
    const int ts = h ? s.ts() : -1;
    const bool in_s = ts >= 0;
    const int sp = in_s ? sink_pos(s) : -1;
    // (a parent's last k-mer — post.cpp pushes it as a cut — is an end of the parent's own interval: a segment with a
    // child on a path to a sink lies on such paths whole)
    const uint32_t np = (in_s && !s.source()) ? (uint32_t)s.npar() : 0u;
    const uint32_t k = in_s ? 2u + (sp >= 0 ? 1u : 0u) : 0u;
    const uint32_t incl = wave_scan(k, lane);
    const uint32_t tot = rl(incl, 63), nps = wave_sum(np);
    uint32_t base = 0;
    if (lane == 0) { base = atomicAdd(&hdr[H_N0], tot); atomicAdd(&hdr[H_N1], nps); }
    uint32_t w = rl(base, 0) + incl - k;
    if (in_s && w + k <= C::BP) {
      K[w++] = (uint64_t)s.idx(0);
      K[w++] = (uint64_t)s.idx(ts);
      if (sp >= 0) K[w++] = (uint64_t)s.idx(sp);
    }
  }
  bsync<NT>();
  {
    const uint32_t n = hdr[H_N0];
    if (n > C::BP || hdr[H_N1] > C::BP) return 1;
    blk_sort_padded<NT>(K, n, tid);
    bsync<NT>();
    if (wave == 0u) { const uint32_t u = wave_unique(K, n, lane); if (lane == 0) hdr[H_M0] = u; }
    bsync<NT>();
  }
  const uint32_t nb = hdr[H_M0];
  for (uint32_t i = tid; i < nb; i += NT) CUT[i] = (uint32_t)K[i];
  bsync<NT>();
  if (tid == 0u) hdr[H_N0] = 0u;
  bsync<NT>();
  // ---- the edges from a parent's last k-mer to an entry, sorted, each once (boost::edge(u, v).second)
  for (uint32_t q0 = wave * 64u; q0 < nrec; q0 += NT) {
    const uint32_t q = q0 + (uint32_t)lane;
    const bool h = q < nrec;
    SegD s;
    if (h) s = seg_load(segs, q); else { s.node = 0; s.flags = 0; s.p01 = s.p23 = 0xFFFFFFFFu; s.ts_tt = 0x7FFF7FFFu; }
    const bool in_s = h && s.ts() >= 0 && !s.source();
    const uint32_t k = in_s ? (uint32_t)s.npar() : 0u;
    const uint32_t incl = wave_scan(k, lane);
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&hdr[H_N0], rl(incl, 63));
    uint32_t w = rl(base, 0) + incl - k;  // (the total is what was counted above: <= BP)
    if (k)
      for (int x = 0; x < 4; x++) {
        const uint32_t p = s.par(x);
        if (p == 0xFFFFu || p >= nrec) continue;
        const SegD ps = seg_load(segs, p);
        if (w < C::BP) K[w++] = ((uint64_t)ps.idx(ps.len() - 1) << 32) | s.idx(0);
      }
  }
  bsync<NT>();
  {
    const uint32_t n = min(hdr[H_N0], C::BP);
    bsync<NT>();
    blk_sort_padded<NT>(K, n, tid);
    bsync<NT>();
    if (wave == 0u) { const uint32_t u = wave_unique(K, n, lane); if (lane == 0) hdr[H_M0] = u; }
    bsync<NT>();
  }
  const uint32_t nsp = hdr[H_M0];
  for (uint32_t i = tid; i < nsp; i += NT) SP[i] = K[i];
  gsync<NT>();
  D2_LAP(3);
  // ---- the sort buffer becomes the tables: cuts, vertex intervals, chain intervals of either direction
  uint32_t* CUTL = lds;
  uint32_t* VML = lds + C::BP;
  uint32_t* UL = VML + 2u * C::NS;
  uint32_t* DL = UL + 2u * MU;
  for (uint32_t i = tid; i < nb; i += NT) CUTL[i] = CUT[i];
  for (uint32_t i = tid; i < 2u * MV; i += NT) VML[i] = VM[i];
  for (uint32_t i = tid; i < 2u * MU; i += NT) UL[i] = UI[i];
  for (uint32_t i = tid; i < 2u * MD; i += NT) DL[i] = DI[i];
  if (tid == 0u) { hdr[H_N0] = 0u; hdr[H_N1] = 0u; hdr[H_N2] = 0u; hdr[H_R0] = 0u; acc64[0] = 0ull; acc64[1] = 0ull; acc64[2] = 0ull; }
  bsync<NT>();
  // ---- runs: every cut k-mer alone, and what lies between two cuts of one vertex interval.  First which cuts have a
  // run behind them (a mask per 64 cuts) and how many such runs lie in front of every 64; then the runs and the chain's
  // own edges between neighbouring runs
  const uint32_t cchunks = (nb + 63u) / 64u;
  auto cut_gap = [&](uint32_t j, uint32_t* c_out, uint32_t* cn_out, bool* more_out) -> bool {
    const bool h = j < nb;
    const uint32_t c = h ? CUTL[j] : 0u, cn = (h && j + 1u < nb) ? CUTL[j + 1u] : 0u;
    const bool more = h && j + 1u < nb;
    uint32_t hi_v = 0;
    const bool inside = h && iv_has(VML, MV, c, &hi_v);
    *c_out = c; *cn_out = cn; *more_out = more;
    return inside && more && cn <= hi_v && cn > c + 1u;
  };
  for (uint32_t ch = wave; ch < cchunks; ch += NT / 64u) {
    uint32_t c, cn; bool more;
    const uint64_t m = __ballot(cut_gap(ch * 64u + (uint32_t)lane, &c, &cn, &more));
    if (lane == 0) { GM[ch] = m; GP[ch] = (uint32_t)__popcll(m); }
  }
  bsync<NT>();
  if (wave == 0u) {
    uint32_t run = 0;
    for (uint32_t c0 = 0; c0 < cchunks; c0 += 64u) {
      const uint32_t c = c0 + (uint32_t)lane;
      const uint32_t mine = c < cchunks ? GP[c] : 0u;
      const uint32_t incl = wave_scan(mine, lane);
      lds_sync();
      if (c < cchunks) GP[c] = run + incl - mine;
      run += rl(incl, 63);
    }
    if (lane == 0) hdr[H_R0] = run;
  }
  bsync<NT>();
  const uint32_t R = nb + hdr[H_R0];
  const uint32_t NV = R + 2u;
  if (NV > C::NV) return 1;
  uint32_t* n_edges = &hdr[H_N0];
  auto emit = [&](bool take, uint32_t from, uint32_t to) {  // (order does not matter: a place by an atomic counter)
    const uint64_t m = __ballot(take);
    const uint32_t at = wave_reserve(n_edges, m, lane);
    if (take && at < C::E) EDGE[at] = (from << 16) | to;
  };
  for (uint32_t ch = wave; ch < cchunks; ch += NT / 64u) {
    const uint32_t j = ch * 64u + (uint32_t)lane;
    uint32_t c, cn; bool more;
    const bool gp = cut_gap(j, &c, &cn, &more);
    const bool h = j < nb;
    const uint32_t r = j + GP[ch] + (uint32_t)__popcll(GM[ch] & below(lane));
    uint32_t internal = 0;
    // chain edges leave the cut k-mer towards the next run (c -> c + 1 upwards, c + 1 -> c downwards) and, where
    // a run lies between, that run towards the next cut
    const bool next_to = more && (gp || cn == c + 1u);
    const bool u0 = next_to && iv_has(UL, MU, c), d0 = next_to && iv_has(DL, MD, c);
    const bool u1 = gp && iv_has(UL, MU, cn - 1u), d1 = gp && iv_has(DL, MD, cn - 1u);
    if (h) {
      RLO[2u + r] = c; RHI[2u + r] = c; NW[2u + r] = 1u;
      if (gp) {
        const uint32_t L = cn - c - 1u;
        uint32_t cover = 0u;
        if (L > 1u) {
          const bool u = iv_has(UL, MU, c + 1u), d = iv_has(DL, MD, c + 1u);
          cover = (u ? 0x40000000u : 0u) | (d ? 0x80000000u : 0u);
          internal = ((u ? 1u : 0u) + (d ? 1u : 0u)) * (L - 1u);
        }
        RLO[3u + r] = c + 1u; RHI[3u + r] = cn - 1u; NW[3u + r] = L | cover;
      }
    }
    const uint32_t isum = wave_sum(internal);
    if (lane == 0 && isum) atomicAdd(&acc64[0], (unsigned long long)isum);
    emit(h && u0, 2u + r, 3u + r);
    emit(h && d0, 3u + r, 2u + r);
    emit(h && u1, 3u + r, 4u + r);
    emit(h && d1, 4u + r, 3u + r);
  }
  if (tid == 0u) { RLO[0] = RHI[0] = 0u; NW[0] = 1u; RLO[1] = RHI[1] = 0u; NW[1] = 1u; }  // 0: sink, 1: source
  bsync<NT>();
  D2_LAP(4);
  // the node of k-mer x (every end of an edge is a cut: a run of its own)
  auto node_of = [&](uint32_t x) -> uint32_t {
    const uint32_t j = cut_index(CUTL, nb, x);
    return 2u + j + GP[j >> 6] + (uint32_t)__popcll(GM[j >> 6] & ((1ull << (j & 63u)) - 1ull));
  };
  // ---- the other edges, one per distinct edge of the reference's graph: those that double a chain edge go, self
  // loops (a homopolymer k-mer: never between components) are set aside
  for (uint32_t i0 = wave * 64u; i0 < nsp; i0 += NT) {
    const uint32_t i = i0 + (uint32_t)lane;
    const bool h = i < nsp;
    const uint64_t e = h ? SP[i] : 0ull;
    const uint32_t a = (uint32_t)(e >> 32), b = (uint32_t)e;
    bool keep = h;
    if (keep && b == a + 1u && iv_has(UL, MU, a)) keep = false;
    if (keep && a == b + 1u && iv_has(DL, MD, b)) keep = false;
    const bool loop = keep && a == b;
    const uint64_t lm = __ballot(loop);
    const uint32_t lat = wave_reserve(&hdr[H_N1], lm, lane);
    if (loop && lat < C::NS) LOOPS[lat] = node_of(a);
    const bool edge = keep && !loop;
    emit(edge, edge ? node_of(a) : 0u, edge ? node_of(b) : 0u);
  }
  // the source's edges to the entries that are left-flank k-mers (:1303-1305) and the sink positions' edges to the sink
  // (:1216-1226), one per k-mer: a bit per node, then one edge per bit
  {
    uint32_t* SM = PM;  // (the pass-through bits come later)
    for (uint32_t i = tid; i < C::NV / 32u; i += NT) { SM[i] = 0u; KM[i] = 0u; }
    bsync<NT>();
    for (uint32_t q = tid; q < nrec; q += NT) {
      const SegD s = seg_load(segs, q);
      if (s.ts() < 0) continue;
      if (s.source()) { const uint32_t v = node_of(s.idx(0)); atomicOr(&SM[v >> 5], 1u << (v & 31u)); }
      const int sp = sink_pos(s);
      if (sp >= 0) { const uint32_t v = node_of(s.idx(sp)); atomicOr(&KM[v >> 5], 1u << (v & 31u)); }
    }
    bsync<NT>();
    for (uint32_t v0 = wave * 64u; v0 < NV; v0 += NT) {
      const uint32_t v = v0 + (uint32_t)lane;
      const bool h = v < NV;
      emit(h && ((SM[v >> 5] >> (v & 31u)) & 1u), 1u, v);
      emit(h && ((KM[v >> 5] >> (v & 31u)) & 1u), v, 0u);
    }
  }
  gsync<NT>();
  const uint32_t ne = hdr[H_N0], nloops = hdr[H_N1];
  if (plog && tid == 0u) { plog[1] = ns | ((unsigned long long)NV << 32); plog[2] = ne; }
  if (ne > C::E || nloops > C::NS) return 1;
  const unsigned long long e_all = acc64[0] + ne + nloops;
  bsync<NT>();
  D2_LAP(5);

  // ---- the graph into LDS (over the tables): offsets and adjacency both ways, live degrees
  uint16_t* off_f = (uint16_t*)lds;
  uint16_t* off_r = (uint16_t*)((char*)off_f + C::OFF_BYTES);
  uint16_t* adj_f = (uint16_t*)((char*)off_r + C::OFF_BYTES);
  uint16_t* adj_r = adj_f + C::E;
  uint32_t* degi = (uint32_t*)(adj_r + C::E);   // packed, NV / 2 words
  uint32_t* dego = degi + C::NV / 2u;
  uint32_t* comp = dego + C::NV / 2u;
  uint16_t* wl0 = (uint16_t*)(comp + C::NV);
  uint16_t* wl1 = wl0 + C::NV;
  for (uint32_t i = tid; i < C::NV / 2u; i += NT) { degi[i] = 0u; dego[i] = 0u; ((uint32_t*)wl0)[i] = 0u; ((uint32_t*)wl1)[i] = 0u; }
  for (uint32_t i = tid; i < C::NV / 32u; i += NT) PM[i] = 0u;
  for (uint32_t v = tid; v < NV; v += NT) comp[v] = D2_ALIVE;
  if (tid == 0u) { hdr[H_NEXT] = 0u; hdr[H_ORDER] = 0u; hdr[H_BIGS] = 0u; hdr[H_N0] = 0u; hdr[H_N1] = 0u; hdr[H_N2] = 0u; acc64[0] = 0ull; }
  bsync<NT>();
  for (uint32_t i = tid; i < ne; i += NT) {
    const uint32_t e = EDGE[i];
    pk_add(dego, e >> 16, 1u);
    pk_add(degi, e & 0xFFFFu, 1u);
  }
  bsync<NT>();
  if (wave == 0u) {
    uint32_t run_f = 0, run_r = 0;
    for (uint32_t v0 = 0; v0 < NV; v0 += 64u) {
      const uint32_t v = v0 + (uint32_t)lane;
      const bool h = v < NV;
      const uint32_t df = h ? pk_get(dego, v) : 0u, dr = h ? pk_get(degi, v) : 0u;
      const uint32_t sf = wave_scan(df, lane), sr = wave_scan(dr, lane);
      if (h) { off_f[v] = (uint16_t)(run_f + sf - df); off_r[v] = (uint16_t)(run_r + sr - dr); }
      run_f += rl(sf, 63); run_r += rl(sr, 63);
    }
    if (lane == 0) { off_f[NV] = (uint16_t)run_f; off_r[NV] = (uint16_t)run_r; }
  }
  bsync<NT>();
  {  // (the two work lists serve as fill cursors first)
    uint32_t* cur_f = (uint32_t*)wl0;
    uint32_t* cur_r = (uint32_t*)wl1;
    for (uint32_t i = tid; i < ne; i += NT) {
      const uint32_t e = EDGE[i];
      const uint32_t a = e >> 16, b = e & 0xFFFFu;
      adj_f[(uint32_t)off_f[a] + pk_add(cur_f, a, 1u)] = (uint16_t)b;
      adj_r[(uint32_t)off_r[b] + pk_add(cur_r, b, 1u)] = (uint16_t)a;
    }
  }
  bsync<NT>();
  D2_LAP(6);

  // ---- pass-through nodes leave the graph: a run with one edge in and one out (and not covered in both directions)
  // only hands on what reaches it.  Every other node's edges skip them — the adjacency entry becomes the node at the
  // chain's end — and a pass-through node remembers which edge of which node its chain is.  What is left has as many
  // levels as the paths have BRANCHINGS, not runs (a tenth to a third).
  for (uint32_t v0 = wave * 64u; v0 < NV; v0 += NT) {
    const uint32_t v = v0 + (uint32_t)lane;
    const bool pass = !(A.pass_all & 2u) && v >= 2u && v < NV && pk_get(degi, v) == 1u && pk_get(dego, v) == 1u && (NW[v] & 0xC0000000u) != 0xC0000000u;
    if (pass) comp[v] = D2_PASS;
    const uint64_t m = __ballot(pass);
    if (lane == 0) { PM[v0 >> 5] = (uint32_t)m; if ((v0 >> 5) + 1u < C::NV / 32u) PM[(v0 >> 5) + 1u] = (uint32_t)(m >> 32); }
  }
  bsync<NT>();
  for (uint32_t u = tid; u < NV; u += NT) {
    if (comp[u] != D2_ALIVE) continue;
    uint32_t k = 0;
    for (uint32_t e = off_f[u]; e < (uint32_t)off_f[u + 1u]; e++, k++) {
      uint32_t w = adj_f[e];
      for (uint32_t steps = 0; d2_is_pass(comp[w]) && steps <= NV; steps++) {
        comp[w] = 0x80000000u | (min(k, 15u) << 16) | u;
        w = adj_f[off_f[w]];
      }
      adj_f[e] = (uint16_t)w;
    }
    for (uint32_t e = off_r[u]; e < (uint32_t)off_r[u + 1u]; e++) {
      uint32_t p = adj_r[e];
      for (uint32_t steps = 0; d2_is_pass(comp[p]) && steps <= NV; steps++) p = adj_r[off_r[p]];
      adj_r[e] = (uint16_t)p;
    }
  }
  bsync<NT>();

  // ---- strong components of what is left.  A node without a live edge in, or without one out, is a component of
  // its own: taken (its state becomes its own id) by whoever sees its last such edge go, and its edges dropped by the
  // same lane at once — a lane follows a chain of such nodes without a round trip through the work list, which only
  // takes the second and further nodes a step frees.
  uint32_t* n_next = &hdr[H_NEXT];
  uint16_t* cur = wl0;
  uint16_t* nxt = wl1;
  auto take = [&](uint32_t v) -> bool { return atomicCAS(&comp[v], D2_ALIVE, v) == D2_ALIVE; };
  auto spill = [&](uint32_t v) { nxt[atomicAdd(n_next, 1u)] = (uint16_t)v; };
  // drops the edges of u (which has left the live graph) and of every node that frees along the way
  auto peel_from = [&](uint32_t u) {
    for (uint32_t steps = 0; steps <= NV; steps++) {
      uint32_t next = D2_NONE;
      for (uint32_t e = off_f[u]; e < (uint32_t)off_f[u + 1u]; e++) {
        const uint32_t w = adj_f[e];
        if (pk_sub(degi, w, 1u) == 1u && take(w)) { if (next == D2_NONE) next = w; else spill(w); }
      }
      for (uint32_t e = off_r[u]; e < (uint32_t)off_r[u + 1u]; e++) {
        const uint32_t p = adj_r[e];
        if (pk_sub(dego, p, 1u) == 1u && take(p)) { if (next == D2_NONE) next = p; else spill(p); }
      }
      if (next == D2_NONE) break;
      u = next;
    }
  };
  for (uint32_t v = tid; v < NV; v += NT)
    if (comp[v] == D2_ALIVE && (pk_get(degi, v) == 0u || pk_get(dego, v) == 0u) && take(v)) peel_from(v);
  uint32_t ncur = 0;
  uint32_t guard = 0;  // (every loop below ends after at most NV rounds by construction; a defect must not hold the GPU)
  const uint32_t guard_max = 8u * NV + 64u;
  for (;;) {
    for (;;) {  // what the lanes above could not follow themselves
      bsync<NT>();
      ncur = hdr[H_NEXT];
      bsync<NT>();
      if (ncur == 0u || ++guard > guard_max) break;
      { uint16_t* t = cur; cur = nxt; nxt = t; }
      if (tid == 0u) *n_next = 0u;
      bsync<NT>();
      for (uint32_t i = tid; i < ncur; i += NT) peel_from(cur[i]);
    }
    // what is left lies on cycles or between them: the component of the lowest live node
    if (tid == 0u) hdr[H_PIVOT] = 0xFFFFFFFFu;
    bsync<NT>();
    for (uint32_t v0 = wave * 64u; v0 < NV; v0 += NT) {
      const uint32_t v = v0 + (uint32_t)lane;
      const uint32_t mn = wave_min((v < NV && comp[v] == D2_ALIVE) ? v : 0xFFFFFFFFu);
      if (lane == 0 && mn != 0xFFFFFFFFu) atomicMin(&hdr[H_PIVOT], mn);
    }
    bsync<NT>();
    const uint32_t pivot = hdr[H_PIVOT];
    bsync<NT>();
    if (pivot == 0xFFFFFFFFu || guard > guard_max) break;
    // forward from the pivot over live nodes ...
    if (tid == 0u) { comp[pivot] = D2_FW; cur[0] = (uint16_t)pivot; *n_next = 0u; }
    ncur = 1u;
    bsync<NT>();
    while (ncur && ++guard <= guard_max) {
      for (uint32_t i = tid; i < ncur; i += NT) {
        const uint32_t u = cur[i];
        for (uint32_t e = off_f[u]; e < (uint32_t)off_f[u + 1u]; e++) {
          const uint32_t w = adj_f[e];
          if (atomicCAS(&comp[w], D2_ALIVE, D2_FW) == D2_ALIVE) nxt[atomicAdd(n_next, 1u)] = (uint16_t)w;
        }
      }
      bsync<NT>();
      ncur = hdr[H_NEXT];
      bsync<NT>();
      { uint16_t* t = cur; cur = nxt; nxt = t; }
      if (tid == 0u) *n_next = 0u;
      bsync<NT>();
    }
    // ... and backward from it over what the forward search reached: the intersection is the component
    if (tid == 0u) { comp[pivot] = D2_FWBW; cur[0] = (uint16_t)pivot; }
    ncur = 1u;
    bsync<NT>();
    while (ncur && ++guard <= guard_max) {
      for (uint32_t i = tid; i < ncur; i += NT) {
        const uint32_t u = cur[i];
        for (uint32_t e = off_r[u]; e < (uint32_t)off_r[u + 1u]; e++) {
          const uint32_t p = adj_r[e];
          if (atomicCAS(&comp[p], D2_FW, D2_FWBW) == D2_FW) nxt[atomicAdd(n_next, 1u)] = (uint16_t)p;
        }
      }
      bsync<NT>();
      ncur = hdr[H_NEXT];
      bsync<NT>();
      { uint16_t* t = cur; cur = nxt; nxt = t; }
      if (tid == 0u) *n_next = 0u;
      bsync<NT>();
    }
    // the component's nodes leave the live graph together; the others the forward search marked are live again
    if (tid == 0u) hdr[H_N2] = 0u;
    bsync<NT>();
    for (uint32_t v0 = wave * 64u; v0 < NV; v0 += NT) {
      const uint32_t v = v0 + (uint32_t)lane;
      const uint32_t st = v < NV ? comp[v] : 0u;
      if (st == D2_FW) comp[v] = D2_ALIVE;
      const bool mem = st == D2_FWBW;
      const uint64_t m = __ballot(mem);
      const uint32_t at = wave_reserve(&hdr[H_N2], m, lane);
      if (mem) { comp[v] = pivot; cur[at] = (uint16_t)v; }
    }
    bsync<NT>();
    const uint32_t nmem = hdr[H_N2];
    for (uint32_t i = tid; i < nmem; i += NT) peel_from(cur[i]);
  }
  bsync<NT>();
  if (guard > guard_max) return 2;
  if (plog && tid == 0u) plog[1] |= (unsigned long long)min(guard, 0xFFFFu) << 48;
  // a pass-through node belongs to the component both ends of its chain are in (the chain then lies on a cycle), and is
  // a component of its own otherwise — HEAD: the component its chain leaves
  for (uint32_t v = tid; v < NV; v += NT) {
    const uint32_t st = comp[v];
    if (!d2_is_pass(st)) continue;
    uint32_t c = v, head = v;
    if (st != D2_PASS) {  // (D2_PASS: a chain no node leads into — cannot be: every node is reached from the source)
      const uint32_t u = st & 0xFFFFu, k = (st >> 16) & 15u;
      const uint32_t w = adj_f[(uint32_t)off_f[u] + k];
      head = comp[u];  // (u is not a pass-through node: its state is final)
      if (head == comp[w]) c = head;
    }
    HEAD[v] = head;
    comp[v] = c;
  }
  bsync<NT>();
  D2_LAP(7);

  // ---- the components: size, which are non-trivial (several nodes, or a run covered in both directions), the
  // contracted multigraph's degrees (:1342-1378); then the statistics (:1404-1409)
  uint32_t* csize = degi;                         // [NV] words (the two degree tables are through)
  uint32_t* cdeg_in = (uint32_t*)adj_r;           // packed: edges into the component from others ...
  uint32_t* cdeg_out = cdeg_in + C::NV / 2u;      // ... and out of it (E * 2 bytes >= NV * 2 bytes: E >= 2 NV)
  uint32_t* cwork = (uint32_t*)off_r;             // packed: edges in not yet accounted for (Kahn)
  for (uint32_t v = tid; v < C::NV; v += NT) csize[v] = 0u;
  for (uint32_t i = tid; i < C::NV / 2u; i += NT) { cdeg_in[i] = 0u; cdeg_out[i] = 0u; }
  for (uint32_t i = tid; i < (C::NV + 2u) / 2u; i += NT) cwork[i] = 0u;
  if (tid == 0u) { hdr[H_R0] = 0u; hdr[H_R1] = 0u; acc64[0] = 0ull; acc64[1] = 0ull; }
  bsync<NT>();
  for (uint32_t v = tid; v < NV; v += NT) atomicAdd(&csize[comp[v]], 1u);
  bsync<NT>();
  auto is_pass_node = [&](uint32_t v) -> bool { return (PM[v >> 5] >> (v & 31u)) & 1u; };
  auto nontriv = [&](uint32_t c) -> bool { return (csize[c] & 0x3FFFFFFFu) > 1u || (NW[c] & 0xC0000000u) == 0xC0000000u; };
  // acc64[0]: edges of the contracted graph, acc64[1]: k-mers in non-trivial components; H_R0: non-trivial components,
  // H_R1: self loops outside them
  for (uint32_t i0 = wave * 64u; i0 < ne; i0 += NT) {
    const uint32_t i = i0 + (uint32_t)lane;
    const bool h = i < ne;
    const uint32_t e = h ? EDGE[i] : 0u;
    const uint32_t a = h ? comp[e >> 16] : 0u, b = h ? comp[e & 0xFFFFu] : 0u;
    const bool cross = h && a != b;
    if (cross) { pk_add(cdeg_out, a, 1u); pk_add(cdeg_in, b, 1u); }
    const uint32_t nc = (uint32_t)__popcll(__ballot(cross));
    if (lane == 0 && nc) atomicAdd(&acc64[0], (unsigned long long)nc);
  }
  // (what Kahn counts down: the edges into a component from other components with the chains skipped)
  for (uint32_t u = tid; u < NV; u += NT) {
    if (is_pass_node(u)) continue;
    const uint32_t a = comp[u];
    for (uint32_t e = off_f[u]; e < (uint32_t)off_f[u + 1u]; e++) { const uint32_t b = comp[adj_f[e]]; if (b != a) pk_add(cwork, b, 1u); }
  }
  for (uint32_t v0 = wave * 64u; v0 < NV; v0 += NT) {
    const uint32_t v = v0 + (uint32_t)lane;
    const bool h = v < NV;
    const uint32_t c = h ? comp[v] : 0u;
    const bool nt = h && nontriv(c);
    const uint32_t wgt = h ? (NW[v] & 0x3FFFFFFFu) : 0u, cov = h ? NW[v] >> 30 : 0u;
    const uint32_t n_nt = (uint32_t)__popcll(__ballot(nt && c == v));
    const uint32_t s_nt = wave_sum(nt ? wgt : 0u);
    // (a run outside the non-trivial components keeps its own edges in the contracted graph)
    const uint32_t own = wave_sum((h && !nt && wgt > 1u) ? ((cov & 1u) + (cov >> 1)) * (wgt - 1u) : 0u);
    if (lane == 0) {
      if (n_nt) atomicAdd(&hdr[H_R0], n_nt);
      if (s_nt) atomicAdd(&acc64[1], (unsigned long long)s_nt);
      if (own) atomicAdd(&acc64[0], (unsigned long long)own);
    }
  }
  for (uint32_t i0 = wave * 64u; i0 < nloops; i0 += NT) {
    const uint32_t i = i0 + (uint32_t)lane;
    const uint32_t lt = (uint32_t)__popcll(__ballot(i < nloops && !nontriv(comp[LOOPS[i]])));  // :1385-1402
    if (lane == 0 && lt) atomicAdd(&hdr[H_R1], lt);
  }
  bsync<NT>();
  const unsigned long long fe = acc64[0], size_nontrivial = acc64[1];
  const uint32_t nontrivial = hdr[H_R0], loops_trivial = hdr[H_R1];
  bsync<NT>();
  D2_LAP(8);

  // ---- the components (chains skipped) in a topological order: a component whose last edge in has been accounted
  // for takes the next place — the lane that accounted for it goes on with it at once; components of several nodes
  // wait for a pass of all threads over their nodes
  uint32_t* n_order = &hdr[H_ORDER];
  uint32_t* n_bigs = &hdr[H_BIGS];
  uint16_t* bigs_a = (uint16_t*)(hdr + H_BIGS_A);  // (a handful: two lists of 12 entries that take turns)
  uint16_t* bigs_b = (uint16_t*)(hdr + H_BIGS_B);
  uint16_t* bigs = bigs_a;
  cur = wl0; nxt = wl1;
  if (tid == 0u) { *n_next = 0u; *n_order = 0u; *n_bigs = 0u; }
  bsync<NT>();
  auto ready = [&](uint32_t b, uint32_t* next) {  // component b has no edge in left
    if ((csize[b] & 0x3FFFFFFFu) > 1u && !is_pass_node(b)) { const uint32_t at = atomicAdd(n_bigs, 1u); if (at < 12u) bigs[at] = (uint16_t)b; else spill(b); }
    else if (*next == D2_NONE) *next = b;
    else spill(b);
  };
  auto order_from = [&](uint32_t c) {  // c: a component of one node whose turn it is
    for (uint32_t steps = 0; steps <= NV; steps++) {
      ORDER[atomicAdd(n_order, 1u)] = c;
      uint32_t next = D2_NONE;
      for (uint32_t e = off_f[c]; e < (uint32_t)off_f[c + 1u]; e++) {
        const uint32_t b = comp[adj_f[e]];
        if (b != c && pk_sub(cwork, b, 1u) == 1u) ready(b, &next);
      }
      if (next == D2_NONE) break;
      c = next;
    }
  };
  // (the components without an edge in — the source's — go through the list: looked for before anything is counted down)
  for (uint32_t v = tid; v < NV; v += NT)
    if (comp[v] == v && !is_pass_node(v) && pk_get(cwork, v) == 0u) spill(v);
  for (guard = 0; guard <= guard_max; guard++) {
    bsync<NT>();
    const uint32_t nbig = min(hdr[H_BIGS], 12u);
    ncur = hdr[H_NEXT];
    bsync<NT>();
    if (nbig == 0u && ncur == 0u) break;
    { uint16_t* t = cur; cur = nxt; nxt = t; }
    const uint16_t* bigs_now = bigs;
    bigs = bigs == bigs_a ? bigs_b : bigs_a;  // (what becomes ready meanwhile is listed in the other half)
    if (tid == 0u) { *n_next = 0u; *n_bigs = 0u; }
    bsync<NT>();
    for (uint32_t x = 0; x < nbig; x++) {  // a component of several nodes: all threads over the nodes
      const uint32_t cc = bigs_now[x];
      if (tid == 0u) ORDER[atomicAdd(n_order, 1u)] = cc;
      for (uint32_t v = tid; v < NV; v += NT) {
        if (comp[v] != cc || is_pass_node(v)) continue;
        uint32_t next = D2_NONE;
        for (uint32_t e = off_f[v]; e < (uint32_t)off_f[v + 1u]; e++) {
          const uint32_t b = comp[adj_f[e]];
          if (b != cc && pk_sub(cwork, b, 1u) == 1u) ready(b, &next);
        }
        if (next != D2_NONE) order_from(next);
      }
    }
    for (uint32_t i = tid; i < ncur; i += NT) {
      const uint32_t c = cur[i];
      if ((csize[c] & 0x3FFFFFFFu) > 1u && !is_pass_node(c)) { const uint32_t at = atomicAdd(n_bigs, 1u); if (at < 12u) bigs[at] = (uint16_t)c; else spill(c); }
      else order_from(c);
    }
  }
  bsync<NT>();
  const uint32_t n_ord = hdr[H_ORDER];
  if (plog && tid == 0u) plog[2] |= (unsigned long long)min(guard, 0xFFFFu) << 48;
  {  // (every component outside the chains has its place: anything else is a defect — the gap is left to the host)
    if (tid == 0u) hdr[H_N2] = 0u;
    bsync<NT>();
    for (uint32_t v0 = wave * 64u; v0 < NV; v0 += NT) {
      const uint32_t v = v0 + (uint32_t)lane;
      const uint32_t nbr = (uint32_t)__popcll(__ballot(v < NV && comp[v] == v && !is_pass_node(v)));
      if (lane == 0 && nbr) atomicAdd(&hdr[H_N2], nbr);
    }
    bsync<NT>();
    if (hdr[H_N2] != n_ord || n_ord == 0u) { if (A.prof && tid == 0u) atomicAdd(A.prof + 10, 1000000ull); return 2; }
  }
  gsync<NT>();
  D2_LAP(9);
  // ---- the branch rule over that order (:1411-1434): -(in - 1) in front of a vertex, +(out - 1) behind it.  Bit 31
  // of a component's word: the count is 1 at it; bit 30: the count is 1 behind it (what the nodes of a chain that
  // leaves it see: one edge in, one out each, the count does not move along them)
  if (wave == 0u) {
    int bc = 1;
    for (uint32_t p0 = 0; p0 < n_ord; p0 += 64u) {
      const uint32_t p = p0 + (uint32_t)lane;
      const bool h = p < n_ord;
      const uint32_t c = h ? ORDER[p] : 0u;
      const int din = h ? (int)pk_get(cdeg_in, c) : 0, dout = h ? (int)pk_get(cdeg_out, c) : 0;
      const bool act = din >= 1 || dout >= 1;
      const int pre = (act && din > 1) ? -(din - 1) : 0, post = (act && dout > 1) ? dout - 1 : 0;
      const int incl = (int)wave_scan((uint32_t)(pre + post), lane);
      const int at = bc + incl - (pre + post) + pre;
      if (h) csize[c] |= ((act && at == 1) ? 0x80000000u : 0u) | ((at + post == 1) ? 0x40000000u : 0u);
      bc += (int)rl((uint32_t)incl, 63);
    }
  }
  bsync<NT>();
  auto verdict = [&](uint32_t v) -> uint32_t {  // branch[v] == 1, v outside the non-trivial components
    const uint32_t c = comp[v];
    if (nontriv(c)) return 0u;
    if (is_pass_node(v)) return (csize[HEAD[v]] >> 30) & 1u;
    return csize[c] >> 31;
  };
  const bool sink_safe = verdict(0u) != 0u;
  // ---- what leaves: the runs with their verdicts, the statistics
  if (tid == 0u) {
    const unsigned long long b0 = atomicAdd(A.run_cursor, (unsigned long long)R);
    hdr[H_BASE_LO] = (uint32_t)b0; hdr[H_BASE_HI] = (uint32_t)(b0 >> 32);
  }
  bsync<NT>();
  const unsigned long long base = ((unsigned long long)hdr[H_BASE_HI] << 32) | hdr[H_BASE_LO];
  if (base + R > A.run_cap) return 2;
  uint32_t* runs = A.runs + 2ull * base;
  for (uint32_t r = tid; r < R; r += NT) {
    runs[2u * r] = RLO[2u + r];
    runs[2u * r + 1u] = RHI[2u + r] | (verdict(2u + r) << 31);
  }
  if (tid == 0u) {
    D2Out o;
    o.run_off = (uint32_t)base; o.n_runs = R;
    o.sub[0] = V;
    o.sub[1] = (uint32_t)(e_all - loops_trivial);
    o.sub[2] = nontrivial;
    o.sub[3] = (uint32_t)size_nontrivial;
    o.sub[4] = (uint32_t)((unsigned long long)V + nontrivial - size_nontrivial);
    o.sub[5] = (uint32_t)fe;
    A.d2out[gap] = o;
    go->sub_vertices = V; go->sub_edges = o.sub[1];
  }
  d2_publish<NT>(go, (dflags0 & ~G2S_DEVA_SINK_SAFE) | G2S_DEVA_ANALYSED | G2S_DEVA_RUNS | (sink_safe ? G2S_DEVA_SINK_SAFE : 0u), A.behind != 0u);
  D2_LAP(10);
  if (A.prof && tid == 0u) { atomicAdd(A.prof + 11, 1ull); atomicAdd(A.prof + 12, (unsigned long long)NV); atomicAdd(A.prof + 13, (unsigned long long)guard); }
  return 0;
}

// the next gap of the list for this workgroup, or D2_NONE when there is none (and, polling, none can come any more)
__device__ __forceinline__ uint32_t d2_claim(const D2Args& A) {
  uint32_t gap = D2_NONE;
  if (!A.poll) {
    // (behind the fill kernels the list is complete: a ticket.  Not a compare-and-swap loop — a thousand workgroups that
    // start together retry half a million times on one word, and for milliseconds everything that touches the counters'
    // cache line waits behind them: tools/r05_c5_wgs.sh, LAB_NOTES)
    const unsigned long long cnt = __hip_atomic_load(A.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_load(A.next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= cnt) return D2_NONE;
    const unsigned long long at = atomicAdd(A.next, 1ull);
    if (at >= cnt || at >= (unsigned long long)A.list_cap) return D2_NONE;
    const uint32_t e = A.list[at];
    if (A.tag) A.list[at] = 0u;  // (tagged entries: the place is left clean for a polling launch of a later list)
    return A.tag ? (e & 0x00FFFFFFu) : e;
  }
  for (uint32_t spins = 0;;) {
    const unsigned long long cur = __hip_atomic_load(A.next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long cnt = __hip_atomic_load(A.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur < cnt && cur < (unsigned long long)A.list_cap) {
      if (atomicCAS(A.next, cur, cur + 1ull) != cur) continue;  // (another workgroup took it)
      uint32_t e = 0;
      if (A.tag) {  // (the entry follows its counter: wait for this list's tag, then leave the place clean)
        for (uint32_t w = 0; w < (1u << 24); w++) {
          e = __hip_atomic_load(&A.list[cur], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((e & 0xFF000000u) == A.tag) break;
          __builtin_amdgcn_s_sleep(4);
        }
        if ((e & 0xFF000000u) != A.tag) continue;  // (never expected: the gap stays pending and the list goes to the host path)
        __hip_atomic_store(&A.list[cur], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        e &= 0x00FFFFFFu;
      } else e = A.list[cur];
      gap = e;
      break;
    }
    if (!A.poll) break;
    // (a look at the list every few microseconds; at the 64 counters that say whether anything can still come only
    // every eighth time: 64 loads from the L2 that the fill kernel's own atomics go through)
    if ((spins & 7u) == 7u) {
      unsigned long long through = 0;
      for (int q = 0; q < 64; q++) through += __hip_atomic_load(A.done + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (through >= A.expected) {  // every gap of the fill launch is through: what is listed now is all there will be
        if (__hip_atomic_load(A.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >
            __hip_atomic_load(A.next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
        break;
      }
    }
    if (++spins > (1u << 17)) break;  // (half a second: a defect must not hold the GPU; what is left is taken by the launch behind the fill kernels)
    __builtin_amdgcn_s_sleep(127);
    __builtin_amdgcn_s_sleep(127);
  }
  return gap;
}

template <class C>
__device__ __forceinline__ void d2_loop(uint32_t* lds, const D2Args& A) {
  constexpr uint32_t NT = C::NT;
  uint32_t* scr = A.scratch + (size_t)blockIdx.x * C::SCR_WORDS;
  uint32_t* hdr_gap = lds + (C::LDS_BYTES - C::HDR_WORDS * 4u) / 4u + H_GAP;
  for (;;) {
    if (threadIdx.x == 0) *hdr_gap = d2_claim(A);
    bsync<NT>();
    const uint32_t gap = uni(*hdr_gap);
    bsync<NT>();
    if (gap == D2_NONE) break;
    // (an entry taken while the fill kernel runs: with the fill wave's fence in front of it, record and closure are here;
    // a launch behind the fill kernels sees everything anyway)
    if (A.tag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const unsigned long long t_in = A.prof ? __builtin_amdgcn_s_memrealtime() : 0ull;
    if (A.prof && A.log) {
      if (threadIdx.x == 0) {
        const uint32_t slot = (uint32_t)atomicAdd(A.prof + 21, 1ull);
        hdr_gap[H_SLOT - H_GAP] = slot;
        if (slot < A.log_cap) { unsigned long long* e = A.log + 16ull * slot; for (int q = 0; q < 16; q++) e[q] = 0ull; e[0] = gap | ((unsigned long long)A.outs[gap].n_xl << 32); e[15] = (C::NT == 64u ? 0ull : 1ull) | ((unsigned long long)(blockIdx.x & 0x7FFFu) << 1) | (t_in << 16); }
      }
      bsync<NT>();
    }
    const int rc = ((A.pass_all & 1u) && A.list_next) ? 1 : d2_one<C>(lds, A, gap, scr);
    bsync<NT>();
    if (A.prof && threadIdx.x == 0) {
      if (rc != 0) atomicAdd(A.prof + 13 + rc, 1ull);  // (14: beyond the capacities, 15: given up)
      const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t_in;
      atomicAdd(A.prof + 17, 1ull); atomicAdd(A.prof + 18, dt);
      if (atomicMax(A.prof + 16, dt) < dt) A.prof[20] = A.outs[gap].n_xl;
      if (A.log && hdr_gap[H_SLOT - H_GAP] < A.log_cap) { unsigned long long* e = A.log + 16ull * hdr_gap[H_SLOT - H_GAP]; e[2] |= (unsigned long long)(rc & 0xFFFF) << 32; e[14] = dt; }
    }
    if (rc == 1 && A.list_next && threadIdx.x == 0) {  // beyond these capacities: the larger instantiation's
      const unsigned long long at = atomicAdd(A.count_next, 1ull);
      if (at < (unsigned long long)A.list_cap) A.list_next[at] = gap;
    }
    // (rc == 2, or 1 where nobody takes the gap over: it stays without G2S_DEVA_ANALYSED — the host's, post.cpp — and says so)
    if (rc != 0 && !(rc == 1 && A.list_next)) d2_publish<NT>(A.outs + gap, uni(A.outs[gap].dflags) | G2S_DEVA_D2_FAILED, A.behind != 0u);
    gsync<NT>();
  }
  if (A.wgs_done && threadIdx.x == 0) { __threadfence(); atomicAdd(A.wgs_done, 1ull); }
}

__global__ __launch_bounds__(64) void g2s_d2_small(const D2Args A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  d2_loop<D2Small>(lds, A);
}
__global__ __launch_bounds__(256) void g2s_d2_small4(const D2Args A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  d2_loop<D2Small4>(lds, A);
}
__global__ __launch_bounds__(G2S_D2_BIG_NT) void g2s_d2_big(const D2Args A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  d2_loop<D2Big>(lds, A);
}

}  // namespace

namespace g2s {

size_t d2_scratch_bytes(bool big, uint32_t workgroups) {
  return (size_t)workgroups * (big ? D2Big::SCR_WORDS : D2Small::SCR_WORDS) * 4u;
}

hipError_t launch_d2(hipStream_t st, const D2Args& A0, uint32_t small_wgs, uint32_t big_wgs, uint32_t* scratch_small,
                     uint32_t* scratch_big, uint32_t* list_big, unsigned long long* count_big, unsigned long long* next_big) {
  hipError_t e = hipFuncSetAttribute((const void*)g2s_d2_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)D2Small::LDS_BYTES);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)g2s_d2_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)D2Big::LDS_BYTES);
  if (e != hipSuccess) return e;
  D2Args A = A0;
  A.poll = 0u;
  if (A.prof && A.log) A.log_cap /= 2u;
  A.scratch = scratch_small;
  A.list_next = big_wgs ? list_big : nullptr;
  A.count_next = count_big;
  // (four waves a closure — G2S_D2_SMALL_WAVES=1: one, as in round 5)
  static const bool four = !(getenv("G2S_D2_SMALL_WAVES") && atoi(getenv("G2S_D2_SMALL_WAVES")) == 1);
  if (four) {
    e = hipFuncSetAttribute((const void*)g2s_d2_small4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)D2Small4::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(g2s_d2_small4, dim3(std::max(1u, small_wgs)), dim3(D2Small4::NT), D2Small4::LDS_BYTES, st, A);
  } else
  hipLaunchKernelGGL(g2s_d2_small, dim3(std::max(1u, small_wgs)), dim3(D2Small::NT), D2Small::LDS_BYTES, st, A);
  if (big_wgs == 0u) return hipGetLastError();  // (what the small one cannot take is then the host's)
  D2Args B = A0;
  if (B.prof) B.prof += 32;  // (the log's cursor is each block's word 21: the large instantiation's entries lie behind the small one's capacity)
  if (B.prof && B.log) { B.log += 16ull * (B.log_cap / 2u); B.log_cap /= 2u; }
  B.tag = 0u; B.poll = 0u;  // (the entries the small instantiation passes on are plain)
  B.list = list_big;
  B.count = count_big;
  B.next = next_big;
  B.scratch = scratch_big;
  B.list_next = nullptr;
  B.count_next = nullptr;
  hipLaunchKernelGGL(g2s_d2_big, dim3(std::max(1u, big_wgs)), dim3(D2Big::NT), D2Big::LDS_BYTES, st, B);
  return hipGetLastError();
}

hipError_t launch_d2_poll(hipStream_t st, const D2Args& A0, uint32_t wgs, uint32_t* scratch_small, uint32_t* list_big,
                          unsigned long long* count_big) {
  hipError_t e = hipFuncSetAttribute((const void*)g2s_d2_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)D2Small::LDS_BYTES);
  if (e != hipSuccess) return e;
  D2Args A = A0;
  A.scratch = scratch_small;
  A.list_next = list_big;
  A.count_next = count_big;
  hipLaunchKernelGGL(g2s_d2_small, dim3(std::max(1u, wgs)), dim3(D2Small::NT), D2Small::LDS_BYTES, st, A);
  return hipGetLastError();
}

}  // namespace g2s
