// gap2seq_amd/csrc/fill_lds.hip — LDS-resident fast tier of the fill path for
// gfx950 (CDNA4): phases A, B, C and D1 of /root/reference/src/Gap2Seq.cpp:858-1312 in
// one kernel (g2s_fill_lds), one 64-lane wavefront per gap.
//
// Why a second tier: the first correct kernels (fill_kernels.hip) keep every
// per-gap table in HBM and pay 5-6 dependent L2/HBM round trips per DP level
// (~1.9 us/level measured, profiles/r01_v1_*).  A gap's working set is tiny —
// frontier width 1-4 on real graphs — so here everything a level touches lives in
// the CU's LDS (160 KB/CU on MI355X):
//   * frontier (node, count) ping-pong buffers;
//   * the right set (phase A's visited k-mers) as a bucketized LDS table;
//   * a per-level merge table keyed (depth, node) so that duplicate targets of one
//     level are combined (64-bit LDS compare-and-swap + wave ballot compaction);
//   * the target k-mers and the list of (target, depth, count) hits for phase C;
//   * phase D1's closure marks and log windows (the same LDS, after phase B).
// HBM sees successor-record loads (16 B per expansion), unitig-bitmap words and
// append-only, fire-and-forget stores of the state log and parent links; a gap's
// results go straight into pinned host memory when the gap is done.
//
// With a single wave per SIMD the depth loop is bound by instruction issue (~2 000
// cycles per level measured with s_memtime), so dependent steps are what every phase
// avoids.  Node ids are numbered along unitigs with a unitig-relative orientation bit
// (dbg.hpp): inside a unitig the successor of v is v+2 / v-2.
//   phase A  searches over unitigs, not levels: an event covers a whole unitig by
//            arithmetic and inserts it into the right set 64 nodes per instruction;
//   phase B  takes BULK STEPS: the unitig-start bitmap says for how many levels every
//            border state stays inside its unitig; lane i then holds the state of level
//            d+i by arithmetic and up to 64 levels are committed per iteration without
//            touching the graph (several parallel runs share the lanes level-major);
//            levels that branch, merge, die or leave a unitig take the per-level step;
//   phase D1 walks the parent links phase B recorded instead of the graph and sweeps
//            the bulk-produced levels 64 at a time.
// A right set or a state log that outgrows its share moves, inside the kernel, to a chunk
// of a per-launch pool in HBM (spill pool / log pool) and the gap goes on.  A gap that does
// not fit a pass otherwise (frontier, label table, target hits, host buffer) is flagged and
// run again by the next pass; the last resort is the HBM tier (also used for even k:
// explicit predecessor table).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "fill_device.h"
#include "fill_launch.h"

#define LDS_F 64u         /* frontier capacity of the first pass */
#define LDS_TG 32u        /* right_max_fuz + 1 must fit        */
#define LDS_TF 128u       /* target filter slots               */
#define LDS_TF_COLLIDE 0xFFFFFFFEu
#define LDS_CW 256u       /* candidates of 64 border entries   */
#ifndef G2S_BULK_MAX
#define G2S_BULK_MAX 64u  /* widest border the bulk step takes (runs share the 64 lanes level-major) */
#endif

namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint64_t lanes_below(int lane) { return (1ull << lane) - 1ull; }
// One wavefront per workgroup: LDS operations of a wave execute in issue order, so
// ordering LDS writes before later LDS reads of other lanes needs no s_barrier and,
// unlike __syncthreads(), must NOT wait for the outstanding global stores of the
// state log (vmcnt) — that wait alone costs an HBM round trip per level.
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t flip(uint32_t v) { return v == G2S_DEV_INVALID ? v : (v ^ 1u); }

// number of leading lanes (from lane 0) whose predicate holds
__device__ __forceinline__ uint32_t leading_true(bool p) {
  const uint64_t m = ~__ballot(p);
  return m ? (uint32_t)__builtin_ctzll(m) : 64u;
}

// Lanes are arranged level-major for a bulk step over R parallel runs: lane = i*Rp + r
// with Rp = R rounded up to a power of two.  Given the per-lane predicate (idle lanes
// r >= R must pass true) return how many leading levels have ALL runs true.
__device__ __forceinline__ uint32_t leading_levels(bool p, uint32_t lg) {
  uint64_t m = __ballot(p);
  const uint32_t rp = 1u << lg;
  for (uint32_t sh = 1; sh < rp; sh <<= 1) m &= (m >> sh);
  const uint64_t gmask = lg == 0 ? ~0ull : lg == 1 ? 0x5555555555555555ull : lg == 2 ? 0x1111111111111111ull
                         : lg == 3 ? 0x0101010101010101ull : lg == 4 ? 0x0001000100010001ull
                         : lg == 5 ? 0x0000000100000001ull : 1ull;
  const uint64_t bad = ~m & gmask;
  return bad ? ((uint32_t)__builtin_ctzll(bad) >> lg) : (64u >> lg);
}
// up to 64 runs share the lanes level-major.  (With verification loads a one-level bulk step
// on a wide border cost 6.8 k cycles against 2.4-5.5 k for the per-level step, so bulk steps
// stopped at 16 runs; driven by the bitmap they are the cheaper ones at any width.)
__device__ __forceinline__ uint32_t log2ceil16(uint32_t r) { return r <= 1 ? 0u : r <= 2 ? 1u : r <= 4 ? 2u : r <= 8 ? 3u : 4u; }
__device__ __forceinline__ uint32_t log2ceil64(uint32_t r) { return r <= 16 ? log2ceil16(r) : r <= 32 ? 5u : 6u; }

// Right set keyed by k-mer index: entry = index << 4 | expanded bits (3..2) | visited bits (1..0),
// one bit per strand.  The reference's set is keyed by k-mer as well (membership at :1050
// ignores the strand), so one probe answers "is either strand of this k-mer in the right
// set" and, on insert, whether the other strand is already there (Q7).  The expanded bit
// marks nodes reached with depth budget left (they were expanded by the reference's BFS):
// it only feeds the expansion counter.  G = false: table in LDS; G = true: table in HBM
// (gaps whose right set outgrows the LDS), re-read with agent-scope loads because it is
// mutated by L2 atomics.  Returns bit0 = orientation newly visited, bit1 = both strands now
// present, bit2 = table full, bit3 = orientation newly expanded.
#define RS_SHIFT 4u
__device__ __forceinline__ uint32_t rs_load(const bool G, const uint32_t* p) {
  return G ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
__device__ __forceinline__ uint32_t rs_result(uint32_t old, uint32_t bits) {
  const uint32_t now = old | bits;
  return ((bits & 3u & ~old) ? 1u : 0u) | (((now & 3u) == 3u) ? 2u : 0u) | ((bits & 12u & ~old) ? 8u : 0u);
}
// LDS flavour: buckets of 4 consecutive slots read with one ds_read_b128.  With linear
// probing one slot at a time the slowest of 64 lanes needed ~8 probes at load factor 0.5
// (measured: 1.4 k cycles per insert round); a 4-slot bucket is almost never full, so
// nearly every lane finishes with a single LDS read.
__device__ __forceinline__ uint32_t lrs_insert(const bool G, uint32_t* tab, uint32_t mask, uint32_t v, bool expanded) {
  const uint32_t idx = v >> 1;
  const uint32_t bits = (1u << (v & 1u)) | (expanded ? (4u << (v & 1u)) : 0u);
  if (!G) {
    const uint32_t bmask = mask >> 2;
    uint32_t b = mix32(idx) & bmask;
    for (uint32_t i = 0; i <= bmask; i++) {
      uint32_t* slot = tab + b * 4u;
      const uint4 c = *(const uint4*)slot;
      const uint32_t cs[4] = {c.x, c.y, c.z, c.w};
      int hit = -1, free_ = -1;
#pragma unroll
      for (int q = 3; q >= 0; q--) {
        if (cs[q] != G2S_DEV_INVALID && (cs[q] >> RS_SHIFT) == idx) hit = q;
        if (cs[q] == G2S_DEV_INVALID) free_ = q;
      }
      if (hit < 0 && free_ >= 0) {
        const uint32_t prev = atomicCAS(&slot[free_], G2S_DEV_INVALID, (idx << RS_SHIFT) | bits);
        if (prev == G2S_DEV_INVALID) return rs_result(0u, bits);
        if ((prev >> RS_SHIFT) == idx) hit = free_;
        else continue;  // somebody else took the slot: look at this bucket again
      }
      if (hit >= 0) return rs_result(atomicOr(&slot[hit], bits), bits);
      b = (b + 1) & bmask;  // bucket full of other k-mers
    }
    return 4u;
  }
  uint32_t h = mix32(idx) & mask;
  for (uint32_t i = 0; i <= mask; i++) {
    uint32_t cur = rs_load(G, &tab[h]);
    if (cur == G2S_DEV_INVALID) {
      cur = atomicCAS(&tab[h], G2S_DEV_INVALID, (idx << RS_SHIFT) | bits);
      if (cur == G2S_DEV_INVALID) return rs_result(0u, bits);
    }
    if ((cur >> RS_SHIFT) == idx) return rs_result(atomicOr(&tab[h], bits), bits);
    h = (h + 1) & mask;
  }
  return 4u;
}
__device__ __forceinline__ bool lrs_has_kmer(const bool G, const uint32_t* tab, uint32_t mask, uint32_t v) {
  const uint32_t idx = v >> 1;
  if (!G) {
    const uint32_t bmask = mask >> 2;
    uint32_t b = mix32(idx) & bmask;
    for (uint32_t i = 0; i <= bmask; i++) {
      const uint4 c = *(const uint4*)(tab + b * 4u);
      const uint32_t key = idx;
      if ((c.x != G2S_DEV_INVALID && (c.x >> RS_SHIFT) == key) || (c.y != G2S_DEV_INVALID && (c.y >> RS_SHIFT) == key) ||
          (c.z != G2S_DEV_INVALID && (c.z >> RS_SHIFT) == key) || (c.w != G2S_DEV_INVALID && (c.w >> RS_SHIFT) == key))
        return true;
      if (c.x == G2S_DEV_INVALID || c.y == G2S_DEV_INVALID || c.z == G2S_DEV_INVALID || c.w == G2S_DEV_INVALID)
        return false;  // inserts fill the first bucket with room along the probe sequence
      b = (b + 1) & bmask;
    }
    return false;
  }
  uint32_t h = mix32(idx) & mask;
  for (uint32_t i = 0; i <= mask; i++) {
    const uint32_t cur = rs_load(G, &tab[h]);
    if (cur == G2S_DEV_INVALID) return false;
    if ((cur >> RS_SHIFT) == idx) return true;
    h = (h + 1) & mask;
  }
  return false;
}

}  // namespace

// ============================================================================
// Phases A + B + C, LDS tier.
// dynamic LDS: [fa 2F][fn 2F][fc 2F][lh 2F x u64][lhslot 2F][th 3*TH][tgt TG][misc 4][tflt 128][cw 3x256][rs rs_cap]
// ============================================================================
// what phase D1 needs from phases A-C of the same gap (wave-uniform values)
struct FillOut {
  uint32_t flags, n_xl, top_level;
  int c_count, n_len, len0, len1, reached_j;
  // where the gap's state log, parent links and extra links ended up (its own slice of the
  // launch's arrays, or a chunk of the log pool) and where D1 keeps its scratch
  uint64_t* log;
  uint32_t* plk;
  uint64_t* xl;
  SubRec* sub;
  uint64_t* xo;
  uint32_t cap;
};

// A gap whose state log outgrows its slice (8 (D+2) states in the first pass) moves to a chunk
// of this pool and goes on, instead of running again in a later launch after every other gap
// is done.  Chunk layout: [log 8C][sub 16C][plk 4C][xl 8C/4][xo 8C/4] bytes, C = states.
struct LogPool {
  uint8_t* base;
  uint32_t chunks, states;
};
__device__ __forceinline__ size_t log_chunk_bytes(uint32_t c) { return (size_t)c * 32u; }

__device__ __forceinline__ FillOut fill_lds_body(const uint32_t* __restrict__ succ,
                                              const uint64_t* __restrict__ ustart, const GapDev* __restrict__ gaps,
                                              const uint32_t* __restrict__ gap_ids,
                                              const uint32_t* __restrict__ flank_nodes, uint64_t* log_all,
                                              uint32_t* lvl_all, uint32_t* plk_all, uint64_t* xl_all, GapOut* outs,
                                              uint32_t num_oriented, uint32_t* rs_global, const uint32_t F,
                                              unsigned long long* pool_cursor, uint32_t* rs_pool, uint32_t pool_chunks,
                                              uint32_t chunk_entries, const LogPool lp, SubRec* sub_scratch,
                                              uint64_t* xo_all) {
  const uint32_t LH = 2u * F;  // merge table slots (F = frontier capacity of this launch, a power of two)
  const uint32_t TH = 2u * F;  // target hits kept for phase C (128 in the first pass, 2048 later)
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t gi = __builtin_amdgcn_readfirstlane(gap_ids[blockIdx.x]);
  const GapDev gd = gaps[gi];
  const int lane = threadIdx.x;
  GapOut* go = &outs[gi];

  uint32_t* fa = lds;                             // phase A frontiers [2][F]
  uint32_t* fn = fa + 2 * F;                  // phase B frontier nodes [2][F]
  uint32_t* fc = fn + 2 * F;                  // phase B frontier counts [2][F]
  uint64_t* lh = (uint64_t*)(fc + 2 * F);     // level merge table keys
  uint32_t* lhslot = (uint32_t*)(lh + LH);
  uint32_t* th_j = lhslot + LH;               // (fn .. th_c are contiguous: phase A keeps its labels there)
  uint32_t* th_d = th_j + TH;
  uint32_t* th_c = th_d + TH;
  uint32_t* tgt = th_c + TH;
  uint32_t* misc = tgt + LDS_TG;              // [0] = number of target hits
  uint32_t* tflt = misc + 4;                  // exact target filter: 128 direct-mapped slots
  uint32_t* prem = tflt + LDS_TF;             // bulk steps: a lower bound of the levels each run still has inside its unitig
  uint32_t* cw_v = prem + G2S_BULK_MAX;       // wide levels: compacted candidate nodes / counts
  uint32_t* cw_c = cw_v + LDS_CW;
  uint32_t* cw_e = cw_c + LDS_CW;
  // right set: in LDS, or in HBM (host pre-filled 0xFF) — the gap's own table when the launch
  // keeps right sets in HBM, or a chunk of the launch's spill pool once the LDS share is outgrown
  bool rsg = rs_global != nullptr;
  uint32_t* rs = rsg ? rs_global + gd.rs_off : cw_e + LDS_CW;
  uint32_t rmask = gd.rs_mask;
  uint32_t rs_cap = rmask + 1u;  // capacity chosen by the host for this gap (LDS: <= rs_cap_max)
  const uint32_t* lseeds = flank_nodes + gd.flank_off;
  const uint32_t* rseeds = lseeds + (uint32_t)(gd.lmf + 1);
  const uint32_t* targets = rseeds + (uint32_t)(gd.rmf + 1);
  uint64_t* log = log_all + gd.slog_off;
  uint32_t* lvl = lvl_all + gd.lvl_off;
  uint32_t cap = gd.slog_cap;
  SubRec* sub = sub_scratch + gd.slog_off;
  uint64_t* xo = xo_all + gd.st_off;
  bool grown = false;
  // Parent links for phase D1 (g2s_extract_lds walks them instead of the graph): plk[s] = the
  // position, within the previous level, of the state that first produced state s; the
  // other parents of merged states go to the list xl as (state << 32 | position).
  uint32_t* plk = plk_all + gd.slog_off;
  uint64_t* xl = xl_all + gd.st_off;  // LDS tier: st_off / pad0 carry the extra-link list's offset / capacity
  uint32_t xcap = gd.pad0;
  uint32_t nxl = 0;

  if (!rsg) for (uint32_t i = (uint32_t)lane; i < rs_cap; i += 64u) rs[i] = G2S_DEV_INVALID;
  if (lane <= gd.rmf && lane < (int)LDS_TG) tgt[lane] = targets[lane];
  if (lane == 0) misc[0] = 0;
  // Is a state's node one of the <= 32 target k-mers?  Two filters in front of the exact scan
  // of tgt[]: a 64-bit Bloom word in registers, then a direct-mapped LDS table that is exact
  // except for slots two targets share.  (The scan used to run on most levels: with 11
  // targets the Bloom word alone passes 1 state in 6.)
  uint64_t tbloom = 0;  // which hash bits any target k-mer sets
  for (int j = 0; j <= gd.rmf; j++) {
    const uint32_t t = targets[j];
    if (t != G2S_DEV_INVALID) tbloom |= 1ull << (mix32(t) & 63u);
  }
  for (uint32_t i = (uint32_t)lane; i < LDS_TF; i += 64u) tflt[i] = G2S_DEV_INVALID;
  lds_sync();
  if (lane <= gd.rmf && lane < (int)LDS_TG) {
    const uint32_t t = targets[lane];
    if (t != G2S_DEV_INVALID) {
      uint32_t* slot = &tflt[(mix32(t) >> 6) & (LDS_TF - 1u)];
      const uint32_t prev = atomicCAS(slot, G2S_DEV_INVALID, t);
      if (prev != G2S_DEV_INVALID && prev != t) *slot = LDS_TF_COLLIDE;
    }
  }
  lds_sync();
  uint32_t nhit = 0;  // == misc[0], kept in a register
  // record (target index, depth, count) for every target k-mer equal to `node`
  auto note_targets = [&](bool active, uint32_t node, uint32_t depth, uint32_t c) {
    bool maybe = active && ((tbloom >> (mix32(node) & 63u)) & 1ull);
    if (__ballot(maybe) == 0) return;
    if (maybe) {
      const uint32_t x = tflt[(mix32(node) >> 6) & (LDS_TF - 1u)];
      maybe = x == node || x == LDS_TF_COLLIDE;
    }
    if (__ballot(maybe) == 0) return;
    if (maybe) {
      for (int j = 0; j <= gd.rmf; j++) {
        if (tgt[j] == node) {
          const uint32_t idx = atomicAdd(&misc[0], 1u);
          if (idx < TH) { th_j[idx] = (uint32_t)j; th_d[idx] = depth; th_c[idx] = c; }
        }
      }
    }
    lds_sync();
    nhit = misc[0];
  };

  uint32_t flags = 0;
  bool overflow = (gd.rmf + 1 > (int)LDS_TG);

  // ---------------- phase A: right BFS (Gap2Seq.cpp:871-982) -------------------
  // The reference's visited-set BFS computes {v : depth(v) <= right_half}, depth = fewest
  // predecessor steps from a right-flank seed (seed j starts at depth j); only membership is
  // consumed later (:1050).  A level-synchronous sweep would need right_half dependent
  // steps; instead the search runs over UNITIGS: an event (entry node, depth) covers the
  // entry's unitig backwards by arithmetic (ids -/+ 2, the unitig-start bitmap says how
  // far), inserts that run into the right set 64 nodes per instruction, and, where the
  // unitig ends with budget left, reads one successor record to propose the predecessor
  // unitigs.  Depth labels per entry node (LDS table, 64-bit atomic min) make this a
  // label-correcting search: an entry is processed again only if it was reached with a
  // smaller depth, so the final set is exactly the reference's.  The number of dependent
  // steps is the number of unitig hops (tens), not the number of levels (hundreds).
  uint32_t nvis = 0, xa = 0;
  uint32_t st_slowA = 0, st_bulkA = 0, st_slowB = 0, st_bulkB = 0;
  const unsigned long long cyc0 = __builtin_amdgcn_s_memtime();
  for (int attempt = 0; attempt < 2 && !overflow; attempt++) {
    uint64_t* lab = (uint64_t*)fn;        // labels (entry node << 32 | depth): the arrays of phase B are idle
    const uint32_t LAB = 8u * F;          // [fn 2F][fc 2F][lh 4F][lhslot 2F][th 6F] = 16F words
    for (uint32_t i = (uint32_t)lane; i < LAB; i += 64u) lab[i] = G2S_DEV_EMPTY64;
    lds_sync();
    uint32_t nlab = 0;
    // propose depth dp for entry node p; true when the label improved (the caller queues p)
    auto relabel = [&](bool active, uint32_t p, uint32_t dp) -> bool {
      bool improved = false, fresh = false;
      if (active) {
        const uint64_t key = ((uint64_t)p << 32) | dp;
        uint32_t h = mix32(p) & (LAB - 1u);
        while (true) {
          const uint64_t c = lab[h];
          if ((uint32_t)(c >> 32) == p) {
            improved = atomicMin((unsigned long long*)&lab[h], (unsigned long long)key) > key;
            break;
          }
          if (c == G2S_DEV_EMPTY64) {
            const unsigned long long prev =
                atomicCAS((unsigned long long*)&lab[h], (unsigned long long)G2S_DEV_EMPTY64, (unsigned long long)key);
            if (prev == G2S_DEV_EMPTY64) { improved = true; fresh = true; break; }
            continue;  // somebody took the slot: look at it again
          }
          h = (h + 1) & (LAB - 1u);
        }
      }
      nlab += (uint32_t)__popcll(__ballot(fresh));
      return improved;
    };
    auto label_of = [&](uint32_t p) -> uint32_t {
      uint32_t h = mix32(p) & (LAB - 1u);
      while (true) {
        const uint64_t c = lab[h];
        if ((uint32_t)(c >> 32) == p) return (uint32_t)c;
        h = (h + 1) & (LAB - 1u);
      }
    };
    uint32_t cur = 0, ne = 0;
    {  // seeds: right.substr(len-k-j, k) enters at depth j (:878-884, :953-976)
      const bool have = lane <= gd.rmf && lane < (int)LDS_TG;
      const uint32_t sd = have ? rseeds[lane] : G2S_DEV_INVALID;
      const bool imp = relabel(sd != G2S_DEV_INVALID && lane <= gd.right_half, sd, (uint32_t)lane);
      const uint64_t m = __ballot(imp);
      if (imp) fa[(uint32_t)__popcll(m & lanes_below(lane))] = sd;
      ne = (uint32_t)__popcll(m);
      lds_sync();
    }
    while (ne > 0 && !overflow) {
      uint32_t* fcur = fa + cur * F;
      uint32_t* fnxt = fa + (cur ^ 1u) * F;
      uint32_t nn = 0;
      st_bulkA++;
      for (uint32_t e0 = 0; e0 < ne && !overflow; e0 += 16u) {
        st_slowA++;
        const uint32_t G = min(16u, ne - e0), lg = log2ceil16(G), Rp = 1u << lg, LP = 64u >> lg;
        const uint32_t r = (uint32_t)lane & (Rp - 1u), jl = (uint32_t)lane >> lg;  // event r of the group, lane jl of it
        const bool mine = r < G;
        const uint32_t v = mine ? fcur[e0 + r] : 0u;
        const uint32_t d = mine ? label_of(v) : 0u;
        const uint32_t B = (uint32_t)gd.right_half - d;  // nodes at offsets t <= B have depth <= right_half
        const bool up = (v & 1u) != 0;                    // odd orientation: predecessors have larger ids
        const uint32_t idx = v >> 1;
        // ---- how far does the unitig go?  each of the event's LP lanes reads one bitmap word
        uint32_t dist = 0xFFFFFFFFu;  // offset of the unitig's last node in walking direction, if seen
        if (mine) {
          const int64_t w0 = (int64_t)(idx >> 6);
          if (!up) {
            uint64_t word = ustart[w0 - (int64_t)jl];
            if (jl == 0) word &= ~0ull >> (63u - (idx & 63u));
            if (word) dist = idx - (uint32_t)((w0 - (int64_t)jl) * 64 + (63 - __builtin_clzll(word)));
          } else {
            uint64_t word = ustart[w0 + (int64_t)jl];
            if (jl == 0) word = (idx & 63u) == 63u ? 0ull : word & (~0ull << ((idx & 63u) + 1u));
            if (word) dist = (uint32_t)((w0 + (int64_t)jl) * 64 + __builtin_ctzll(word)) - 1u - idx;
          }
        }
        const uint64_t runmask = (lg == 0 ? ~0ull : lg == 1 ? 0x5555555555555555ull : lg == 2 ? 0x1111111111111111ull
                                  : lg == 3 ? 0x0101010101010101ull : 0x0001000100010001ull) << r;
        const uint64_t fm = __ballot(dist != 0xFFFFFFFFu) & runmask;
        // the nearest boundary is the one seen by the event's lowest lane; else the window's edge
        uint32_t tmax;
        bool bounded = fm != 0;
        if (bounded) tmax = (uint32_t)__shfl((int)dist, __builtin_ctzll(fm));
        // (walking up, the bit that ends the unitig is the one AFTER its last node: stop one
        // short of the window's edge, so that the continuation node's own bit has been seen)
        else tmax = !up ? (idx & 63u) + 64u * (LP - 1u) : (63u - (idx & 63u)) + 64u * (LP - 1u) - 1u;
        const uint32_t L = mine ? min(tmax, B) + 1u : 0u;  // nodes of the run: offsets 0 .. L-1
        // ---- insert the runs of the group, 64 nodes per round, balanced over the lanes
        uint32_t incl = (jl == 0 && mine) ? L : 0u;  // inclusive prefix of L over the events, in lanes 0..Rp-1
        for (uint32_t o = 1; o < Rp; o <<= 1) {
          const uint32_t t = (uint32_t)__shfl_up((int)incl, (int)o);
          if ((uint32_t)lane >= o && (uint32_t)lane < Rp) incl += t;
        }
        const uint32_t total = (uint32_t)__shfl((int)incl, (int)(Rp - 1u));
        for (uint32_t k0 = 0; k0 < total; k0 += 64u) {
          const uint32_t k = k0 + (uint32_t)lane;
          uint32_t ev = 0;  // event whose run holds item k: the first with inclusive prefix > k
          for (uint32_t q = 0; q < G; q++) {
            const uint32_t iq = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)q);
            ev += (k >= iq) ? 1u : 0u;
          }
          if (ev >= G) ev = G - 1u;  // lanes past the end: any valid source lane
          // (cross-lane reads happen with every lane active: a disabled source lane reads as 0)
          const uint32_t startq = (uint32_t)__shfl((int)incl, (int)(ev ? ev - 1u : 0u));
          const uint32_t ve = (uint32_t)__shfl((int)v, (int)ev), de = (uint32_t)__shfl((int)d, (int)ev);
          uint32_t isnew = 0, isexp = 0;
          if (k < total) {
            const uint32_t t = k - (ev ? startq : 0u);
            const uint32_t node = (ve & 1u) ? ve + 2u * t : ve - 2u * t;
            const uint32_t rr = lrs_insert(rsg, rs, rmask, node, de + t < (uint32_t)gd.right_half);
            isnew = rr & 1u;
            isexp = (rr >> 3) & 1u;
            if (rr & 2u) flags |= G2S_DEV_Q7_A;
            if (rr & 4u) flags |= G2S_DEV_OVERFLOW_A | G2S_DEV_WHY_RS;
          }
          nvis += (uint32_t)__popcll(__ballot(isnew));
          xa += (uint32_t)__popcll(__ballot(isexp));
        }
        if (nvis > rs_cap / 4u * 3u) { overflow = true; flags |= G2S_DEV_WHY_RS; break; }
        // ---- where a run stopped with budget left: the predecessors of its last node
        // (unitig boundary: one record, 4 lanes), or the next node of the same unitig when
        // the bitmap window was too short to see the boundary
        {
          const uint32_t er = (uint32_t)lane >> 2, nt = (uint32_t)lane & 3u;  // event, slot
          const uint32_t ve = (uint32_t)__shfl((int)v, (int)er), de = (uint32_t)__shfl((int)d, (int)er);
          const uint32_t te = (uint32_t)__shfl((int)tmax, (int)er);
          const bool be = __shfl((int)bounded, (int)er) != 0;
          const bool live = er < G && de + te < (uint32_t)gd.right_half;  // the run's last node is expanded
          const uint32_t last = (ve & 1u) ? ve + 2u * te : ve - 2u * te;
          uint32_t p = G2S_DEV_INVALID;
          if (live && be) p = flip(succ[(size_t)(last ^ 1u) * 4 + nt]);       // graph.predecessors(last)[nt]
          else if (live && nt == 0) p = (ve & 1u) ? last + 2u : last - 2u;     // still inside the unitig
          const bool imp = relabel(p != G2S_DEV_INVALID, p, de + te + 1u);
          const uint64_t m = __ballot(imp);
          if (imp) {
            const uint32_t at = nn + (uint32_t)__popcll(m & lanes_below(lane));
            if (at < F) fnxt[at] = p;
          }
          nn += (uint32_t)__popcll(m);
        }
        if (nn > F) { overflow = true; flags |= G2S_DEV_WHY_FRONTIER; break; }
        if (nlab > LAB / 2u) { overflow = true; break; }
      }
      lds_sync();
      cur ^= 1u;
      ne = nn;
    }
    lds_sync();
    // A right set that outgrew its share of the LDS (8 % of C3's gaps at 1250+ gaps per GPU)
    // moves to a chunk of the launch's spill pool in HBM and the search starts over right
    // here, instead of in a second launch after every other gap is done.
    if (overflow && !rsg && pool_chunks && (flags & G2S_DEV_WHY_RS) && !(flags & G2S_DEV_WHY_FRONTIER) && nlab <= LAB / 2u) {
      unsigned long long chunk = 0;
      if (lane == 0) chunk = atomicAdd(pool_cursor, 1ull);
      chunk = __shfl(chunk, 0);
      if (chunk < pool_chunks) {
        rsg = true;
        rs = rs_pool + (size_t)chunk * chunk_entries;
        rmask = chunk_entries - 1u;
        rs_cap = chunk_entries;
        overflow = false;
        flags &= ~(G2S_DEV_OVERFLOW_A | G2S_DEV_WHY_RS | G2S_DEV_Q7_A);
        flags |= G2S_DEV_RS_POOL;
        nvis = 0; xa = 0;
        continue;
      }
    }
    // hand the arrays back to phase B
    for (uint32_t i = (uint32_t)lane; i < LH; i += 64u) lh[i] = G2S_DEV_EMPTY64;
    lds_sync();
    break;
  }
  if (overflow) flags |= G2S_DEV_OVERFLOW_A;

  const unsigned long long cyc1 = __builtin_amdgcn_s_memtime();
  // ---------------- phase B + C: left DP (Gap2Seq.cpp:984-1167) -----------------
  uint32_t nlog = 0, xb = 0;
  int lvl_top = -1;
  bool found = false;
  int c_count = 0, n_len = 0, len0 = 0, len1 = 0, reached_j = 0, final_d = 0;
  if (!overflow) {
    uint32_t cur = 0, nb = 0;
    {
      const uint32_t s0 = lseeds[0];  // leftmost k-mer: count 1 at depth 0 (:995-1015)
      if (s0 != G2S_DEV_INVALID) {
        if (lane == 0) { fn[0] = s0; fc[0] = 1; log[0] = ((uint64_t)s0 << 32) | 1ull; plk[0] = G2S_DEV_INVALID; }
        nb = 1;
        nlog = 1;
      }
      if (lane == 0) { lvl[0] = 0; lvl[1] = nlog; }
    }
    // the leftmost seed can itself be a target k-mer (tandem flanks): depth 0 hit
    if (lane == 0 && nb == 1) {
      for (int j = 0; j <= gd.rmf; j++)
        if (tgt[j] == fn[0]) { const uint32_t idx = misc[0]++; th_j[idx] = (uint32_t)j; th_d[idx] = 0; th_c[idx] = 1; }
    }
    lds_sync();
    nhit = misc[0];
    int d = 1, lvl_written = 1;  // lvl[0..lvl_written] hold valid offsets
    lvl_top = nb ? 0 : -1;
    uint32_t bulk_epoch = 0x80000000u;  // tags merge-table entries of bulk steps; never equals a depth
    bool rem_kept = false;              // prem[] is valid for the border as it stands (set by a bulk step)
    for (; d <= gd.D; d++) {
      // a level adds at most F states (and 64 from a bulk step): with less room than that left,
      // move the log, its links and the extra links to a chunk of the pool, once
      if ((nlog + max(F, 64u) > cap || nxl + max(F, 64u) > xcap) && !grown && lp.chunks && lp.states > cap) {
        unsigned long long chunk = 0;
        if (lane == 0) chunk = atomicAdd(pool_cursor + 1, 1ull);
        chunk = __shfl(chunk, 0);
        grown = true;  // (also when the pool is used up: do not ask again)
        if (chunk < lp.chunks) {
          uint8_t* cb = lp.base + chunk * log_chunk_bytes(lp.states);
          uint64_t* nlogp = (uint64_t*)cb;
          SubRec* nsubp = (SubRec*)(cb + (size_t)lp.states * 8u);
          uint32_t* nplk = (uint32_t*)(cb + (size_t)lp.states * 24u);
          uint64_t* nxlp = (uint64_t*)(cb + (size_t)lp.states * 28u);
          uint64_t* nxop = nxlp + lp.states / 4u;
          __threadfence();  // this wave's earlier stores are in L2 before they are read back
          for (uint32_t i = (uint32_t)lane; i < nlog; i += 64u) {
            nlogp[i] = __hip_atomic_load(&log[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nplk[i] = __hip_atomic_load(&plk[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          for (uint32_t i = (uint32_t)lane; i < nxl; i += 64u)
            nxlp[i] = __hip_atomic_load(&xl[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          log = nlogp; plk = nplk; xl = nxlp; sub = nsubp; xo = nxop;
          cap = lp.states;
          xcap = lp.states / 4u;
          flags |= G2S_DEV_LOG_POOL;
        }
      }
      uint32_t* ncur = fn + cur * F;
      uint32_t* ccur = fc + cur * F;
      uint32_t* nnxt = fn + (cur ^ 1u) * F;
      uint32_t* cnxt = fc + (cur ^ 1u) * F;
      const bool unpruned = d < gd.prune_from;  // :1050 first disjunct
      uint32_t nnew = 0;
      // ---- bulk step: every border state (<= 64 of them) sits inside a unitig, where the
      // only successor of id v is v+2 (even orientation) or v-2 (odd), see dbg.hpp.
      // Lanes are level-major: lane = i*Rp + r holds run r at level d+i, by arithmetic; the
      // unitig-start bitmap bounds i; the pruning rule is checked per lane; the leading
      // levels on which ALL runs hold are appended to the state log at once, and each
      // level's target check is evaluated by one lane.  Runs stay on distinct (node, depth)
      // diagonals inside unitigs, so no merging is needed; a run at the end of its unitig
      // (branching, dead end, merge) or a pruned state ends the bulk and the per-level code
      // below handles that level.
      if (nb >= 1 && nb <= G2S_BULK_MAX && d > gd.lmf) {
        const uint32_t R = nb, lg = log2ceil64(R), Rp = 1u << lg;
        const uint32_t r = (uint32_t)lane & (Rp - 1u), i = (uint32_t)lane >> lg;
        const bool mine = r < R;
        const uint32_t n = mine ? ncur[r] : 0u;
        uint32_t np = mine ? ccur[r] : 0u;
        if (np > G2S_DEV_MAX_PATHS) np = G2S_DEV_MAX_PATHS;
        const uint32_t step = 2u * (i + 1u);
        const bool up = (n & 1u) == 0;  // even orientation: successors have larger ids
        const uint32_t v = up ? n + step : n - step;  // state of run r at level d+i
        // How many more steps does every run stay inside its unitig?  Two words of the
        // unitig-start bitmap per run (lanes i = 0 and 1; the bitmap lives in L2) instead of
        // 64/Rp successor records from HBM, and nothing to verify afterwards: inside a
        // unitig the only successor of v is v +/- 2 and v is its only predecessor.
        // (64 = "no unitig start within the two words": at least 64 levels are left)
        uint32_t rem = 64u;
        auto rem_from = [&](uint32_t wi) -> uint32_t {  // from word wi (0: the run's own, 1: the next in walking direction)
          const uint32_t idx = n >> 1, b = idx & 63u;
          const int64_t w0 = (int64_t)(idx >> 6);
          if (up) {  // steps until the position before the next unitig start
            const uint64_t word = ustart[w0 + (int64_t)wi];
            const uint64_t m = wi == 1u ? word : (b == 63u ? 0ull : word & (~0ull << (b + 1u)));
            return m ? min(64u, (wi == 1u ? 64u : 0u) + (uint32_t)__builtin_ctzll(m) - 1u - b) : 64u;
          }
          // steps down to the unitig's own start
          const uint64_t word = ustart[w0 - (int64_t)wi];
          const uint64_t m = wi == 1u ? word : word & (~0ull >> (63u - b));
          return m ? min(64u, wi == 1u ? b + 1u + (uint32_t)__builtin_clzll(m) : b - (63u - (uint32_t)__builtin_clzll(m))) : 64u;
        };
        // The previous bulk step left a lower bound per run in LDS (what the bitmap said, minus
        // the levels taken since): no trip to the bitmap while every bound is positive.
        // min(smallest bound over the runs, cap_l) by bisection with ballots: a handful of
        // scalar steps where a shuffle reduction over 64 lanes is six LDS-crossbar round trips
        const uint32_t cap_l = min(64u >> lg, (uint32_t)(gd.D - d + 1));
        auto least = [&](uint32_t x) -> uint32_t {  // lanes without a run pass 64
          if (lg < 4u) {  // few runs: the shuffle reduction is the shorter one (lg steps against log2(cap_l) + 1)
            for (uint32_t o = 1; o < Rp; o <<= 1) x = min(x, (uint32_t)__shfl_xor((int)x, (int)o));
            return min((uint32_t)__builtin_amdgcn_readfirstlane((int)x), cap_l);
          }
          uint32_t lo_ = 0, hi_ = cap_l;
          while (lo_ < hi_) {
            const uint32_t mid = (lo_ + hi_ + 1u) >> 1;
            if (__ballot(x < mid)) hi_ = mid - 1u; else lo_ = mid;
          }
          return lo_;
        };
        uint32_t rem_own = 0, rem_all = 0;
        if (rem_kept) {
          rem_own = (mine && i == 0u) ? prem[r] : 64u;
          rem_all = least(rem_own);
        }
        if (rem_all == 0u) {
          if (Rp <= 32u) {
            if (mine && i < 2u) rem = rem_from(i);
            rem = min(rem, (uint32_t)__shfl_xor((int)rem, (int)Rp));  // the two words of a run (lanes i = 0 and 1)
          } else if (mine) {
            rem = min(rem_from(0u), rem_from(1u));  // 33..64 runs: one lane per run reads both words
          }
          rem_own = rem;
          rem_all = least(rem_own);
        }
        rem = rem_all;
        const uint32_t L = min(min(rem, 64u >> lg), (uint32_t)(gd.D - d + 1));
        bool ok = !mine;
        if (mine && i < L) ok = d + (int)i < gd.prune_from || lrs_has_kmer(rsg, rs, rmask, v);  // :1050
        uint32_t lrun = L ? leading_levels(ok, lg) : 0u;
        if (lrun > L) lrun = L;
        if (lrun >= 1u && nlog + lrun * R <= cap && nhit + 64u <= TH) {
          const bool act = mine && i < lrun;
          // (both strands of a k-mer are an even-orientation and an odd-orientation run of one
          // unitig walking towards each other: nothing to look for when all runs walk one way)
          const bool mixed = __ballot(mine && i == 0u && up) != 0ull && __ballot(mine && i == 0u && !up) != 0ull;
          if (act && R > 1 && mixed) {
            // Q7: the other strand of my k-mer on another run at this level.  All states of the
            // bulk are distinct, so claiming (epoch | k-mer index, level) in the merge table can
            // only collide with the other strand.  (Index truncated above 2^26 k-mers: the flag
            // is conservative, a false positive only narrows the bit-exact claim.)
            const uint64_t key = ((uint64_t)bulk_epoch << 32) | (uint32_t)(((v >> 1) << 6) | i);
            uint32_t h = mix32((uint32_t)key) & (LH - 1u);
            while (true) {
              const uint64_t c = lh[h];
              if ((uint32_t)(c >> 32) == bulk_epoch) {
                if (c == key) { flags |= G2S_DEV_Q7_B; break; }
                h = (h + 1) & (LH - 1u);
                continue;
              }
              if (atomicCAS((unsigned long long*)&lh[h], (unsigned long long)c, (unsigned long long)key) == c) break;
            }
          }
          bulk_epoch++;
          note_targets(act, v, (uint32_t)d + i, np);
          if (nhit > TH) { overflow = true; flags |= G2S_DEV_WHY_HITS; break; }
          // phase C for the levels of the run, one lane (r == 0) per level (:1107-1159)
          if (!found) {
            const int dl = d + (int)i;
            int bj = 1 << 30;
            uint32_t c1 = 0, c2 = 0;
            if (r == 0 && i < lrun && dl >= gd.g + gd.lmf + gd.rmf) {
              const int err = dl - gd.g - (gd.lmf + gd.rmf);
              const uint32_t nth = min(nhit, TH);
              for (uint32_t t = 0; t < nth; t++) {
                const int tj = (int)th_j[t], td = (int)th_d[t];
                const int l1 = gd.g + gd.lmf + tj + err, l2 = gd.g + gd.lmf + tj - err;
                const bool h1 = td == l1, h2 = err != 0 && l2 >= 0 && td == l2;
                if (!(h1 || h2) || tj > bj) continue;
                if (tj < bj) { bj = tj; c1 = 0; c2 = 0; }
                if (h1) c1 = th_c[t];
                if (h2) c2 = th_c[t];
              }
            }
            const uint64_t hm = __ballot(bj < (1 << 30));
            if (hm) {
              const int l0 = __builtin_ctzll(hm);  // lane of the first level with a hit
              const int i0 = l0 >> lg;
              bj = __shfl(bj, l0);
              c1 = (uint32_t)__shfl((int)c1, l0);
              c2 = (uint32_t)__shfl((int)c2, l0);
              const int err = d + i0 - gd.g - (gd.lmf + gd.rmf);
              const uint32_t sum = c1 + c2;
              c_count = (int)(sum > G2S_DEV_MAX_PATHS ? G2S_DEV_MAX_PATHS : sum);
              reached_j = bj;
              const int l1 = gd.g + gd.lmf + bj + err, l2 = gd.g + gd.lmf + bj - err;
              if (c1 > 0) { len0 = l1; n_len = 1; if (c2 > 0) { len1 = l2; n_len = 2; } }
              else { len0 = l2; n_len = 1; }
              found = true;
              if (!gd.all_paths) lrun = (uint32_t)i0 + 1u;  // -best-only: the DP stops after this level
            }
          }
          if (mine && i < lrun) {
            log[nlog + i * R + r] = ((uint64_t)v << 32) | np;
            plk[nlog + i * R + r] = r;  // run r of the level above, no other parent
            if (r == 0) lvl[d + (int)i + 1] = (nlog + (i + 1u) * R) | G2S_LVL_UNIFORM;
          }
          nlog += lrun * R;
          xb += lrun * R;
          lvl_written = d + (int)lrun;
          lvl_top = d + (int)lrun - 1;
          const uint32_t last = (uint32_t)__shfl((int)v, (int)(((lrun - 1u) << lg) + r));
          lds_sync();
          if (mine && i == 0) {
            ncur[r] = last;  // counts are unchanged along a run
            prem[r] = rem_own - lrun;
          }
          rem_kept = true;
          lds_sync();
          d += (int)lrun - 1;
          st_bulkB++;
          if (found && !gd.all_paths) break;  // (:1156-1158)
          continue;
        }
      }
      st_slowB++;
      rem_kept = false;  // the per-level step rebuilds the border
      xb += nb;
      if (nb == 1) {
        // single-entry frontier: its <=4 successors are distinct, no merging needed
        const uint32_t n = ncur[0];
        uint32_t np = ccur[0];
        if (np > G2S_DEV_MAX_PATHS) np = G2S_DEV_MAX_PATHS;
        const bool valid = lane < 4;
        const uint32_t v = valid ? succ[(size_t)n * 4 + ((uint32_t)lane & 3u)] : G2S_DEV_INVALID;
        const bool pass = v != G2S_DEV_INVALID && (unpruned || lrs_has_kmer(rsg, rs, rmask, v));
        const uint64_t m = __ballot(pass);
        if (pass) {
          const uint32_t off = (uint32_t)__popcll(m & lanes_below(lane));
          nnxt[off] = v;
          cnxt[off] = np;
          if (nlog + off < cap) plk[nlog + off] = 0u;
        }
        nnew = (uint32_t)__popcll(m);
        for (int q = 0; q < 4; q++) {  // Q7: two successors that are each other's reverse complement
          const uint32_t o = (uint32_t)__shfl((int)v, q);
          const bool op = (m >> q) & 1ull;
          if (pass && op && o == (v ^ 1u)) flags |= G2S_DEV_Q7_B;
        }
      } else if (nb > 1) {
        // Per-level step with merging.  One round takes up to 64 candidate states (node, count
        // of the expanded border state), one per lane:
        //   * claim (depth, node) in the merge table — stale entries of older levels count as
        //     free, so the table is never cleared; the hash is taken over the k-mer index, so
        //     both strands of a k-mer probe the same slots and Q7 is seen on the way;
        //   * the winner of a claim appends the node to the next border, every claimant adds
        //     its count (<= 4 predecessors x <= MAX_PATHS: the u32 sum cannot wrap, :1058-1060).
        auto merge_round = [&](uint32_t v, uint32_t np, uint32_t e) {  // e = position of the expanded border state
          const bool pass = v != G2S_DEV_INVALID && (unpruned || lrs_has_kmer(rsg, rs, rmask, v));  // :1050
          uint32_t h = 0, won = 0;
          if (pass) {
            const uint64_t key = ((uint64_t)(uint32_t)d << 32) | v;
            h = mix32(v >> 1) & (LH - 1u);
            while (true) {
              const uint64_t c = lh[h];
              if ((uint32_t)(c >> 32) == (uint32_t)d) {
                const uint32_t cv = (uint32_t)c;
                if (cv == v) break;                         // already claimed at this level
                if ((cv ^ v) == 1u) flags |= G2S_DEV_Q7_B;  // the other strand is a state of this level
                h = (h + 1) & (LH - 1u);
                continue;
              }
              const unsigned long long old =
                  atomicCAS((unsigned long long*)&lh[h], (unsigned long long)c, (unsigned long long)key);
              if (old == c) { won = 1; break; }             // else somebody changed the slot: look again
            }
          }
          const uint64_t m = __ballot(won);
          if (won) {
            const uint32_t off = nnew + (uint32_t)__popcll(m & lanes_below(lane));
            lhslot[h] = off < F ? off : 0u;
            if (off < F) {
              nnxt[off] = v;
              cnxt[off] = 0;
              if (nlog + off < cap) plk[nlog + off] = e;
            }
          }
          nnew += (uint32_t)__popcll(m);
          lds_sync();
          if (nnew > F) return;  // keeps the merge table (2F slots) from filling up: at most F + 64 claims
          uint32_t slot = 0;
          if (pass) { slot = lhslot[h]; atomicAdd(&cnxt[slot], np); }
          const uint64_t xm = __ballot(pass && !won);  // further parents of a merged state
          if (xm) {
            if (pass && !won) {
              const uint32_t at = nxl + (uint32_t)__popcll(xm & lanes_below(lane));
              if (at < xcap) xl[at] = ((uint64_t)(nlog + slot) << 32) | e;
            }
            nxl += (uint32_t)__popcll(xm);
          }
        };
        if (nb <= 16) {
          // narrow border: 4 lanes per entry, one per successor slot (one coalesced 16 B access)
          uint32_t v = G2S_DEV_INVALID, np = 0;
          if ((uint32_t)lane < nb * 4u) {
            v = succ[(size_t)ncur[(uint32_t)lane >> 2] * 4 + ((uint32_t)lane & 3u)];
            np = ccur[(uint32_t)lane >> 2];
          }
          if (np > G2S_DEV_MAX_PATHS) np = G2S_DEV_MAX_PATHS;
          merge_round(v, np, (uint32_t)lane >> 2);
        } else {
          // wide border: one entry per lane with its whole record in one load (a level costs
          // one HBM round trip whatever its width).  Nearly every entry has exactly one
          // successor: the first valid slot of every lane goes through a merge round straight
          // from the register; only the further slots (branching entries) are compacted
          // through LDS and merged 64 per round.
          for (uint32_t e0 = 0; e0 < nb && nnew <= F; e0 += 64u) {
            const uint32_t e = e0 + (uint32_t)lane;
            uint4 rec = make_uint4(G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID);
            uint32_t np = 0;
            if (e < nb) { rec = *(const uint4*)(succ + (size_t)ncur[e] * 4); np = ccur[e]; }
            if (np > G2S_DEV_MAX_PATHS) np = G2S_DEV_MAX_PATHS;
            const uint32_t fq = rec.x != G2S_DEV_INVALID ? 0u : rec.y != G2S_DEV_INVALID ? 1u : rec.z != G2S_DEV_INVALID ? 2u : 3u;
            const uint32_t first = fq == 0u ? rec.x : fq == 1u ? rec.y : fq == 2u ? rec.z : rec.w;
            merge_round(first, np, e);
            if (nnew > F) break;
            uint32_t ncand = 0;
#pragma unroll
            for (uint32_t q = 1; q < 4; q++) {
              const uint32_t v = q == 1 ? rec.y : q == 2 ? rec.z : rec.w;
              const bool more = v != G2S_DEV_INVALID && q > fq;
              const uint64_t m = __ballot(more);
              if (m) {
                if (more) {
                  const uint32_t at = ncand + (uint32_t)__popcll(m & lanes_below(lane));
                  cw_v[at] = v;
                  cw_c[at] = np;
                  cw_e[at] = e;
                }
                ncand += (uint32_t)__popcll(m);
              }
            }
            if (ncand) {
              lds_sync();
              for (uint32_t c0 = 0; c0 < ncand && nnew <= F; c0 += 64u) {
                const uint32_t c = c0 + (uint32_t)lane;
                merge_round(c < ncand ? cw_v[c] : G2S_DEV_INVALID, c < ncand ? cw_c[c] : 0u, c < ncand ? cw_e[c] : 0u);
              }
              lds_sync();
            }
          }
        }
      }
      if (nnew > F) { overflow = true; flags |= G2S_DEV_WHY_FRONTIER; break; }
      lds_sync();
      if (d <= gd.lmf) {  // next left-flank seed, row value ASSIGNED 1 (:1082-1105)
        const uint32_t s = lseeds[d];
        if (s != G2S_DEV_INVALID) {
          int at = -1;
          for (uint32_t e0 = 0; e0 < nnew; e0 += 64u) {
            const uint32_t e = e0 + (uint32_t)lane;
            if (__ballot(e < nnew && nnxt[e] == (s ^ 1u))) flags |= G2S_DEV_Q7_B;
            const uint64_t m = __ballot(e < nnew && nnxt[e] == s);
            if (m && at < 0) at = (int)e0 + __builtin_ctzll(m);
          }
          if (at >= 0) {
            if (lane == 0) cnxt[at] = 1;
          } else if (nnew < F) {
            if (lane == 0) { nnxt[nnew] = s; cnxt[nnew] = 1; if (nlog + nnew < cap) plk[nlog + nnew] = G2S_DEV_INVALID; }
            nnew++;
          } else { overflow = true; flags |= G2S_DEV_WHY_FRONTIER; break; }
          lds_sync();
        }
      }
      // append the level to the state log (fire and forget) and note target k-mers
      if (nlog + nnew > cap || nxl > xcap) { overflow = true; flags |= G2S_DEV_WHY_LOG; break; }
      for (uint32_t e0 = 0; e0 < nnew; e0 += 64u) {
        const uint32_t e = e0 + (uint32_t)lane;
        const bool have = e < nnew;
        const uint32_t node = have ? nnxt[e] : 0u;
        uint32_t c = have ? cnxt[e] : 0u;
        if (c > G2S_DEV_MAX_PATHS) c = G2S_DEV_MAX_PATHS;
        if (have) log[nlog + e] = ((uint64_t)node << 32) | c;
        note_targets(have, node, (uint32_t)d, c);
      }
      nlog += nnew;
      if (lane == 0) lvl[d + 1] = nlog;
      lvl_written = d + 1;
      if (nnew) lvl_top = d;
      lds_sync();
      if (nhit > TH) { overflow = true; flags |= G2S_DEV_WHY_HITS; break; }
      cur ^= 1u;
      nb = nnew;

      // ---- phase C: target check (:1107-1159) over the recorded hits ---------------
      if (!found && d >= gd.g + gd.lmf + gd.rmf) {
        const int err = d - gd.g - (gd.lmf + gd.rmf);
        const uint32_t nth = nhit;
        int bestj = 1 << 30;
        uint32_t c1 = 0, c2 = 0;
        for (uint32_t t0 = 0; t0 < nth; t0 += 64u) {
          const uint32_t t = t0 + (uint32_t)lane;
          int j = 1 << 30;
          uint32_t h1 = 0, h2 = 0;
          if (t < nth) {
            const int tj = (int)th_j[t], td = (int)th_d[t];
            const int l1 = gd.g + gd.lmf + tj + err, l2 = gd.g + gd.lmf + tj - err;
            if (td == l1) h1 = th_c[t];
            if (err != 0 && l2 >= 0 && td == l2) h2 = th_c[t];
            if (h1 | h2) j = tj;
          }
          if (__ballot(j < (1 << 30)) == 0ull) continue;  // no hit of an accepted length among these 64
          int jm = j;
          for (int o = 32; o > 0; o >>= 1) jm = min(jm, __shfl_xor(jm, o));
          if (jm < bestj) { bestj = jm; c1 = 0; c2 = 0; }
          if (jm == bestj && jm < (1 << 30)) {
            // (node, depth) states are unique: at most one lane holds each of the two lengths
            const uint64_t m1 = __ballot(j == bestj && h1), m2 = __ballot(j == bestj && h2);
            if (m1) c1 = (uint32_t)__shfl((int)h1, __builtin_ctzll(m1));
            if (m2) c2 = (uint32_t)__shfl((int)h2, __builtin_ctzll(m2));
          }
        }
        if (bestj < (1 << 30)) {
          const uint32_t sum = c1 + c2;
          c_count = (int)(sum > G2S_DEV_MAX_PATHS ? G2S_DEV_MAX_PATHS : sum);
          reached_j = bestj;
          const int l1 = gd.g + gd.lmf + bestj + err, l2 = gd.g + gd.lmf + bestj - err;
          if (c1 > 0) { len0 = l1; n_len = 1; if (c2 > 0) { len1 = l2; n_len = 2; } }
          else { len0 = l2; n_len = 1; }
          found = true;
        }
        if (found && !gd.all_paths) break;  // -best-only (:1156-1158)
      }
      // nothing left to expand, no seed to come and nothing more to find: the remaining
      // levels of the reference's loop are empty
      if (nb == 0 && d > gd.lmf && found) { d = gd.D + 1; break; }
    }
    final_d = d;
    // levels that were never reached are empty
    for (int dd = lvl_written + 1 + lane; dd <= gd.D + 1; dd += 64) lvl[dd] = nlog;
    if (overflow) flags |= G2S_DEV_OVERFLOW_B;
  }
  for (int o = 32; o > 0; o >>= 1) flags |= __shfl_xor(flags, o);
  if (lane == 0) {
    const unsigned long long cyc2 = __builtin_amdgcn_s_memtime();
    go->stat[0] = st_slowA; go->stat[1] = st_bulkA; go->stat[2] = st_slowB; go->stat[3] = st_bulkB;
    go->stat[4] = (uint32_t)((cyc1 - cyc0) >> 8); go->stat[5] = (uint32_t)((cyc2 - cyc1) >> 8);
    go->flags = flags;
    go->n_right = nvis;
    go->x_right = xa;
    go->n_states = nlog;
    go->x_left = xb;
    go->final_d = final_d;
    go->c_count = c_count;
    go->n_len = n_len;
    go->len[0] = len0;
    go->len[1] = len1;
    go->reached_j = reached_j;
    go->n_xl = nxl;
    go->top_level = (uint32_t)max(0, min(lvl_top, gd.D));
  }
  FillOut fo;
  fo.flags = flags; fo.n_xl = nxl; fo.top_level = (uint32_t)max(0, min(lvl_top, gd.D));
  fo.c_count = c_count; fo.n_len = n_len; fo.len0 = len0; fo.len1 = len1; fo.reached_j = reached_j;
  fo.log = log; fo.plk = plk; fo.xl = xl; fo.sub = sub; fo.xo = xo; fo.cap = cap;
  return fo;
}

// ============================================================================
// Phase D1 (Gap2Seq.cpp:1169-1312) plus the traceback's closure, LDS tier.
//
// The reference discovers the subgraph by walking graph predecessors backwards from
// the targets and looking every one of them up in the DP table.  Here phase B has
// already written, next to every state of the level-ordered log, the position of its
// parent in the level above (plk) and the list of further parents of merged states
// (xl); the backward sweep therefore touches neither the graph nor a hash table: one
// wave per gap walks the levels from D to 0, the closure marks of a level live in LDS,
// and every marked state is emitted once (depth descending) with the emit indices of
// its parents.  Runs of levels that phase B produced with a bulk step (flagged in
// lvl[]: same width, parent = same position, no merges) are swept up to 64 levels per
// iteration.  Parents are emitted in pred[0..] in arrival order, not GATB order; the
// traceback recovers the GATB order of a multi-parent state's parents from the graph.
// dynamic LDS: [wl W+1][wen W][wec W][wpl W][mk 2F][em F][ch 2x3F][pc F][xc 3x64]
// ============================================================================
#define LDS_XC 64u /* extra links handled per level */
__device__ __forceinline__ void extract_lds_body(const FillOut fo, const GapDev* __restrict__ gaps,
                                                 const uint32_t* __restrict__ gap_ids,
                                                 const uint32_t* __restrict__ flank_nodes,
                                                 const uint64_t* __restrict__ log_all,
                                                 const uint32_t* __restrict__ lvl_all,
                                                 const uint32_t* __restrict__ plk_all,
                                                 const uint64_t* __restrict__ xl_all, uint64_t* xo_all,
                                                 SubRec* sub_scratch, SubRec* sub_out, unsigned long long out_cap,
                                                 unsigned long long* out_counter, GapOut* outs, GapOut* outs_host,
                                                 uint32_t* done_list, int skip_confident, const uint32_t F) {
  const uint32_t W = F > 256u ? F : 256u;  // log / level-offset window: holds at least one whole level
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t gi = __builtin_amdgcn_readfirstlane(gap_ids[blockIdx.x]);
  const GapDev gd = gaps[gi];
  const int lane = threadIdx.x;
  GapOut* go = &outs[gi];
  const uint32_t gflags = fo.flags;
  const int c_count = fo.c_count, n_len = fo.n_len;
  // sub_out and outs_host are pinned host memory: the closure and the per-gap record go
  // straight over the link as each gap finishes, there is no device-to-host copy afterwards
  // ... and the gap's index is appended to done_list (pinned host memory too) once all of
  // that is on its way: the host analyses finished gaps while the others are still running.
  // (The cursor of the list lives in device memory next to the output cursor.)
  auto publish = [&]() {
    if ((uint32_t)lane < sizeof(GapOut) / 4u)
      ((uint32_t*)&outs_host[gi])[lane] = __hip_atomic_load(&((const uint32_t*)go)[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();  // every lane's stores to host memory are out before the list entry
    if (lane == 0) {
      const unsigned long long at = atomicAdd(out_counter + 1, 1ull);
      __hip_atomic_store(&done_list[at], gi, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  };
  if ((gflags & (G2S_DEV_OVERFLOW_A | G2S_DEV_OVERFLOW_B)) || !(c_count > 0 && n_len > 0)) {  // :1169
    publish();
    return;
  }

  uint32_t* wl = lds;                  // level offsets window [W+1] (bit 31: G2S_LVL_UNIFORM)
  uint32_t* wen = wl + (W + 1u);       // log window: nodes, counts, parent positions
  uint32_t* wec = wen + W;
  uint32_t* wpl = wec + W;
  uint32_t* mk = wpl + W;              // closure marks (IN_S | IN_T) by position: [0,F) this level, [F,2F) the one below
  uint32_t* em = mk + 2u * F;          // emit index by position, this level
  uint32_t* ch = em + F;               // closure states of a level: emit index, parent position, "more parents" [3][F]; ping-pong
  uint32_t* pc = ch + 6u * F;          // has further parents, by position (levels with merged states only)
  uint32_t* xc = pc + F;               // merged states of the level above: emit index, parent position, slot [3][XC]
  uint32_t msel = 0, csel = 0;

  (void)log_all; (void)plk_all; (void)xl_all; (void)xo_all; (void)sub_scratch;
  const uint64_t* log = fo.log;
  const uint32_t* lvl = lvl_all + gd.lvl_off;
  const uint32_t* plk = fo.plk;
  const uint64_t* xl = fo.xl;
  SubRec* sub = fo.sub;
  uint64_t* xo = fo.xo;  // the closure's side list: parents beyond the first, (state << 32 | parent)
  uint32_t nxo = 0;
  const uint32_t* lseeds = flank_nodes + gd.flank_off;
  const uint32_t* targets = lseeds + (uint32_t)(gd.lmf + 1) + (uint32_t)(gd.rmf + 1);
  const int len0 = fo.len0, len1 = fo.len1;
  const uint32_t reached = targets[fo.reached_j];
  const bool want_s = !skip_confident;
  const uint32_t sinknode = (want_s && gd.all_paths && gd.rmf >= 1) ? targets[gd.rmf - 1] : G2S_DEV_INVALID;
  const int lo_sink = max(0, gd.lmf + gd.g - gd.e);  // :1196
  const uint32_t t_flags = G2S_SUB_IN_T | G2S_SUB_START_T | ((want_s && !gd.all_paths) ? (G2S_SUB_IN_S | G2S_SUB_SINK) : 0u);
  const int min_len = n_len > 1 ? min(len0, len1) : len0;
  const uint32_t FL = G2S_SUB_IN_S | G2S_SUB_IN_T;
  auto own_flags = [&](uint32_t node, int depth) -> uint32_t {
    uint32_t f = 0;
    if (node == sinknode && depth >= lo_sink) f |= G2S_SUB_IN_S | G2S_SUB_SINK;              // :1195-1244
    if (node == reached && (depth == len0 || (n_len > 1 && depth == len1))) f |= t_flags;    // :1245-1259
    return f;
  };

  for (uint32_t i = (uint32_t)lane; i < 2u * F; i += 64u) mk[i] = 0;
  lds_sync();
  uint32_t st_slowD = 0, st_bulkD = 0;
  const unsigned long long cyc0 = __builtin_amdgcn_s_memtime();
  int wl_lo = gd.D + 2;          // wl[i] = lvl[wl_lo + i], i in [0, W]
  uint32_t we_lo = 0, we_hi = 0; // log positions [we_lo, we_hi) are in the entry window
  uint32_t nsub = 0, nch = 0, nxc = 0, xcount = 0, lflags = 0;
  uint32_t xpos = fo.n_xl;       // extra links [0, xpos) not yet consumed (sorted by state, ascending)
  uint32_t xtop = xpos ? (uint32_t)(xl[xpos - 1u] >> 32) : 0u;
  const uint32_t cap = fo.cap;
  bool over = false;

  // nothing can start above the last level that holds a state, nor (without a sink k-mer)
  // above the longest path length
  int d_top = min(gd.D, (int)fo.top_level);
  if (sinknode == G2S_DEV_INVALID) d_top = min(d_top, n_len > 1 ? max(len0, len1) : len0);
  if (xpos > 0) {  // extra links of the levels that are skipped
    const uint32_t hi_top = lvl[d_top + 1] & ~G2S_LVL_UNIFORM;
    while (xpos > 0) {
      const bool above = (uint32_t)lane < xpos && (uint32_t)(xl[xpos - 1u - (uint32_t)lane] >> 32) >= hi_top;
      const uint32_t cnt = leading_true(above);
      xpos -= cnt;
      if (cnt < 64u) break;
    }
    xtop = xpos ? (uint32_t)(xl[xpos - 1u] >> 32) : 0u;
  }
  for (int d2 = d_top; d2 >= 0; d2--) {
    if (nch == 0 && nxc == 0 && d2 < min_len && (sinknode == G2S_DEV_INVALID || d2 < lo_sink)) break;  // nothing can start below
    uint32_t* mcur = mk + msel * F;
    uint32_t* mnxt = mk + (msel ^ 1u) * F;
    // ---- level offsets through the LDS window -----------------------------------------
    if (d2 < wl_lo) {
      lds_sync();
      wl_lo = max(0, d2 + 1 - (int)W);
      for (uint32_t i = (uint32_t)lane; i <= W; i += 64u) {
        const int dd = wl_lo + (int)i;
        wl[i] = dd <= gd.D + 1 ? lvl[dd] : 0x7FFFFFFFu;
      }
      lds_sync();
    }
    const uint32_t lo = wl[d2 - wl_lo] & ~G2S_LVL_UNIFORM, hi_raw = wl[d2 + 1 - wl_lo];
    const uint32_t hi = hi_raw & ~G2S_LVL_UNIFORM;
    const uint32_t w = hi - lo;  // <= F by construction of the log
    if (w == 0) {                // empty level: nothing above can have a parent here
      nch = 0; nxc = 0;
      continue;
    }
    // ---- bulk step: this level and the K-1 below it were produced by bulk steps of phase B
    // (lvl flag): same width R, state (level, r) has the single parent (level-1, r).  Lane
    // i*Rp + r takes state r of level d2-i; closure membership flows down each run.
    if ((hi_raw & G2S_LVL_UNIFORM) && w <= 32u && d2 > gd.lmf + 1 && nxc == 0) {  // (33+ wide: one level per iteration, no gain)
      const uint32_t R = w, lg = log2ceil64(R), Rp = 1u << lg;
      const uint32_t r = (uint32_t)lane & (Rp - 1u), i = (uint32_t)lane >> lg;
      const bool mine = r < R;
      const int li = d2 - (int)i;
      // levels d2-i for i < K must all carry the flag and stay above the left flank and inside the window
      const bool lvl_ok = li > gd.lmf + 1 && li >= wl_lo && (wl[li + 1 - wl_lo] & G2S_LVL_UNIFORM) != 0;
      uint32_t K = leading_levels(lvl_ok || !mine, lg);
      K = min(K, 64u >> lg);
      if (K >= 2 && nsub + K * R <= cap) {
        const bool act = mine && i < K;
        // the K levels are contiguous in the log (uniform levels of one width): [hi - K*R, hi)
        if (hi - K * R < we_lo || hi > we_hi) {
          lds_sync();
          we_hi = hi;
          we_lo = hi > W ? hi - W : 0u;
          for (uint32_t q = (uint32_t)lane; q < we_hi - we_lo; q += 64u) {
            const uint64_t e = log[we_lo + q];
            wen[q] = (uint32_t)(e >> 32);
            wec[q] = (uint32_t)e;
            wpl[q] = plk[we_lo + q];
          }
          lds_sync();
        }
        const uint32_t wi = act ? (lo - i * R + r) - we_lo : 0u;
        const uint32_t node = act ? wen[wi] : 0u;
        const uint32_t own = act ? own_flags(node, li) : 0u;
        const uint32_t seed = own | ((act && i == 0) ? mcur[r] : 0u);
        // closure membership flows down each run: lane (i, r) has a flag when any lane (i' <= i, r) seeds it
        const uint64_t runmask = (lg == 0 ? ~0ull : lg == 1 ? 0x5555555555555555ull : lg == 2 ? 0x1111111111111111ull
                                  : lg == 3 ? 0x0101010101010101ull : lg == 4 ? 0x0001000100010001ull
                                  : 0x0000000100000001ull) << r;
        const uint64_t upto = lanes_below(lane) | (1ull << lane);
        const uint64_t ms = __ballot((seed & G2S_SUB_IN_S) != 0), mt = __ballot((seed & G2S_SUB_IN_T) != 0);
        const uint32_t prop = ((ms & runmask & upto) ? G2S_SUB_IN_S : 0u) | ((mt & runmask & upto) ? G2S_SUB_IN_T : 0u);
        const uint32_t f = act ? (own | prop) : 0u;
        const bool in = f != 0;
        const uint64_t m = __ballot(in);
        const uint32_t slot = nsub + (uint32_t)__popcll(m & lanes_below(lane));
        // the parent (i+1, r) is marked whenever (i, r) is; it is emitted by lane + Rp
        const uint32_t pslot = lane + (int)Rp < 64 ? nsub + (uint32_t)__popcll(m & lanes_below(lane + (int)Rp)) : 0u;
        if (in) {
          SubRec st;
          st.node = node; st.cnt = wec[wi]; st.meta = (uint32_t)li | (f << G2S_SUB_META_FLAG_SHIFT);
          st.pred = (i + 1u < K) ? (int32_t)pslot : -1;
          sub[slot] = st;
        }
        if (mine && i == 0) em[r] = slot;  // for the links from the level above
        lds_sync();
        uint32_t* chc = ch + csel * 3u * F;
        for (uint32_t j = (uint32_t)lane; j < nch; j += 64u) sub[chc[j]].pred = (int32_t)em[chc[F + j]];
        lds_sync();
        // the last level of the stretch becomes "the level above" of the next iteration
        const uint64_t lm = __ballot(in && i == K - 1u);
        if (in && i == K - 1u) {
          const uint32_t at = (uint32_t)__popcll(lm & lanes_below(lane));
          chc[at] = slot;
          chc[F + at] = r;
          chc[2u * F + at] = 0u;
        }
        nch = (uint32_t)__popcll(lm);
        // the same buffer takes the marks of the level below the stretch (msel does not move)
        if (mine && i == 0) mcur[r] = 0;
        lds_sync();
        if (in && i == K - 1u) mcur[r] = f & FL;
        lds_sync();
        const uint32_t nin = (uint32_t)__popcll(m);
        xcount += nin;
        nsub += nin;
        d2 -= (int)K - 1;
        st_bulkD++;
        continue;
      }
    }
    st_slowD++;
    // ---- the common per-level step, straight-line: the level fits one pass of the wave, is
    // in the LDS window, lies above the left flank (no sources) and neither it nor the level
    // above has merged states.  Same work as the general code below, a third of the
    // instructions (a lone wave pays ~5 cycles per instruction and ~64 per LDS round trip).
    if (w <= 64u && nxc == 0 && d2 > gd.lmf && lo >= we_lo && hi <= we_hi && !(xtop >= lo && xpos > 0)) {
      const uint32_t c = (uint32_t)lane;
      const bool have = c < w;
      const uint32_t wi = lo - we_lo + c;
      const uint32_t cn = have ? wen[wi] : 0u, cc = have ? wec[wi] : 0u;
      const uint32_t pp = have ? wpl[wi] : G2S_DEV_INVALID;
      const uint32_t cf = have ? (mcur[c] | own_flags(cn, d2)) : 0u;
      const bool in = cf != 0;
      const uint64_t m = __ballot(in);
      const uint32_t nin = (uint32_t)__popcll(m);
      if (nsub + nin > cap) { over = true; break; }
      const uint32_t slot = nsub + (uint32_t)__popcll(m & lanes_below(lane));
      const bool expand = in && pp != G2S_DEV_INVALID;  // d2 > lmf >= 0: neither a source nor depth 0
      const uint64_t lm = __ballot(expand);
      uint32_t* chc = ch + csel * 3u * F;
      uint32_t* chn = ch + (csel ^ 1u) * 3u * F;
      if (in) {
        SubRec st;
        st.node = cn; st.cnt = cc; st.meta = (uint32_t)d2 | (cf << G2S_SUB_META_FLAG_SHIFT);
        st.pred = -1;
        sub[slot] = st;
        em[c] = slot;
      }
      if (expand) {
        atomicOr(&mnxt[pp], cf & FL);
        const uint32_t at = (uint32_t)__popcll(lm & lanes_below(lane));
        chn[at] = slot;
        chn[F + at] = pp;
        chn[2u * F + at] = 0u;
      }
      if (have) mcur[c] = 0;  // recycled for the level after next
      lds_sync();
      for (uint32_t j = (uint32_t)lane; j < nch; j += 64u) sub[chc[j]].pred = (int32_t)em[chc[F + j]];
      lds_sync();
      nch = (uint32_t)__popcll(lm);
      csel ^= 1u;
      msel ^= 1u;
      xcount += nin;
      nsub += nin;
      continue;
    }
    // ---- entries of the level through the LDS window ----------------------------------
    if (lo < we_lo || hi > we_hi) {
      lds_sync();
      we_hi = hi;
      we_lo = hi > W ? hi - W : 0u;
      for (uint32_t i = (uint32_t)lane; i < we_hi - we_lo; i += 64u) {
        const uint64_t e = log[we_lo + i];
        wen[i] = (uint32_t)(e >> 32);
        wec[i] = (uint32_t)e;
        wpl[i] = plk[we_lo + i];
      }
      lds_sync();
    }
    const uint32_t wbase = lo - we_lo;
    // ---- merged states of this level: their further parents (rare) ----------------------
    uint32_t xfirst = xpos;
    if (xtop >= lo && xpos > 0) {  // xtop = state of the last unconsumed extra link
      while (xfirst > 0 && (uint32_t)(xl[xfirst - 1u] >> 32) >= lo) xfirst--;
      xtop = xfirst > 0 ? (uint32_t)(xl[xfirst - 1u] >> 32) : 0u;
    }
    const uint32_t nx = xpos - xfirst;  // extra links of states in [lo, hi)
    if (nx > LDS_XC) { over = true; break; }
    if (nx) {  // which positions of this level have further parents
      for (uint32_t c = (uint32_t)lane; c < w; c += 64u) pc[c] = 0u;
      lds_sync();
      if ((uint32_t)lane < nx) pc[(uint32_t)(xl[xfirst + (uint32_t)lane] >> 32) - lo] = 1u;
      lds_sync();
    }
    // ---- per-level step: emit the marked states, pass the marks on to their parents -------
    const uint32_t lidx = (d2 <= gd.lmf) ? (lseeds[d2] >> 1) : 0xFFFFFFFFu;  // buildNode(kmer_left.substr(d2,k))
    uint32_t* chc = ch + csel * 3u * F;           // closure states of the level above (emit index, parent position, more)
    uint32_t* chn = ch + (csel ^ 1u) * 3u * F;    // the same for this level, built here
    uint32_t nin_total = 0, nlink = 0;
    for (uint32_t c0 = 0; c0 < w; c0 += 64u) {
      const uint32_t c = c0 + (uint32_t)lane;
      const uint32_t cn = c < w ? wen[wbase + c] : 0u;
      uint32_t cf = c < w ? (mcur[c] | own_flags(cn, d2)) : 0u;
      const bool in = cf != 0;
      const uint64_t m = __ballot(in);
      const uint32_t nin = (uint32_t)__popcll(m);
      if (nsub + nin_total + nin > cap) { over = true; break; }
      if (in && d2 <= gd.lmf && (cn >> 1) == lidx) cf |= G2S_SUB_SOURCE;  // :1270 end condition, k-mer comparison only
      const uint32_t slot = nsub + nin_total + (uint32_t)__popcll(m & lanes_below(lane));
      const uint32_t pp = in ? wpl[wbase + c] : G2S_DEV_INVALID;
      // the parent joins the closure (:1266-1301); sources are not expanded
      const bool expand = in && !(cf & G2S_SUB_SOURCE) && d2 > 0 && pp != G2S_DEV_INVALID;
      const uint64_t lm = __ballot(expand);
      if (in) {
        SubRec st;
        st.node = cn; st.cnt = wec[wbase + c]; st.meta = (uint32_t)d2 | (cf << G2S_SUB_META_FLAG_SHIFT);
        st.pred = -1;
        sub[slot] = st;
        em[c] = slot;
      }
      if (expand) {
        atomicOr(&mnxt[pp], cf & FL);
        const uint32_t at = nlink + (uint32_t)__popcll(lm & lanes_below(lane));
        chn[at] = slot;
        chn[F + at] = pp;
        chn[2u * F + at] = nx ? pc[c] : 0u;
      }
      nin_total += nin;
      nlink += (uint32_t)__popcll(lm);
    }
    if (over) break;
    lds_sync();
    // ---- links from the level above to this level ----------------------------------------
    // (a state with a side-list entry always has a first parent: its plk link is in chc too)
    for (uint32_t j = (uint32_t)lane; j < nch; j += 64u) {
      const uint32_t more = chc[2u * F + j] ? (uint32_t)G2S_SUB_MORE : 0u;
      sub[chc[j]].pred = (int32_t)(em[chc[F + j]] | more);
    }
    for (uint32_t j = (uint32_t)lane; j < nxc; j += 64u) xo[nxo + j] = ((uint64_t)xc[j] << 32) | em[xc[LDS_XC + j]];
    nxo += nxc;
    lds_sync();
    nch = nlink;
    csel ^= 1u;
    // ---- further parents of merged states of this level -----------------------------------
    nxc = 0;
    if (nx) {
      const uint32_t j = (uint32_t)lane;
      bool link = false;
      uint32_t c = 0, pp = 0;
      if (j < nx) {
        const uint64_t e = xl[xfirst + j];
        c = (uint32_t)(e >> 32) - lo;
        pp = (uint32_t)e;
        const uint32_t cn = wen[wbase + c];
        const uint32_t f = mcur[c] | own_flags(cn, d2);
        const bool src = d2 <= gd.lmf && (cn >> 1) == lidx;
        link = f != 0 && !src && d2 > 0;
        if (link) atomicOr(&mnxt[pp], f & FL);
      }
      const uint64_t m = __ballot(link);
      if (link) {
        const uint32_t at = (uint32_t)__popcll(m & lanes_below(lane));
        xc[at] = em[c];
        xc[LDS_XC + at] = pp;
      }
      nxc = (uint32_t)__popcll(m);
      xpos = xfirst;
    }
    lds_sync();
    for (uint32_t c = (uint32_t)lane; c < w; c += 64u) mcur[c] = 0;  // recycled for the level after next
    lds_sync();
    msel ^= 1u;
    xcount += nin_total;
    nsub += nin_total;
  }
  if (over) lflags |= G2S_DEV_OVERFLOW_B;
  for (int o = 32; o > 0; o >>= 1) lflags |= __shfl_xor(lflags, o);
  if (lflags & G2S_DEV_OVERFLOW_B) {
    if (lane == 0) go->flags = gflags | lflags;
    __threadfence();
    publish();
    return;
  }
  // ---- pack: reserve exactly what the closure takes in the host buffer: n_sub records, then
  // the side list (two 8-byte entries per record slot).  The fence makes this wave's own
  // stores, including the 4-byte link patches, visible to its loads: they were written
  // through to L2, the L1 copies are dropped.
  __threadfence();
  const uint32_t nres = nsub + (nxo + 1u) / 2u;
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(out_counter, (unsigned long long)nres);
  base = __shfl(base, 0);
  if (base + nres > out_cap) {  // the host buffer is full: the gap is run again with the next pass
    if (lane == 0) go->flags = gflags | lflags | G2S_DEV_OVERFLOW_B | G2S_DEV_WHY_LOG;
    __threadfence();
    publish();
    return;
  }
  {
    const uint4* src = (const uint4*)sub;
    uint4* dst = (uint4*)(sub_out + base);
    for (uint32_t i0 = 0; i0 < nsub; i0 += 256u) {
      uint4 v[4];
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {  // four independent loads in flight per lane
        const uint32_t i = i0 + q * 64u + (uint32_t)lane;
        if (i < nsub) v[q] = src[i];
      }
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {
        const uint32_t i = i0 + q * 64u + (uint32_t)lane;
        if (i < nsub) dst[i] = v[q];
      }
    }
    uint64_t* xdst = (uint64_t*)(sub_out + base + nsub);
    for (uint32_t i = (uint32_t)lane; i < nxo; i += 64u) xdst[i] = xo[i];
  }
  if (lane == 0) {
    go->flags = gflags | lflags;
    go->n_sub = nsub;
    go->n_xp = nxo;
    go->sub_off = base;
    go->x_sub = xcount;
    go->stat[6] = st_slowD | (st_bulkD << 16);
    go->stat[7] = (uint32_t)((__builtin_amdgcn_s_memtime() - cyc0) >> 8);
  }
  __threadfence();
  publish();
}

// ============================================================================
// One wave per gap, phases A-D1 back to back: a gap's closure leaves for the host as soon
// as that gap is done, whatever the other gaps of the launch are doing.
// ============================================================================
#define G2S_FUSED_PARAMS                                                                                              \
  const uint32_t *__restrict__ succ, const uint64_t *__restrict__ ustart, const GapDev *__restrict__ gaps,            \
      const uint32_t *__restrict__ gap_ids, const uint32_t *__restrict__ flank_nodes, uint64_t *log_all,              \
      uint32_t *lvl_all, uint32_t *plk_all, uint64_t *xl_all, uint64_t *xo_all, SubRec *sub_scratch, SubRec *sub_out, \
      unsigned long long out_cap, unsigned long long *out_counter, GapOut *outs, GapOut *outs_host,                   \
      uint32_t *done_list, int skip_confident, uint32_t num_oriented, LogPool lp
__global__ __launch_bounds__(64) void g2s_fill_lds(G2S_FUSED_PARAMS, uint32_t fcap, uint32_t* rs_pool,
                                                    uint32_t pool_chunks, uint32_t chunk_entries) {
  const FillOut fo = fill_lds_body(succ, ustart, gaps, gap_ids, flank_nodes, log_all, lvl_all, plk_all, xl_all, outs,
                                   num_oriented, nullptr, fcap, out_counter + 2, rs_pool, pool_chunks, chunk_entries, lp,
                                   sub_scratch, xo_all);
  __threadfence();  // the log, level offsets and links of this gap were written through: read them back from L2
  extract_lds_body(fo, gaps, gap_ids, flank_nodes, log_all, lvl_all, plk_all, xl_all, xo_all, sub_scratch, sub_out,
                   out_cap, out_counter, outs, outs_host, done_list, skip_confident, fcap);
}
// Same kernel with every right set in HBM from the start: for gaps known to outgrow the
// LDS (deep DP, -dist-error in the thousands); everything else of the gap stays in LDS.
__global__ __launch_bounds__(64) void g2s_fill_lds_rsg(G2S_FUSED_PARAMS, uint32_t* rs_global, uint32_t fcap) {
  const FillOut fo = fill_lds_body(succ, ustart, gaps, gap_ids, flank_nodes, log_all, lvl_all, plk_all, xl_all, outs,
                                   num_oriented, rs_global, fcap, out_counter + 2, nullptr, 0u, 0u, lp, sub_scratch,
                                   xo_all);
  __threadfence();
  extract_lds_body(fo, gaps, gap_ids, flank_nodes, log_all, lvl_all, plk_all, xl_all, xo_all, sub_scratch, sub_out,
                   out_cap, out_counter, outs, outs_host, done_list, skip_confident, fcap);
}

// ---------------------------------------------------------------------------
// launchers (host)
// ---------------------------------------------------------------------------
namespace g2s {

size_t fill_lds_bytes(uint32_t rs_cap, uint32_t fcap) {
  return 4u * (2 * fcap * 3 + (2 * fcap) * 2 + (2 * fcap) + LDS_TG + 3 * (2 * fcap) + 4 + LDS_TF + G2S_BULK_MAX + 3 * LDS_CW + rs_cap);
}
size_t extract_lds_bytes(uint32_t fcap) {
  const uint32_t w = fcap > 256u ? fcap : 256u;
  return 4u * ((w + 1) + 3 * w + 2 * fcap + fcap + 6 * fcap + fcap + 3 * LDS_XC + 4);
}
uint32_t fill_lds_frontier_cap() { return LDS_F; }
uint32_t fill_lds_max_fuz() { return LDS_TG - 1; }
size_t fill_lds_log_chunk_bytes(uint32_t states) { return (size_t)states * 32u; }

hipError_t launch_fill_lds(hipStream_t st, uint32_t ngaps, uint32_t rs_cap_max, uint32_t num_oriented,
                           const uint32_t* succ, const uint64_t* ustart, const GapDev* gaps, const uint32_t* gap_ids,
                           const uint32_t* flank_nodes, uint64_t* log_all, uint32_t* lvl_all, uint32_t* plk_all,
                           uint64_t* xl_all, uint64_t* xo_all, SubRec* sub_scratch, SubRec* sub_out,
                           unsigned long long out_cap,
                           unsigned long long* out_counter, GapOut* outs, GapOut* outs_host, uint32_t* done_list,
                           int skip_confident, uint32_t* rs_global, uint32_t fcap, uint32_t* rs_pool,
                           uint32_t pool_chunks, uint32_t chunk_entries, void* log_pool, uint32_t log_chunks,
                           uint32_t log_chunk_states) {
  if (ngaps == 0) return hipSuccess;
  LogPool lp;
  lp.base = (uint8_t*)log_pool;
  lp.chunks = log_pool ? log_chunks : 0u;
  lp.states = log_chunk_states;
  const size_t bytes = std::max(fill_lds_bytes(rs_global ? 0 : rs_cap_max, fcap), extract_lds_bytes(fcap));
  if (rs_global) {  // right set in HBM: no LDS for it
    hipError_t e = hipFuncSetAttribute((const void*)g2s_fill_lds_rsg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(g2s_fill_lds_rsg, dim3(ngaps), dim3(64), bytes, st, succ, ustart, gaps, gap_ids, flank_nodes,
                       log_all, lvl_all, plk_all, xl_all, xo_all, sub_scratch, sub_out, out_cap, out_counter, outs,
                       outs_host, done_list, skip_confident, num_oriented, lp, rs_global, fcap);
    return hipGetLastError();
  }
  hipError_t e = hipFuncSetAttribute((const void*)g2s_fill_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(g2s_fill_lds, dim3(ngaps), dim3(64), bytes, st, succ, ustart, gaps, gap_ids, flank_nodes, log_all,
                     lvl_all, plk_all, xl_all, xo_all, sub_scratch, sub_out, out_cap, out_counter, outs, outs_host,
                     done_list, skip_confident, num_oriented, lp, fcap, rs_pool, pool_chunks, chunk_entries);
  return hipGetLastError();
}

}  // namespace g2s
