// gap2seq_amd/csrc/fill_kernels.hip — HIP kernels (gfx950 / CDNA4) for phases
// A-C of Gap2Seq-core's fill_gap (/root/reference/src/Gap2Seq.cpp:858-1167).
//
// Mapping: one 64-lane wavefront per gap, the depth loop runs inside the kernel
// (no per-level launches).  A frontier entry is expanded by 4 adjacent lanes,
// one per nucleotide slot, so the 16-byte successor record of a node is read by
// one coalesced 4-lane access; duplicate targets inside a level are merged by
// a per-gap open-addressing table keyed by (oriented node, depth), the new
// states of a level are compacted into the next frontier with a wave ballot +
// popcount prefix.  All tables of a gap live in a work area in HBM (they are
// L2 resident in practice: a C2 gap touches ~100 KB).
//
// Integer/bitset work only: no MFMA on this path (HBM/latency bound).
//
// Memory-model note (MI355X_MICROARCH.md, "Inter-workgroup visibility"): a work
// area is only ever touched by the one wave that owns the gap, but it is
// mutated by L2 atomics, so every re-read of mutated words uses an agent-scope
// relaxed atomic load (sc1, served by L2) rather than a plain load that could
// hit a stale line in the CU's vector L1.
#include <hip/hip_runtime.h>

#include "fill_device.h"
#include "fill_launch.h"

namespace {

__device__ __forceinline__ uint32_t ld32(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t ld64(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Everything this wave has stored or issued atomically is complete before any
// later load is issued.
__device__ __forceinline__ void wave_mem_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t mix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return (uint32_t)k;
}
__device__ __forceinline__ uint64_t lanes_below(int lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ uint32_t flip(uint32_t v) { return v == G2S_DEV_INVALID ? v : (v ^ 1u); }

// ---- right set: open addressing over oriented node ids ----------------------
// 1 = inserted, 0 = already present, 2 = table full
__device__ int rs_insert(uint32_t* tab, uint32_t mask, uint32_t v) {
  uint32_t h = mix32(v) & mask;
  for (uint32_t i = 0; i <= mask; i++) {
    uint32_t old = atomicCAS(&tab[h], G2S_DEV_INVALID, v);
    if (old == G2S_DEV_INVALID) return 1;
    if (old == v) return 0;
    h = (h + 1) & mask;
  }
  return 2;
}
template <bool COHERENT>
__device__ bool rs_contains(const uint32_t* tab, uint32_t mask, uint32_t v) {
  uint32_t h = mix32(v) & mask;
  for (uint32_t i = 0; i <= mask; i++) {
    uint32_t cur = COHERENT ? ld32(&tab[h]) : tab[h];
    if (cur == v) return true;
    if (cur == G2S_DEV_INVALID) return false;
    h = (h + 1) & mask;
  }
  return false;
}

// ---- DP states: open addressing over (node << 32 | depth) ------------------
__device__ __forceinline__ uint64_t state_key(uint32_t v, int d) { return ((uint64_t)v << 32) | (uint32_t)d; }

__device__ uint32_t st_insert(uint64_t* keys, uint32_t mask, uint64_t key, uint32_t* isnew) {
  uint32_t h = mix64(key) & mask;
  for (uint32_t i = 0; i <= mask; i++) {
    unsigned long long old =
        atomicCAS((unsigned long long*)&keys[h], (unsigned long long)G2S_DEV_EMPTY64, (unsigned long long)key);
    if (old == G2S_DEV_EMPTY64) { *isnew = 1; return h; }
    if (old == key) { *isnew = 0; return h; }
    h = (h + 1) & mask;
  }
  *isnew = 0;
  return G2S_DEV_INVALID;
}
__device__ uint32_t st_find(const uint64_t* keys, uint32_t mask, uint64_t key) {
  uint32_t h = mix64(key) & mask;
  for (uint32_t i = 0; i <= mask; i++) {
    uint64_t cur = ld64(&keys[h]);
    if (cur == key) return h;
    if (cur == G2S_DEV_EMPTY64) return G2S_DEV_INVALID;
    h = (h + 1) & mask;
  }
  return G2S_DEV_INVALID;
}

}  // namespace

// ============================================================================
// Phase A — right BFS (Gap2Seq.cpp:871-982).
// The reference re-expands the whole border level by level and keeps per-depth
// rows, but the only consumer is a k-mer membership test (Gap2Seq.cpp:1050), so
// a visited-set BFS with the staggered right-flank seeds yields the same set.
// ============================================================================
__global__ __launch_bounds__(64) void g2s_right_bfs(const uint32_t* __restrict__ succ,
                                                     const uint32_t* __restrict__ predtab,
                                                     const GapDev* __restrict__ gaps,
                                                     const uint32_t* __restrict__ gap_ids,
                                                     const uint32_t* __restrict__ flank_nodes, uint32_t* rs_all,
                                                     uint32_t* rlog_all, GapOut* outs) {
  const uint32_t gi = __builtin_amdgcn_readfirstlane(gap_ids[blockIdx.x]);
  const GapDev gd = gaps[gi];
  const int lane = threadIdx.x;
  uint32_t* tab = rs_all + gd.rs_off;
  uint32_t* log = rlog_all + gd.rlog_off;
  const uint32_t mask = gd.rs_mask, cap = gd.rlog_cap;
  const uint32_t* rseeds = flank_nodes + gd.flank_off + (uint32_t)(gd.lmf + 1);

  uint32_t nlog = 0, flags = 0, xcount = 0;
  {
    const uint32_t s0 = rseeds[0];  // rightmost k-mer of the right flank (:878-899)
    if (s0 != G2S_DEV_INVALID) {
      if (lane == 0) { rs_insert(tab, mask, s0); log[0] = s0; }
      nlog = 1;
    }
  }
  wave_mem_fence();
  uint32_t bstart = 0, bend = nlog;
  bool overflow = false;
  for (int d = 1; d <= gd.right_half && !overflow; d++) {
    const uint32_t nb = bend - bstart;
    xcount += nb;
    for (uint32_t i0 = 0; i0 < nb * 4u; i0 += 64u) {
      const uint32_t i = i0 + (uint32_t)lane;
      uint32_t isnew = 0, p = G2S_DEV_INVALID;
      if (i < nb * 4u) {
        const uint32_t n = ld32(&log[bstart + (i >> 2)]);
        const uint32_t nt = i & 3u;
        // graph.predecessors(n) in GATB order: pred(v)[i] = succ(v^1)[i]^1
        p = predtab ? predtab[(size_t)n * 4 + nt] : flip(succ[(size_t)(n ^ 1u) * 4 + nt]);
        if (p != G2S_DEV_INVALID) {
          const int r = rs_insert(tab, mask, p);
          isnew = (r == 1);
          if (r == 2) flags |= G2S_DEV_OVERFLOW_A;
          // Q7: both strands of a k-mer in the set (a superset of "in one border").  Checked
          // by whoever comes second; two lanes inserting the pair in one instruction both look
          // after both inserts.
          if (isnew && rs_contains<true>(tab, mask, p ^ 1u)) flags |= G2S_DEV_Q7_A;
        }
      }
      const uint64_t m = __ballot(isnew);
      if (isnew) {
        const uint32_t off = nlog + (uint32_t)__popcll(m & lanes_below(lane));
        if (off < cap) log[off] = p;
      }
      nlog += (uint32_t)__popcll(m);
      if (nlog > cap) { overflow = true; break; }
    }
    wave_mem_fence();
    bstart = bend;
    bend = nlog;
    if (!overflow && d <= gd.rmf) {  // next right-flank seed (:953-976)
      const uint32_t s = rseeds[d];
      if (s != G2S_DEV_INVALID) {
        int r = 0;
        if (lane == 0) {
          r = rs_insert(tab, mask, s);
          if (r == 1 && rs_contains<true>(tab, mask, s ^ 1u)) flags |= G2S_DEV_Q7_A;
        }
        r = __shfl(r, 0);
        if (r == 1) {
          if (nlog < cap) { if (lane == 0) log[nlog] = s; } else overflow = true;
          nlog++;
          bend = nlog;
        }
        wave_mem_fence();
      }
    }
    if (bend == bstart && d >= gd.rmf) break;  // empty border and no seed left
  }
  // lanes carry partial flags
  for (int o = 32; o > 0; o >>= 1) flags |= __shfl_xor(flags, o);
  if (overflow) flags |= G2S_DEV_OVERFLOW_A;
  if (lane == 0) {
    GapOut* go = &outs[gi];
    go->flags = flags;
    go->n_right = nlog;
    go->x_right = xcount;
  }
}

// ============================================================================
// Phases B + C — left DP with pruning, flank seeds and target check
// (Gap2Seq.cpp:984-1167).  The frontier kernel.
// ============================================================================
__global__ __launch_bounds__(64) void g2s_left_dp(const uint32_t* __restrict__ succ,
                                                   const GapDev* __restrict__ gaps,
                                                   const uint32_t* __restrict__ gap_ids,
                                                   const uint32_t* __restrict__ flank_nodes,
                                                   const uint32_t* __restrict__ rs_all, uint64_t* st_keys_all,
                                                   uint32_t* st_cnt_all, uint32_t* slog_all, GapOut* outs) {
  const uint32_t gi = __builtin_amdgcn_readfirstlane(gap_ids[blockIdx.x]);
  const GapDev gd = gaps[gi];
  const int lane = threadIdx.x;
  GapOut* go = &outs[gi];
  uint32_t flags = go->flags;  // phase A flags (uniform)
  if (flags & G2S_DEV_OVERFLOW_A) return;

  const uint32_t* rs = rs_all + gd.rs_off;
  const uint32_t rmask = gd.rs_mask;
  uint64_t* keys = st_keys_all + gd.st_off;
  uint32_t* cnt = st_cnt_all + gd.st_off;
  uint32_t* log = slog_all + gd.slog_off;
  const uint32_t smask = gd.st_mask, cap = gd.slog_cap;
  const uint32_t* lseeds = flank_nodes + gd.flank_off;
  const uint32_t* targets = lseeds + (uint32_t)(gd.lmf + 1) + (uint32_t)(gd.rmf + 1);

  uint32_t nlog = 0, xcount = 0, lflags = 0;
  {
    const uint32_t s0 = lseeds[0];  // leftmost k-mer, count 1 at depth 0 (:995-1015)
    if (s0 != G2S_DEV_INVALID) {
      if (lane == 0) {
        uint32_t isnew;
        const uint32_t pos = st_insert(keys, smask, state_key(s0, 0), &isnew);
        atomicExch(&cnt[pos], 1u);
        log[0] = pos;
      }
      nlog = 1;
    }
  }
  wave_mem_fence();

  uint32_t bstart = 0, bend = nlog;
  bool overflow = false, found = false;
  int c_count = 0, n_len = 0, len0 = 0, len1 = 0, reached_j = 0;
  int d = 1;
  for (; d <= gd.D; d++) {
    const uint32_t nb = bend - bstart;
    xcount += nb;
    const bool unpruned = d < gd.prune_from;  // :1050 first disjunct
    for (uint32_t i0 = 0; i0 < nb * 4u; i0 += 64u) {
      const uint32_t i = i0 + (uint32_t)lane;
      uint32_t isnew = 0, pos2 = G2S_DEV_INVALID;
      if (i < nb * 4u) {
        const uint32_t pos = ld32(&log[bstart + (i >> 2)]);
        const uint32_t n = (uint32_t)(ld64(&keys[pos]) >> 32);
        const uint32_t nt = i & 3u;
        uint32_t np = ld32(&cnt[pos]);  // num_paths of the parent, saturated (:1047)
        if (np > G2S_DEV_MAX_PATHS) np = G2S_DEV_MAX_PATHS;
        const uint32_t v = succ[(size_t)n * 4 + nt];  // graph.successors(n), GATB order
        if (v != G2S_DEV_INVALID &&
            (unpruned || rs_contains<false>(rs, rmask, v) || rs_contains<false>(rs, rmask, v ^ 1u))) {
          pos2 = st_insert(keys, smask, state_key(v, d), &isnew);
          // <= 4 predecessors, each <= MAX_PATHS < 2^30: the u32 sum cannot wrap;
          // min(MAX, .) on read equals the reference's saturating adds (:1058-1060)
          if (pos2 != G2S_DEV_INVALID) atomicAdd(&cnt[pos2], np); else lflags |= G2S_DEV_OVERFLOW_B;
          // Q7: the other strand of this k-mer at the same level, the last level included
          // (seen by whichever of the two states is set second, or by both)
          if (isnew && st_find(keys, smask, state_key(v ^ 1u, d)) != G2S_DEV_INVALID) lflags |= G2S_DEV_Q7_B;
        }
      }
      const uint64_t m = __ballot(isnew);
      if (isnew) {
        const uint32_t off = nlog + (uint32_t)__popcll(m & lanes_below(lane));
        if (off < cap) log[off] = pos2;
      }
      nlog += (uint32_t)__popcll(m);
      if (nlog > cap) { overflow = true; break; }
    }
    if (overflow) break;
    wave_mem_fence();
    bstart = bend;
    bend = nlog;
    if (d <= gd.lmf) {  // next left-flank seed, row value ASSIGNED 1 (:1082-1105)
      const uint32_t s = lseeds[d];
      if (s != G2S_DEV_INVALID) {
        uint32_t isnew = 0, pos = G2S_DEV_INVALID;
        if (lane == 0) {
          pos = st_insert(keys, smask, state_key(s, d), &isnew);
          if (pos != G2S_DEV_INVALID) atomicExch(&cnt[pos], 1u);
          if (isnew && st_find(keys, smask, state_key(s ^ 1u, d)) != G2S_DEV_INVALID) lflags |= G2S_DEV_Q7_B;
        }
        isnew = __shfl(isnew, 0);
        pos = __shfl(pos, 0);
        if (pos == G2S_DEV_INVALID) { overflow = true; break; }
        if (isnew) {
          if (nlog >= cap) { overflow = true; break; }
          if (lane == 0) log[nlog] = pos;
          nlog++;
          bend = nlog;
        }
        wave_mem_fence();
      }
    }

    // ---- phase C: target check (:1107-1159) --------------------------------
    if (!found && d >= gd.g + gd.lmf + gd.rmf) {
      const int err = d - gd.g - (gd.lmf + gd.rmf);
      for (int jbase = 0; jbase <= gd.rmf && !found; jbase += 32) {
        const int j = jbase + (lane >> 1);
        const int which = lane & 1;
        const int L = which == 0 ? gd.g + gd.lmf + j + err : gd.g + gd.lmf + j - err;
        uint32_t c = 0;
        if (j <= gd.rmf && (which == 0 || (err != 0 && L >= 0))) {
          const uint32_t t = targets[j];
          if (t != G2S_DEV_INVALID) {
            const uint32_t pos = st_find(keys, smask, state_key(t, L));
            if (pos != G2S_DEV_INVALID) {
              c = ld32(&cnt[pos]);
              if (c > G2S_DEV_MAX_PATHS) c = G2S_DEV_MAX_PATHS;
            }
          }
        }
        const uint64_t hits = __ballot(c > 0);
        if (hits) {
          const int first = __builtin_ctzll(hits) >> 1;  // lowest j with a hit
          const uint32_t c1 = __shfl(c, first * 2), c2 = __shfl(c, first * 2 + 1);
          const uint32_t sum = c1 + c2;
          c_count = (int)(sum > G2S_DEV_MAX_PATHS ? G2S_DEV_MAX_PATHS : sum);
          reached_j = jbase + first;
          const int l1 = gd.g + gd.lmf + reached_j + err, l2 = gd.g + gd.lmf + reached_j - err;
          if (c1 > 0) { len0 = l1; n_len = 1; if (c2 > 0) { len1 = l2; n_len = 2; } }
          else { len0 = l2; n_len = 1; }
          found = true;
        }
      }
      if (found && !gd.all_paths) break;  // -best-only (:1156-1158)
    }
  }
  const int final_d = d;  // currentD when the reference's while loop ends (D+1 unless -best-only broke out)
  if (overflow) lflags |= G2S_DEV_OVERFLOW_B;
  for (int o = 32; o > 0; o >>= 1) lflags |= __shfl_xor(lflags, o);
  flags |= lflags;
  if (lane == 0) {
    go->flags = flags;
    go->n_states = nlog;
    go->x_left = xcount;
    go->final_d = final_d;
    go->c_count = c_count;
    go->n_len = n_len;
    go->len[0] = len0;
    go->len[1] = len1;
    go->reached_j = reached_j;
  }
}

// ============================================================================
// Phase D1 — backward closure over (node, depth) states
// (Gap2Seq.cpp:1169-1312), plus the closure the traceback (:1437-1517) can
// reach from (reachedTarget, pathLengths[i]).  One wave per gap, depth loop in
// the kernel, 4 lanes per border state (one per predecessor slot).  Output: a
// packed SubState array per gap; the host runs SCC / branch rule / traceback on it.
// ============================================================================
namespace {
// lanes with pos != INVALID claim their state; first claimer appends it to sub[].
__device__ __forceinline__ uint32_t sub_discover(uint32_t pos, uint32_t node, int depth, const uint32_t* cnt,
                                                 uint32_t* mark, SubState* sub, uint32_t nsub, uint32_t cap, int lane) {
  uint32_t won = 0;
  if (pos != G2S_DEV_INVALID) won = (atomicCAS(&mark[pos], 0u, 0xFFFFFFFFu) == 0u);
  const uint64_t m = __ballot(won);
  if (won) {
    const uint32_t idx = nsub + (uint32_t)__popcll(m & lanes_below(lane));
    if (idx < cap) {
      uint32_t c = ld32(&cnt[pos]);
      if (c > G2S_DEV_MAX_PATHS) c = G2S_DEV_MAX_PATHS;
      SubState st;
      st.node = node; st.depth = (uint32_t)depth; st.cnt = c; st.flags = 0;
      st.pred[0] = st.pred[1] = st.pred[2] = st.pred[3] = -1;
      sub[idx] = st;
    }
    __hip_atomic_store(&mark[pos], idx + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return nsub + (uint32_t)__popcll(m);
}
}  // namespace

__global__ __launch_bounds__(64) void g2s_extract(const uint32_t* __restrict__ succ,
                                                   const uint32_t* __restrict__ predtab,
                                                   const GapDev* __restrict__ gaps,
                                                   const uint32_t* __restrict__ gap_ids,
                                                   const uint32_t* __restrict__ flank_nodes,
                                                   const uint64_t* st_keys_all, const uint32_t* st_cnt_all,
                                                   uint32_t* st_mark_all, SubState* sub_scratch, SubState* sub_out,
                                                   unsigned long long* out_counter, GapOut* outs,
                                                   int skip_confident) {
  const uint32_t gi = __builtin_amdgcn_readfirstlane(gap_ids[blockIdx.x]);
  const GapDev gd = gaps[gi];
  const int lane = threadIdx.x;
  GapOut* go = &outs[gi];
  const uint32_t gflags = go->flags;
  const int c_count = go->c_count, n_len = go->n_len;
  if ((gflags & (G2S_DEV_OVERFLOW_A | G2S_DEV_OVERFLOW_B)) || !(c_count > 0 && n_len > 0)) return;  // :1169

  const uint64_t* keys = st_keys_all + gd.st_off;
  const uint32_t* cnt = st_cnt_all + gd.st_off;
  uint32_t* mark = st_mark_all + gd.st_off;
  SubState* sub = sub_scratch + gd.slog_off;
  const uint32_t smask = gd.st_mask, cap = gd.slog_cap;
  const uint32_t* lseeds = flank_nodes + gd.flank_off;
  const uint32_t* targets = lseeds + (uint32_t)(gd.lmf + 1) + (uint32_t)(gd.rmf + 1);
  const int len0 = go->len[0], len1 = go->len[1];
  const uint32_t reached = targets[go->reached_j];
  // all-paths mode: the only sink k-mer is kmer_right[rmf-1..] (Q3), none when rmf == 0 (Q4)
  const bool want_s = !skip_confident;
  const uint32_t sinknode = (want_s && gd.all_paths && gd.rmf >= 1) ? targets[gd.rmf - 1] : G2S_DEV_INVALID;
  const int lo_sink = max(0, gd.lmf + gd.g - gd.e);  // :1196
  const uint32_t t_flags = G2S_SUB_IN_T | G2S_SUB_START_T | ((want_s && !gd.all_paths) ? (G2S_SUB_IN_S | G2S_SUB_SINK) : 0u);

  uint32_t nsub = 0, bstart = 0, xcount = 0, lflags = 0;
  int win_lo = gd.D + 1;      // depths [win_lo, win_lo+63] have been probed for the sink state
  uint64_t sink_mask = 0;
  const int min_len = n_len > 1 ? min(len0, len1) : len0;
  for (int d2 = gd.D; d2 >= 0; d2--) {
    // ---- new paths starting at this depth (:1195-1259) -----------------------
    if (sinknode != G2S_DEV_INVALID && d2 >= lo_sink && d2 < win_lo) {
      win_lo = max(0, d2 - 63);
      const int dd = win_lo + lane;
      const bool hit = dd <= d2 && dd >= lo_sink && st_find(keys, smask, state_key(sinknode, dd)) != G2S_DEV_INVALID;
      sink_mask = __ballot(hit);
    }
    const bool s_here = sinknode != G2S_DEV_INVALID && d2 >= lo_sink && ((sink_mask >> (d2 - win_lo)) & 1ull);
    const bool t_here = (d2 == len0) || (n_len > 1 && d2 == len1);
    if (s_here || t_here) {
      uint32_t pos = G2S_DEV_INVALID, node = G2S_DEV_INVALID, f = 0;
      if (lane == 0 && s_here) { node = sinknode; f = G2S_SUB_IN_S | G2S_SUB_SINK; }
      if (lane == 1 && t_here) { node = reached; f = t_flags; }
      if (node != G2S_DEV_INVALID) pos = st_find(keys, smask, state_key(node, d2));
      nsub = sub_discover(pos, node, d2, cnt, mark, sub, nsub, cap, lane);
      if (nsub > cap) { lflags |= G2S_DEV_OVERFLOW_B; break; }
      wave_mem_fence();
      if (pos != G2S_DEV_INVALID) atomicOr(&sub[ld32(&mark[pos]) - 1u].flags, f);
      wave_mem_fence();
    }
    const uint32_t bend = nsub, nb = bend - bstart;
    if (nb == 0) {
      if (d2 < min_len && (sinknode == G2S_DEV_INVALID || d2 < lo_sink)) break;  // nothing can start below
      continue;
    }
    xcount += nb;
    const uint32_t lidx = (d2 <= gd.lmf) ? (lseeds[d2] >> 1) : 0xFFFFFFFFu;  // buildNode(kmer_left.substr(d2,k))
    // pass 1: claim predecessor states; pass 2: link + propagate closure flags
    for (int pass = 0; pass < 2; pass++) {
      for (uint32_t i0 = 0; i0 < nb * 4u; i0 += 64u) {
        const uint32_t i = i0 + (uint32_t)lane;
        uint32_t pos = G2S_DEV_INVALID, p = G2S_DEV_INVALID, f = 0, si = 0, nt = 0;
        if (i < nb * 4u) {
          si = bstart + (i >> 2);
          nt = i & 3u;
          const uint32_t cur = ld32(&sub[si].node);
          f = ld32(&sub[si].flags) & (G2S_SUB_IN_S | G2S_SUB_IN_T);
          if (d2 <= gd.lmf && (cur >> 1) == lidx) {  // :1270 end condition, k-mer comparison only
            if (pass == 0 && nt == 0) atomicOr(&sub[si].flags, G2S_SUB_SOURCE);
          } else if (d2 > 0) {
            p = predtab ? predtab[(size_t)cur * 4 + nt] : flip(succ[(size_t)(cur ^ 1u) * 4 + nt]);
            if (p != G2S_DEV_INVALID) pos = st_find(keys, smask, state_key(p, d2 - 1));
          }
        }
        if (pass == 0) {
          nsub = sub_discover(pos, p, d2 - 1, cnt, mark, sub, nsub, cap, lane);
        } else if (pos != G2S_DEV_INVALID) {
          const uint32_t idx = ld32(&mark[pos]) - 1u;
          if (idx < cap) {
            sub[si].pred[nt] = (int32_t)idx;
            atomicOr(&sub[idx].flags, f);
          }
          const uint32_t pos2 = st_find(keys, smask, state_key(p ^ 1u, d2 - 1));  // Q7: other strand in this border?
          if (pos2 != G2S_DEV_INVALID && ld32(&mark[pos2]) != 0u) lflags |= G2S_DEV_Q7_D;
        }
      }
      if (nsub > cap) break;
      wave_mem_fence();
    }
    if (nsub > cap) { lflags |= G2S_DEV_OVERFLOW_B; break; }
    bstart = bend;
  }
  for (int o = 32; o > 0; o >>= 1) lflags |= __shfl_xor(lflags, o);
  if (lflags & G2S_DEV_OVERFLOW_B) {
    if (lane == 0) go->flags = gflags | lflags;
    return;
  }
  // ---- pack: reserve exactly n_sub records in the dense output --------------------
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(out_counter, (unsigned long long)nsub);
  base = __shfl(base, 0);
  const uint4* src = (const uint4*)sub;
  uint4* dst = (uint4*)(sub_out + base);
  for (uint32_t i = (uint32_t)lane; i < nsub * 2u; i += 64u) {
    uint4 v;
    v.x = ld32((const uint32_t*)&src[i] + 0); v.y = ld32((const uint32_t*)&src[i] + 1);
    v.z = ld32((const uint32_t*)&src[i] + 2); v.w = ld32((const uint32_t*)&src[i] + 3);
    dst[i] = v;
  }
  if (lane == 0) {
    go->flags = gflags | lflags;
    go->n_sub = nsub;
    go->sub_off = base;
    go->x_sub = xcount;
  }
}

// ---------------------------------------------------------------------------
// launchers (host)
// ---------------------------------------------------------------------------
namespace g2s {

hipError_t launch_right_bfs(hipStream_t st, uint32_t ngaps, const uint32_t* succ, const uint32_t* predtab,
                            const GapDev* gaps, const uint32_t* gap_ids, const uint32_t* flank_nodes, uint32_t* rs_all,
                            uint32_t* rlog_all, GapOut* outs) {
  if (ngaps == 0) return hipSuccess;
  hipLaunchKernelGGL(g2s_right_bfs, dim3(ngaps), dim3(64), 0, st, succ, predtab, gaps, gap_ids, flank_nodes, rs_all,
                     rlog_all, outs);
  return hipGetLastError();
}

hipError_t launch_left_dp(hipStream_t st, uint32_t ngaps, const uint32_t* succ, const GapDev* gaps,
                          const uint32_t* gap_ids, const uint32_t* flank_nodes, const uint32_t* rs_all,
                          uint64_t* st_keys_all, uint32_t* st_cnt_all, uint32_t* slog_all, GapOut* outs) {
  if (ngaps == 0) return hipSuccess;
  hipLaunchKernelGGL(g2s_left_dp, dim3(ngaps), dim3(64), 0, st, succ, gaps, gap_ids, flank_nodes, rs_all, st_keys_all,
                     st_cnt_all, slog_all, outs);
  return hipGetLastError();
}

hipError_t launch_extract(hipStream_t st, uint32_t ngaps, const uint32_t* succ, const uint32_t* predtab,
                          const GapDev* gaps, const uint32_t* gap_ids, const uint32_t* flank_nodes,
                          const uint64_t* st_keys_all, const uint32_t* st_cnt_all, uint32_t* st_mark_all,
                          SubState* sub_scratch, SubState* sub_out, unsigned long long* out_counter, GapOut* outs,
                          int skip_confident) {
  if (ngaps == 0) return hipSuccess;
  hipLaunchKernelGGL(g2s_extract, dim3(ngaps), dim3(64), 0, st, succ, predtab, gaps, gap_ids, flank_nodes, st_keys_all,
                     st_cnt_all, st_mark_all, sub_scratch, sub_out, out_counter, outs, skip_confident);
  return hipGetLastError();
}

}  // namespace g2s
