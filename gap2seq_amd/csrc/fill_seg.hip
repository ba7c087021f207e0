// gap2seq_amd/csrc/fill_seg.hip — SEGMENT TIER of the fill path for gfx950 (CDNA4): phases A, B, C, D1 and
// (for closures without a repeated k-mer) D2 of /root/reference/src/Gap2Seq.cpp:858-1435 in one kernel, as a
// search over UNITIG SEGMENTS instead of DP levels.  tests/seg_model.py is the executable restatement of this
// algorithm that the CPU suite checks without a GPU; read the two side by side.
//
// Three kernels from one template (seg_fill_one):
//   g2s_fill_seg    one wave per gap, everything in LDS and registers: lists that fill the chip
//   g2s_fill_seg2   the same with two waves per gap (phase A beside the first half of phase B): short lists,
//                   whose launch ends with its slowest gap
//   g2s_fill_segx   the large variant for gaps that outgrow the LDS-resident capacities (-dist-error 2000):
//                   segments in global scratch, pending events in an LDS hash table, the right set as sorted
//                   index intervals searched per lane; persistent workgroups, one per compute unit
//
// Why segments: one wave per SIMD is bound by instruction issue and by dependent LDS / memory round trips
// (rocprofv3, profiles/r02_pmc_sq_*.json: a third of the wave cycles issue instructions, two thirds wait),
// so what a gap costs is the number of dependent steps it takes.  The LDS tier (fill_lds.hip) takes one
// step per DP level that is not unitig-internal for EVERY border state (~190 + ~190 steps for the slowest
// gaps of BASELINE config 2).  Here the unit of work is the segment: node ids are numbered along unitigs
// (dbg.hpp: inside a unitig the only successor of an even id v is v+2, of an odd id v-2, and v is that node's
// only predecessor), so a DP state (v, d, count) that enters a unitig determines the whole diagonal
// (v +- 2t, d + t, count) up to the unitig's end (one 32-byte record of urec[], seg_tables.hip), the last
// level D, or the first state the pruning rule (:1050) rejects.  A gap of config 2 has ~1 000 states but
// only ~17 segments, found in ~10 steps.
//
//   phase A  (:871-982) label-correcting search over unitigs; the right set is never materialised node by
//            node: it IS the table of (entry node, depth label) pairs, i.e. a union of k-mer index intervals
//            kept in registers (lane = entry).  Membership of a k-mer (:1050 ignores strand and depth) is one
//            compare + ballot; "how far does this run stay inside the right set" is interval arithmetic.
//   phase B  (:984-1105) ENTRY EVENTS (node, depth, count, <= 4 parents, stop depths) live in registers
//            (lane = pending event; merging = compare + ballot, no hash table).  An event is final once its
//            depth is below the HORIZON = min over pending events of (depth + states to the end of its
//            unitig): no pending event can still create a child at or above it.  All final events are
//            expanded in one step: their lengths under the pruning rule, one record per segment that
//            reaches its unitig's end, the children merged into the pending set.  Left-flank seeds
//            (:1082-1105, value ASSIGNED 1, Q6) are pre-inserted events with a fixed count; events above the
//            flank (depth < lmf) are cut after one state so that no segment runs across a seed state.
//   phase C  (:1107-1159) in closed form from the target hits of the segments: a hit (j, depth) is found at
//            level |depth - (g+lmf+j)| + g+lmf+rmf; smallest level, then smallest j.
//   Q7       both strands of a k-mer at one depth: among pending events (compare + ballot) and where an
//            upward and a downward segment of one unitig cross (arithmetic).
//   phase D1 (:1169-1312) backward closure over the segments, chunks of 64 in reverse (children
//            were created after their parent was expanded).
//   phase D2 (:1314-1435) when no k-mer repeats in the closure: the branch rule as a prefix sum over segments.
//   output   the closure as 32-byte segment records (SegRec), children before parents, parents in GATB's
//            predecessor order, written straight into pinned host memory; the host walks them (post.cpp).
// A gap that outgrows a capacity (segments, pending events, right-set entries, host buffer) is flagged and
// runs again in the next kernel of the chain (g2s_fill_segx, then the LDS tier).  Integer work only: no MFMA.
#include "sync_debug.h"
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <mutex>
#include <set>
#include <utility>

#include "../../include/g2s.h"
#include "fill_device.h"
#include "fill_seg.h"
#include "flank_device.h"

#include <hip/hip_ext.h>

#include "seg_device.h"


// Q7 between segments, the sorted way (see seg_fill_one): true/false in *found; returns false when there are more upward
// segments than sbuf holds (nothing decided)
__device__ __forceinline__ bool seg_q7_sorted(const uint32_t* s_node, const uint32_t* s_dl, uint32_t nseg, uint64_t* sbuf, uint32_t cap, int lane,
                                              bool* found) {
    uint32_t nu = 0;
    bool anydn = false;
    for (uint32_t b0 = 0; b0 < nseg; b0 += 64u) {
      const uint32_t b = b0 + (uint32_t)lane;
      const bool hb = b < nseg;
      const uint32_t nb_ = hb ? s_node[b] : 0u, lb = hb ? s_dl[b] >> 16 : 0u;
      const bool up = hb && !(nb_ & 1u) && lb > 0u;
      const uint64_t m = __ballot(up);
      if (nu + (uint32_t)__popcll(m) > cap) return false;
      if (up) sbuf[nu + (uint32_t)__popcll(m & below(lane))] = ((uint64_t)(nb_ >> 1) << 32) | (uint64_t)((nb_ >> 1) + lb - 1u);
      nu += (uint32_t)__popcll(m);
      if (__ballot(hb && (nb_ & 1u) && lb > 0u)) anydn = true;
    }
    lds_sync();
    if (nu > 0u && anydn) {
      uint32_t n2 = 2;
      while (n2 < nu) n2 <<= 1;
      for (uint32_t i = nu + (uint32_t)lane; i < n2; i += 64u) sbuf[i] = SEGX_EMPTY64;
      lds_sync();
      lds_sort64(sbuf, n2, lane);
      bool unused = false;
      const uint32_t Mu = lds_merge_intervals(sbuf, nu, lane, &unused);
      const uint32_t Pu = 1u << (31 - __builtin_clz(Mu));
      const uint32_t* uw = (const uint32_t*)sbuf;
      for (uint32_t b0 = 0; b0 < nseg && !*found; b0 += 64u) {
        const uint32_t b = b0 + (uint32_t)lane;
        const bool hb = b < nseg;
        const uint32_t nb_ = hb ? s_node[b] : 0u, dlb = hb ? s_dl[b] : 0u;
        const int ib = (int)(nb_ >> 1), db = (int)(dlb & 0xFFFFu), lb = (int)(dlb >> 16);
        const bool down = hb && (nb_ & 1u) && lb > 0;
        uint32_t pos = 0;
        for (uint32_t st = Pu; st; st >>= 1) {
          const uint32_t pp = pos + st;
          if (pp <= Mu && uw[2u * (pp - 1u) + 1u] <= (uint32_t)ib) pos = pp;
        }
        const bool cand = down && pos > 0u && (int)uw[2u * (pos - 1u)] >= ib - lb + 1;
        for (uint64_t cm = __ballot(cand); cm && !*found; cm &= cm - 1) {
          const int l = __builtin_ctzll(cm);
          const int ibl = (int)rl((uint32_t)ib, l), dbl = (int)rl((uint32_t)db, l), lbl = (int)rl((uint32_t)lb, l);
          for (uint32_t a0 = 0; a0 < nseg; a0 += 64u) {
            const uint32_t a = a0 + (uint32_t)lane;
            const bool ha = a < nseg;
            const uint32_t na = ha ? s_node[a] : 1u, dla = ha ? s_dl[a] : 0u;
            const int ia = (int)(na >> 1), da = (int)(dla & 0xFFFFu), la = (int)(dla >> 16);
            const int sdiff = ibl - ia, ddiff = dbl - da;
            const int t1 = (sdiff + ddiff) >> 1, t2 = (sdiff - ddiff) >> 1;
            if (ballot_and(ha, !(na & 1u), !((sdiff + ddiff) & 1), t1 >= 0, t1 < la, t2 >= 0, t2 < lbl)) { *found = true; break; }
          }
        }
      }
    }
    lds_sync();
    return true;
  }
// every downward segment against every upward one, both sides a chunk at a time in registers
__device__ __forceinline__ bool seg_q7_all_pairs(const uint32_t* s_node, const uint32_t* s_dl, uint32_t nseg, int lane) {
  uint64_t anyup = 0, anydn = 0;
  for (uint32_t b0 = 0; b0 < nseg; b0 += 64u) {
    const uint32_t b = b0 + (uint32_t)lane;
    anyup |= __ballot(b < nseg && !(s_node[b < nseg ? b : 0] & 1u));
    anydn |= __ballot(b < nseg && (s_node[b < nseg ? b : 0] & 1u));
  }
  if (!(anyup && anydn)) return false;
  for (uint32_t b0 = 0; b0 < nseg; b0 += 64u) {
    const uint32_t b = b0 + (uint32_t)lane;
    const bool hb = b < nseg;
    const uint32_t nb_ = hb ? s_node[b] : 0u, dlb = hb ? s_dl[b] : 0u;
    const int ib = (int)(nb_ >> 1), db = (int)(dlb & 0xFFFFu), lb = (int)(dlb >> 16);
    const uint64_t dm0 = __ballot(hb && (nb_ & 1u));
    if (!dm0) continue;
    for (uint32_t a0 = 0; a0 < nseg; a0 += 64u) {
      const uint32_t a = a0 + (uint32_t)lane;
      const bool ha = a < nseg;
      const uint32_t na = ha ? s_node[a] : 1u, dla = ha ? s_dl[a] : 0u;
      const bool upa = ha && !(na & 1u);
      if (!__ballot(upa)) continue;
      const int ia = (int)(na >> 1), da = (int)(dla & 0xFFFFu), la = (int)(dla >> 16);
      const uint64_t upm = __ballot(upa && la > 0);
      for (uint64_t dm = dm0; dm; dm &= dm - 1) {
        const int l = __builtin_ctzll(dm);
        const int ibl = (int)rl((uint32_t)ib, l), lbl = (int)rl((uint32_t)lb, l);
        // (two segments can only meet on a k-mer both hold: the downward one's indices ibl - lbl + 1 .. ibl against the
        // upward ones' ia .. ia + la - 1 — nearly every pair is settled by these two comparisons)
        if (!(upm & __ballot(ia <= ibl) & __ballot(ibl - lbl < ia + la))) continue;
        const int sdiff = ibl - ia, ddiff = (int)rl((uint32_t)db, l) - da;
        const int t1 = (sdiff + ddiff) >> 1, t2 = (sdiff - ddiff) >> 1;
        if (upm & ballot_and(!((sdiff + ddiff) & 1), t1 >= 0, t1 < la, t2 >= 0, t2 < lbl)) return true;
      }
    }
  }
  return false;
}
// Do two segments share a k-mer (as index intervals, whatever their depths)?  When none do, no closure holds a k-mer at
// two depths either: phase D2 can skip its own test over the closure's extents.  Every chunk against itself and
// the chunks before it, both sides in registers.
__device__ __forceinline__ bool seg_extents_overlap(const uint32_t* s_node, const uint32_t* s_dl, uint32_t nseg, int lane) {
  for (uint32_t b0 = 0; b0 < nseg; b0 += 64u) {
    const uint32_t b = b0 + (uint32_t)lane;
    const bool hb = b < nseg;
    const uint32_t v0 = hb ? s_node[b] : 0u, lb = hb ? s_dl[b] >> 16 : 0u;
    const bool in = hb && lb > 0u;
    const uint32_t idx = v0 >> 1;
    const uint32_t ilo = (v0 & 1u) ? idx - (lb - 1u) : idx, ihi = (v0 & 1u) ? idx : idx + (lb - 1u);
    for (uint32_t a0 = 0; a0 <= b0; a0 += 64u) {
      uint32_t alo_ = ilo, ahi_ = ihi;
      bool a_in = in;
      if (a0 != b0) {
        const uint32_t a = a0 + (uint32_t)lane;  // (a < nseg: an earlier chunk is full)
        const uint32_t va = s_node[a], la = s_dl[a] >> 16, ia = va >> 1;
        a_in = la > 0u;
        alo_ = (va & 1u) ? ia - (la - 1u) : ia;
        ahi_ = (va & 1u) ? ia : ia + (la - 1u);
      }
      for (uint64_t am = __ballot(a_in); am; am &= am - 1) {
        const int al = __builtin_ctzll(am);
        const uint32_t lo_a = rl(alo_, al), hi_a = rl(ahi_, al);
        if (ballot_and(in, b > a0 + (uint32_t)al, ilo <= hi_a, lo_a <= ihi)) return true;
      }
    }
  }
  return false;
}
// (regular tier) Q7 between the segments and "no two segments share a k-mer" in one go.  A few dozen segments: both
// directly.  Beyond: ALL segments' index intervals sorted and merged once — no overlap at all settles both (no pair
// can meet without overlapping), an overlap only within one orientation still settles Q7; only when an upward and a
// downward segment overlap does the detailed check run.
__device__ __forceinline__ bool seg_q7_between(const uint32_t* s_node, const uint32_t* s_dl, uint32_t nseg, uint64_t* sbuf, uint32_t cap, int lane);
__device__ __forceinline__ bool seg_extents_overlap(const uint32_t* s_node, const uint32_t* s_dl, uint32_t nseg, int lane);
// (want_apart: the second verdict is of use — phase D2 runs on the device for this gap, and somebody has the time)
__device__ __forceinline__ void seg_cross_check(const uint32_t* s_node, const uint32_t* s_dl, uint32_t nseg, uint64_t* sbuf, uint32_t cap, int lane,
                                                bool want_apart, bool* q7, bool* apart) {
  *q7 = false;
  *apart = false;
  if (nseg <= 64u || !want_apart) {
    *q7 = seg_q7_between(s_node, s_dl, nseg, sbuf, cap, lane);
    if (want_apart) *apart = !seg_extents_overlap(s_node, s_dl, nseg, lane);
    return;
  }
  uint32_t nk = 0;
  bool fits = true;
  for (uint32_t b0 = 0; b0 < nseg; b0 += 64u) {
    const uint32_t b = b0 + (uint32_t)lane;
    const bool hb = b < nseg;
    const uint32_t v0 = hb ? s_node[b] : 0u, lb = hb ? s_dl[b] >> 16 : 0u;
    const bool in = hb && lb > 0u;
    const uint64_t m = __ballot(in);
    if (nk + (uint32_t)__popcll(m) > cap) { fits = false; break; }
    const uint32_t idx = v0 >> 1;
    const uint32_t lo = (v0 & 1u) ? idx - (lb - 1u) : idx, hi = (v0 & 1u) ? idx : idx + (lb - 1u);
    if (in) sbuf[nk + (uint32_t)__popcll(m & below(lane))] = ((uint64_t)lo << 32) | ((uint64_t)(v0 & 1u) << 31) | (uint64_t)hi;
    nk += (uint32_t)__popcll(m);
  }
  lds_sync();
  if (!fits) { *q7 = seg_q7_between(s_node, s_dl, nseg, sbuf, cap, lane); return; }
  if (nk < 2u) { *apart = true; return; }
  uint32_t n2 = 2;
  while (n2 < nk) n2 <<= 1;
  for (uint32_t i = nk + (uint32_t)lane; i < n2; i += 64u) sbuf[i] = SEGX_EMPTY64;
  lds_sync();
  lds_sort64(sbuf, n2, lane);
  bool cross = false, overlap = false;
  (void)lds_merge_intervals(sbuf, nk, lane, &cross, &overlap);
  *apart = !overlap;
  if (cross) *q7 = seg_q7_between(s_node, s_dl, nseg, sbuf, cap, lane);
}
// (regular tier) the whole check: direct for a few dozen segments, sorted beyond
__device__ __forceinline__ bool seg_q7_between(const uint32_t* s_node, const uint32_t* s_dl, uint32_t nseg, uint64_t* sbuf, uint32_t cap, int lane) {
  bool found = false;
  if (nseg <= 64u || !seg_q7_sorted(s_node, s_dl, nseg, sbuf, cap, lane, &found)) found = seg_q7_all_pairs(s_node, s_dl, nseg, lane);
  return found;
}

// One gap, one wave.  BIG = false: the tier proper (segments in LDS, pending events and the right set in
// registers).  BIG = true: the same search for the gaps that outgrow those capacities (-dist-error 2000:
// thousands of segments, hundreds of pending events, thousands of right-set entries): segments in the
// workgroup's global scratch `scr`, pending events in an LDS hash table, the right set as a sorted array
// of disjoint index intervals in LDS that every lane searches on its own.
// TWO (with BIG = false): the workgroup has two waves — wave 1 runs phase A while wave 0 runs the part of
// phase B that does not look at the right set yet (:1050 first consults it at depth g/2 + e/2 + lmf), so
// that phase A leaves the critical path of the slowest gaps; they meet at one barrier.
#ifndef G2S_GUESS_LATE_CYCLES_LONG
#define G2S_GUESS_LATE_CYCLES_LONG 200000u /* one wave per gap: the same for the gaps a chip-filling list's launch ends with */
#endif
#ifndef G2S_GUESS_LATE_CYCLES
#define G2S_GUESS_LATE_CYCLES 100000u /* two waves per gap: a search that ends later than this leaves its traceback to the trace kernel */
#endif

template <bool BIG, bool TWO>
__device__ __forceinline__ void seg_fill_one(uint32_t* lds, const SegArgs& A, const uint32_t x /* position in the launch */,
                                             uint32_t* scr) {
  static_assert(!(BIG && TWO), "the large variant is one wave per gap");
  const uint32_t* __restrict__ succ = A.succ;
  const uint32_t* __restrict__ urec = A.urec;
  const uint32_t* __restrict__ gap_ids = A.gap_ids;
  const uint32_t* __restrict__ flank_nodes = A.flank_nodes;
  SubRec* sub_out = A.sub_out;
  const unsigned long long out_cap = A.out_cap;
  unsigned long long* out_counter = A.out_counter;
  GapOut* outs = A.outs;
  GapOut* outs_host = A.outs_host;
  uint32_t* done_list = A.done_list;
  const int skip_confident = A.skip_confident;
  uint32_t* dbg = A.dbg;
  const uint32_t dbg_words = A.dbg_words;
  constexpr uint32_t CAP = BIG ? G2S_SEGX_CAP : G2S_SEG_CAP;
  // segment arrays: LDS (7 arrays of G2S_SEG_CAP words + left seeds), or the scratch; s_aux / s_t always in LDS
  uint32_t* s_node = BIG ? scr : lds;           // entry node of the segment
  uint32_t* s_dl = s_node + CAP;                // entry depth | length << 16
  uint32_t* s_cnt = s_dl + CAP;                 // path count of every state of the segment
  uint32_t* s_p01 = s_cnt + CAP;                // parents (segment ids, 16 bits each, 0xFFFF = none)
  uint32_t* s_p23 = s_p01 + CAP;
  uint32_t* s_gen = s_p23 + CAP;                // (BIG) generation, copied into s_aux before phase D1
  uint32_t* s_aux = BIG ? lds : s_p23 + CAP;    // generation | closure marks of the children << 16; later: emit offset
  uint32_t* s_t = s_aux + CAP;                  // last closure state: towards a sink | from a traceback start << 16 (0xFFFF none)
  uint32_t* l_seed = BIG ? lds + (SEGX_LDS_WORDS - 32u) : s_t + CAP;  // left-flank seeds by depth [32]

  const int lane = (int)(threadIdx.x & 63u);
  const int wave = TWO ? (int)(threadIdx.x >> 6) : 0;
  const uint32_t gi = gap_ids ? uni(gap_ids[x]) : x;  // (no list: the launch takes the gaps in list order)
  // (look-ups in this kernel, the text at a fixed stride by gap: asked for beside the gap's descriptor — one round trip
  // instead of two; on a short list both come over the link)
  bool inl = false, inl_early = false;
  uint32_t tw0 = 0u, tw1 = 0u;  // (the text's words lane and lane + 64, in flight while the descriptor travels)
  if constexpr (!BIG) {
    inl = A.inl_text != nullptr;
    if (inl && A.inl_stride) {
      const uint32_t* src = (const uint32_t*)(A.inl_text + (size_t)gi * A.inl_stride);
      const uint32_t words = A.inl_stride / 4u, w = threadIdx.x & 63u;
      if (w < words) tw0 = src[w];
      if (w + 64u < words) tw1 = src[w + 64u];
      inl_early = true;
    }
  }
  const GapDev gd = A.gaps.load(gi);
  GapOut* go = &outs[gi];
  const uint32_t* lseeds = flank_nodes + gd.flank_off;
  const uint32_t* rseeds = lseeds + (uint32_t)(gd.lmf + 1);
  const uint32_t* targets = rseeds + (uint32_t)(gd.rmf + 1);
  const int D = gd.D, lmf = gd.lmf, rmf = gd.rmf;
  const unsigned long long cyc0 = __builtin_amdgcn_s_memtime();

  uint32_t flags = 0;
  bool overflow = D >= 32767 || rmf + 1 > 32 || lmf + 1 > 32;  // (16-bit depths and lengths, 32 seeds / targets)

  // Results go to pinned host memory as each gap finishes (as in the LDS tier).  One wave publishes what it stored
  // itself: the record it wrote to device memory went through the L1 into this XCD's L2, where the copy below reads
  // it back (agent-scope loads bypass the L1) once "s_waitcnt vmcnt(0)" says the stores were acknowledged — no
  // fence is needed for that.  A system-scope release (a write-back of this XCD's L2) then puts closure segments
  // and record in host memory, in front of the gap's entry in done_list.  (This used to be __threadfence() +
  // __threadfence_system() + a release store: three write-backs of the XCD's L2 and two invalidations per gap,
  // which every other wave of the XCD pays for.  A write-back has to stay.  Tried on the GPU: "vmcnt(0)" alone in
  // its place — the host saw flags before records, plain stores to this memory do stay in the L2; and
  // write-through (sc0 sc1) stores with "vmcnt(0)" — correct, but every 16-byte store is then a write over the
  // link of its own: config 3's launch took 10-16 ms instead of 0.8.)
  // Batches (pub_batch > 1): one write-back announces pub_batch gaps.  A finished wave — its stores acknowledged,
  // i.e. in the L2 of ITS XCD — takes a ticket from that XCD's counter and leaves its gap in that XCD's list;
  // the wave whose ticket completes a batch waits for the batch's entries (their waves hold tickets, so they are
  // running and about to store them), writes this XCD's L2 back — which holds everything those waves stored
  // before they took their tickets — and enters the whole batch in done_list.  The XCD is read from the hardware
  // register, not inferred from the block index.  Gaps of a batch that never completes are not announced: the
  // host takes them at the end of the launch, which flushes everything.
  // MEMORY-MODEL ASSUMPTION of the batches (not something the HIP memory model promises; checked on gfx950 by
  // the parity suite and the differential campaigns on the host path, which compare every announced gap field by field):
  // plain stores of wave A to fine-grained host memory that A has waited for ("s_waitcnt vmcnt(0)") sit in the L2 of
  // A's XCD, and a later "buffer_wbl2 sc0 sc1" by ANOTHER wave B of the same XCD writes them back with B's own.
  // The ticket (an agent-scope atomic A performs after its wait) is what orders A's stores before B's write-back.
  // Since round 3 this path only serves the host path of lists beyond 2 048 gaps (G2S_RESIDENT=0, or a list the
  // device gave back): resident mode announces nothing.  A batch whose closing wave gives up waiting (2^18 looks:
  // never seen) or that never fills is not lost — the host takes every gap not announced when the launch has ended.
  // The large variant's workgroups publish what several waves stored: it keeps the fences.
  // (look-ups in this kernel: the host's copy of the gap's node ids is written only for the gaps the host will look
  // at — a closure it analyses, a gap that outgrew the tier — : 132 bytes a gap through the link for every gap were
  // 1.3 MB of a 10 000-gap list, 30 us of its launch)
  auto nodes_to_host = [&]() {
    if constexpr (!BIG) {
      if (A.inl_text != nullptr && A.inl_nodes_host != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t nit = (uint32_t)(gd.lmf + 1) + 2u * (uint32_t)(gd.rmf + 1);
        for (uint32_t q = (uint32_t)(threadIdx.x & 63u); q < nit; q += 64u)
          A.inl_nodes_host[gd.flank_off + q] = __hip_atomic_load(&A.inl_nodes_dev[gd.flank_off + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  auto publish = [&]() {
    if (A.resident) {
      if (flags & (G2S_DEV_OVERFLOW_A | G2S_DEV_OVERFLOW_B)) nodes_to_host();
      if (A.ovf_list && lane == 0 && (flags & (G2S_DEV_OVERFLOW_A | G2S_DEV_OVERFLOW_B)) && !(flags & G2S_DEV_WATCHDOG))
        A.ovf_list[atomicAdd(out_counter + 1, 1ull)] = gi;
      // (this gap is through: what a polling g2s_d2_small counts to know that no entry can come any more — behind the
      // gap's own entry, whose counter this wave has waited for; 64 counters, the result not asked for)
      if (A.d2_list && A.d2_tag && lane == 0) (void)atomicAdd(out_counter + 32 + (gi & 63u), 1ull);
      return;
    }
    if constexpr (BIG) __threadfence();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((uint32_t)lane < sizeof(GapOut) / 4u)
      ((uint32_t*)&outs_host[gi])[lane] = __hip_atomic_load(&((const uint32_t*)go)[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (BIG) {
      __threadfence_system();
      if (lane == 0) {
        const unsigned long long at = atomicAdd(out_counter + 1, 1ull);
        __hip_atomic_store(&done_list[at], gi, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    } else {
      const uint32_t B = A.pub_batch;
      if (B <= 1u) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: buffer_wbl2 sc0 sc1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler leaves no wait between the write-back and a following store)
        if (lane == 0) {
          const unsigned long long at = atomicAdd(out_counter + 1, 1ull);
          __hip_atomic_store(&done_list[at], gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the copy of the record too
      const uint32_t xcd = (uint32_t)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;  // HW_REG_XCC_ID, bits 0-3
      uint32_t* list = A.xcd_list + (size_t)xcd * A.xcd_stride;
      uint32_t t = 0;
      if (lane == 0) {
        t = (uint32_t)atomicAdd(A.xcd_tickets + xcd, 1ull);
        __hip_atomic_store(&list[t], gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      t = uni((uint32_t)__shfl((int)t, 0));
      if (((t + 1u) & (B - 1u)) != 0u) return;
      const uint32_t first = t + 1u - B;
      uint32_t e = G2S_DEV_INVALID;
      bool all = true;
      if ((uint32_t)lane < B) {
        uint32_t spins = 0;
        do {
          e = __hip_atomic_load(&list[first + (uint32_t)lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while (e == G2S_DEV_INVALID && ++spins < (1u << 18));
        all = e != G2S_DEV_INVALID;
      }
      if (__ballot(!all)) return;  // (never expected; the batch is then taken at the end of the launch)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(out_counter + 1, (unsigned long long)B);
      base = (unsigned long long)(uint32_t)__shfl((int)(uint32_t)base, 0);
      if ((uint32_t)lane < B) __hip_atomic_store(&done_list[base + (uint32_t)lane], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  };

  // ---- the gap's flank nodes: lane d holds left seed d (f_l), lane j right seed j (f_r) and target j (tg).  From the
  // table the look-up kernel wrote (flank_lookup.hip) — or, when the list stays on the device and the launch says so
  // (SegArgs.inl_text), resolved HERE from the gap's flank text: the look-up kernel in front of this one was a launch,
  // a dependency and two round trips of the link per gap (10 us of a 500-gap step, 75 us of a 10 000-gap one, most
  // of it the copy of descriptors and text in front of it); a wave's own look-ups are ~2 us at the head of its gap.
  // Two waves per gap: wave 1 resolves (and stores) the right seeds, wave 0 the left seeds and the targets.  The
  // ids also go where the table would hold them (g2s_d2_*, the large variant's reruns and the host read them there).
  uint32_t f_l = G2S_DEV_INVALID, f_r = G2S_DEV_INVALID, tg = G2S_DEV_INVALID;
  if (inl) {
    if constexpr (!BIG) {
      // LDS of this wave (nothing of the search lives there yet): text [0, 128) words, node ids [128, 224)
      uint32_t* stage = (TWO && wave == 1) ? lds + (7u * G2S_SEG_CAP + 32u) : lds;
      const int k = A.lk.k;
      const int llen = k + lmf, rlen = k + rmf, tail = llen + rlen;
      const uint32_t words = (uint32_t)(tail + rlen + 3) / 4u;  // (k <= 63, lmf, rmf <= 31: at most 71)
      if (!inl_early) {
        const uint32_t* src = (const uint32_t*)(A.inl_text + gd.rs_mask);  // (4-byte aligned, padded: g2s_batch_prepare)
        if ((uint32_t)lane < words) tw0 = src[lane];
        if ((uint32_t)lane + 64u < words) tw1 = src[lane + 64];
      }
      stage[lane] = tw0;
      stage[lane + 64] = tw1;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      uint32_t* fn = stage + 128;
      const int nl = lmf + 1, nr = rmf + 1;
      for (int i = lane; i < nl + 2 * nr; i += 64) {
        const bool right_seed = i >= nl && i < nl + nr;
        if (TWO && right_seed != (wave == 1)) continue;
        const int at = g2s::flank_item_offset(i, k, lmf, rmf, tail);
        const uint32_t node = A.lk.wide ? g2s::flank_node_of<g2s::u128>(A.lk, stage, at) : g2s::flank_node_of<uint64_t>(A.lk, stage, at);
        fn[i] = node;
        A.inl_nodes_dev[gd.flank_off + (uint32_t)i] = node;  // (the pinned copy: only of the gaps the host will look at — publish)
      }
      lds_sync();
      __builtin_amdgcn_wave_barrier();
      if (lane <= lmf && lane < 32 && (!TWO || wave == 0)) f_l = fn[lane];
      if (lane <= rmf && lane < 32 && (!TWO || wave == 1)) f_r = fn[nl + lane];
      if (lane <= rmf && lane < 32 && (!TWO || wave == 0)) tg = fn[nl + nr + lane];
      lds_sync();
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    if (lane <= lmf && lane < 32) f_l = lseeds[lane];
    if (lane <= rmf && lane < 32) { f_r = rseeds[lane]; tg = targets[lane]; }  // lane j: target k-mer j
  }
  if (lane < 32 && wave == 0) l_seed[lane] = f_l;

  uint32_t nA = 0, roundsA = 0, nvis = 0, xa = 0;
  uint32_t nM = 0;  // (regular tier) the right set's entries as merged k-mer index intervals: how many
  // (!BIG) the right-set entries in registers: lane l of set s holds entry 64 s + l, as a k-mer index interval
  uint32_t an[G2S_SEG_ASETS], al[G2S_SEG_ASETS], ar[G2S_SEG_ASETS], alo[G2S_SEG_ASETS], ahi[G2S_SEG_ASETS];
#pragma unroll
  for (int s = 0; s < G2S_SEG_ASETS; s++) { an[s] = G2S_DEV_INVALID; al[s] = 0; ar[s] = 0; alo[s] = 1u; ahi[s] = 0u; }
  // (BIG) the right set as M disjoint, sorted index intervals in LDS: ivw[2 i] = last index, ivw[2 i + 1] = first
  uint32_t* ivw = lds;
  uint32_t M = 0, ivP = 0;  // ivP: largest power of two <= M
  // ---------------- phase A: the right set as (entry node, depth label) pairs ----------------
  // Label-correcting search over unitigs (:871-982 computes {v : fewest predecessor steps from a
  // right seed <= right_half}, seed j entering at depth j; only membership is consumed, :1050).
  // An entry (node, label) covers its unitig backwards for min(rem, right_half - label) steps;
  // where the unitig ends with budget left, the predecessors of its last node are proposed with
  // label + steps + 1.  All entries of a round are expanded at once (lane = entry): one load of
  // rem[], one successor record, and the proposals of the whole wave go through one LDS table
  // keyed by node with 64-bit atomic min on (node << 32 | label) — a per-proposal compare against
  // register-resident entries costs ~1.5 k cycles of scalar/vector ping-pong each (measured).
  // LDS (one wave: aliasing the segment arrays, which phase B fills later; two waves: behind them):
  //   lab[ALAB] u64 | labrem[ALAB] u32 | q[2][ACAP] u32 nodes | q[2][ACAP] u32 table positions | 16 words
  constexpr uint32_t ACAP = 64u * G2S_SEG_ASETS, ALAB = 2u * ACAP;
  uint64_t* lab = (uint64_t*)(TWO ? lds + (7u * G2S_SEG_CAP + 32u) : lds);
  uint32_t* labrem = (uint32_t*)(lab + ALAB);
  uint32_t* aq0 = labrem + ALAB;
  uint32_t* ash = aq0 + 4u * ACAP;  // (two waves) what wave 1 hands over: entries, rounds, flags, overflow, cycles
  // (two waves) the records of the entries queued for the next round, asked for when they are PROPOSED — before the
  // probe of the table and its atomic tell whether the proposal improves anything — and parked here by queue position:
  // the next round reads LDS where it waited a thousand cycles for memory (40 % of a round of the search, and the
  // gaps that end a short list's launch wait for this wave)
  uint4* qrec = (uint4*)(ash + 16u);               // [2][ACAP] the unitig's exit
  uint32_t* qrem = (uint32_t*)(qrec + 2u * ACAP);  // [2][ACAP] steps to its end
  // the table's entries, packed (the queues are idle by then), and from there into registers: lane l of set s
  // holds entry 64 s + l as a k-mer index interval; then Q7 in the right set
  auto load_right_set = [&]() {
    if constexpr (!BIG) {
      const uint32_t* cnode = aq0;
      const uint32_t* clab = aq0 + ACAP;
      const uint32_t* crem = (const uint32_t*)lab;
      if (!overflow) {
#pragma unroll
        for (int s = 0; s < G2S_SEG_ASETS; s++) {
          const uint32_t e = (uint32_t)s * 64u + (uint32_t)lane;
          if (e < nA) { an[s] = cnode[e]; al[s] = clab[e]; ar[s] = crem[e]; }
        }
        lds_sync();
      }
      // the set's size (intervals of one unitig may overlap: an upper bound) and the expansions of the search
#pragma unroll
      for (int s = 0; s < G2S_SEG_ASETS; s++) {
        const bool have = (uint32_t)s * 64u + (uint32_t)lane < nA;
        const uint32_t steps = have ? min(ar[s], (uint32_t)gd.right_half - al[s]) : 0u;
        if ((uint32_t)s * 64u < nA) {
          nvis += wave_sum(have ? steps + 1u : 0u);
          xa += wave_sum(have ? min(steps + 1u, (uint32_t)gd.right_half - al[s]) : 0u);
        }
      }
      // the merged intervals [alo, ahi]: lane l of set s holds the (64 s + l)-th; lanes without one an empty interval
      const uint64_t* ivl = (const uint64_t*)(aq0 + 2u * ACAP);
#pragma unroll
      for (int s = 0; s < G2S_SEG_ASETS; s++) {
        const uint32_t e = (uint32_t)s * 64u + (uint32_t)lane;
        const uint64_t x = (!overflow && e < nM) ? ivl[e] : 1ull << 32;
        alo[s] = (uint32_t)(x >> 32);
        ahi[s] = (uint32_t)x;
      }
      lds_sync();
    }
  };
  unsigned long long cyc_a_end = cyc0;
#ifdef G2S_SEG_PROFILE
  uint32_t prof_a[4] = {0, 0, 0, 0};  // phase A (one wave per gap): records wait | probes + atomics | results, slow path, queue | rest of the round
#endif
  if constexpr (!BIG) {
   if (!TWO || wave == 1) {
    for (uint32_t i = (uint32_t)lane; i < ALAB; i += 64u) lab[i] = G2S_DEV_EMPTY64;
    lds_sync();
    auto a_hash = [&](uint32_t p) -> uint32_t { uint32_t x = p; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x & (ALAB - 1u); };
    // propose label dp for node p (per lane) from table position h on; true when the label improved (the
    // caller queues p); *at = where p sits in the table
    auto relabel = [&](bool active, uint32_t p, uint32_t dp, uint32_t h, uint32_t* at) -> bool {
      bool improved = false, fresh = false;
      if (active) {
        const uint64_t key = ((uint64_t)p << 32) | dp;
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
          h = (h + 1u) & (ALAB - 1u);
        }
        *at = h;
      }
      nA += (uint32_t)__popcll(__ballot(fresh));
      return improved;
    };
    // the queues hold (node, table position) pairs: q[2][ACAP] nodes, then q[2][ACAP] positions
    uint32_t* aqs = aq0 + 2u * ACAP;
    if (!overflow) {
      uint32_t cur = 0, ne = 0;
      {  // seeds: right.substr(len-k-j, k) enters at depth j (:878-884, :953-976)
        const uint32_t sd = f_r;
        uint32_t at = 0;
        const bool imp = relabel(sd != G2S_DEV_INVALID && lane <= gd.right_half, sd, (uint32_t)lane, a_hash(sd), &at);
        const uint64_t m = __ballot(imp);
        if (imp) { aq0[(uint32_t)__popcll(m & below(lane))] = sd; aqs[(uint32_t)__popcll(m & below(lane))] = at; }
        if constexpr (TWO) {
          if (imp) {
            const uint4* u = (const uint4*)(urec + (size_t)(sd ^ 1u) * 8);
            qrec[(uint32_t)__popcll(m & below(lane))] = u[0];
            qrem[(uint32_t)__popcll(m & below(lane))] = u[1].x;
          }
        }
        ne = (uint32_t)__popcll(m);
        lds_sync();
      }
      while (ne > 0 && !overflow) {
        roundsA++;
        const uint32_t* qc = aq0 + cur * ACAP;
        const uint32_t* qcs = aqs + cur * ACAP;
        uint32_t* qn = aq0 + (cur ^ 1u) * ACAP;
        uint32_t* qns = aqs + (cur ^ 1u) * ACAP;
        const uint4* qcr = qrec + cur * ACAP;
        const uint32_t* qcm = qrem + cur * ACAP;
        uint4* qnr = qrec + (cur ^ 1u) * ACAP;
        uint32_t* qnm = qrem + (cur ^ 1u) * ACAP;
        uint32_t nn = 0;
#ifdef G2S_SEG_PROFILE
        unsigned long long pa_round = __builtin_amdgcn_s_memtime(), pa_in = 0;
#endif
        // Lane = (entry, predecessor slot): sixteen entries and their four proposals each per pass.  (Lane = entry
        // with a loop over the four slots spent 70 % of a round's 3.7-4.4 k cycles in four rounds of hashing,
        // probing, ballots and queue appends for the 3-4 entries a round has on average.)
        for (uint32_t e0 = 0; e0 < ne && !overflow; e0 += 16u) {
#ifdef G2S_SEG_PROFILE
          const unsigned long long pa0 = __builtin_amdgcn_s_memtime();
#endif
          const uint32_t eidx = e0 + ((uint32_t)lane >> 2), q = (uint32_t)lane & 3u;
          const bool mine = eidx < ne;
          const uint32_t v = mine ? qc[eidx] : 0u;
          const uint32_t slot = mine ? qcs[eidx] : 0u;
          // walking back from v = walking on from v^1: steps left in the unitig and the successor record
          // of the walk's last node (graph.predecessors(last)[i] = succ(last^1)[i] ^ 1) in one record,
          // asked for before the entry's label is read (the label comes from LDS while the record travels)
          uint32_t w = G2S_DEV_INVALID, r = 0;
          if (mine) {
            if constexpr (TWO) {
              w = ((const uint32_t*)(qcr + eidx))[q];
              r = qcm[eidx];
            } else {
              const uint32_t* u = urec + (size_t)(v ^ 1u) * 8;
              w = u[q];
              r = u[4];
            }
          }
#ifdef G2S_SEG_PROFILE
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const unsigned long long pa1 = __builtin_amdgcn_s_memtime();
#endif
          const uint32_t d = mine ? (uint32_t)lab[slot] : 0u;  // the entry's current label
          if (mine && q == 0u) labrem[slot] = r;
          const uint32_t steps = min(r, (uint32_t)gd.right_half - d);
          const bool live = mine && d + steps < (uint32_t)gd.right_half;  // (then steps == r: the record is the last node's)
          const uint32_t dchild = d + steps + 1u;
          // the proposal of this lane: first probe, then the atomic that fits what the probe saw; the rare collision
          // takes the general loop
          const bool act = live && w != G2S_DEV_INVALID;
          const uint32_t pnode = w ^ 1u;
          // (two waves: the proposed node's own record, for the round in which it would be an entry — see qrec)
          uint4 prec = make_uint4(G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID);
          uint32_t prem = 0u;
          if constexpr (TWO) {
            if (act) {
              const uint4* u = (const uint4*)(urec + (size_t)w * 8);  // (pnode ^ 1 = w)
              prec = u[0];
              prem = u[1].x;
            }
          }
          const uint64_t key = ((uint64_t)pnode << 32) | dchild;
          const uint32_t hq = a_hash(pnode);
          const uint64_t cq = act ? lab[hq] : 0ull;
          uint64_t oldmin = 0ull, oldcas = 0ull;
          const bool hit = act && (uint32_t)(cq >> 32) == pnode;
          if (hit) oldmin = atomicMin((unsigned long long*)&lab[hq], (unsigned long long)key);
          if (act && cq == G2S_DEV_EMPTY64)
            oldcas = atomicCAS((unsigned long long*)&lab[hq], (unsigned long long)G2S_DEV_EMPTY64, (unsigned long long)key);
#ifdef G2S_SEG_PROFILE
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const unsigned long long pa2 = __builtin_amdgcn_s_memtime();
#endif
          const bool claimed = act && cq == G2S_DEV_EMPTY64 && oldcas == G2S_DEV_EMPTY64;
          bool imp = (hit && oldmin > key) || claimed;
          nA += (uint32_t)__popcll(__ballot(claimed));
          uint32_t at = hq;
          const bool slow = act && !hit && !claimed;  // another node there, or the slot was taken meanwhile
          if (__ballot(slow)) { if (relabel(slow, pnode, dchild, hq, &at)) imp = true; }
          const uint64_t m = __ballot(imp);
          if (imp) {
            const uint32_t pos = nn + (uint32_t)__popcll(m & below(lane));
            if (pos < ACAP) {
              qn[pos] = pnode; qns[pos] = at;
              if constexpr (TWO) { qnr[pos] = prec; qnm[pos] = prem; }
            }
          }
          nn += (uint32_t)__popcll(m);
          // (the table has 2 ACAP slots: at most ACAP + 64 are ever taken, so every probe ends)
          if (nn > ACAP || nA > ACAP) { overflow = true; flags |= G2S_DEV_OVERFLOW_A | G2S_DEV_WHY_RS; }
#ifdef G2S_SEG_PROFILE
          const unsigned long long pa3 = __builtin_amdgcn_s_memtime();
          prof_a[0] += (uint32_t)(pa1 - pa0); prof_a[1] += (uint32_t)(pa2 - pa1); prof_a[2] += (uint32_t)(pa3 - pa2);
          pa_in += pa3 - pa0;
#endif
        }
        lds_sync();
#ifdef G2S_SEG_PROFILE
        prof_a[3] += (uint32_t)(__builtin_amdgcn_s_memtime() - pa_round - pa_in);
#endif
        cur ^= 1u;
        ne = nn;
      }
    }
    // the table's entries, packed into the (idle) queues
    if (!overflow) {
      uint32_t* cnode = aq0;
      uint32_t* clab = aq0 + ACAP;
      uint32_t* crem = (uint32_t*)lab;       // written only after the whole table was read: see the two loops
      uint32_t got = 0;
      uint32_t keep_n[ALAB / 64u], keep_l[ALAB / 64u], keep_r[ALAB / 64u], keep_at[ALAB / 64u];
#pragma unroll
      for (uint32_t c = 0; c < ALAB / 64u; c++) {
        const uint64_t e = lab[c * 64u + (uint32_t)lane];
        const bool have = e != G2S_DEV_EMPTY64;
        const uint64_t m = __ballot(have);
        keep_n[c] = (uint32_t)(e >> 32); keep_l[c] = (uint32_t)e; keep_r[c] = labrem[c * 64u + (uint32_t)lane];
        keep_at[c] = have ? got + (uint32_t)__popcll(m & below(lane)) : G2S_DEV_INVALID;
        got += (uint32_t)__popcll(m);
      }
      lds_sync();
#pragma unroll
      for (uint32_t c = 0; c < ALAB / 64u; c++)
        if (keep_at[c] != G2S_DEV_INVALID) { cnode[keep_at[c]] = keep_n[c]; clab[keep_at[c]] = keep_l[c]; crem[keep_at[c]] = keep_r[c]; }
      lds_sync();
    }
    // The entries as k-mer index intervals, sorted and merged in LDS (holes between any two): phase B's pruning test
    // and the lengths of its runs then take ONE look at the list instead of following chains of overlapping entries,
    // and the list is shorter (fewer 64-entry sets to ballot over per child).  Q7 in the right set, conservatively as
    // in the LDS tier — both strands of some k-mer are in it = an entry of each orientation with overlapping
    // intervals — falls out of the merge.
    if (!overflow && nA > 0u) {
      const uint32_t* cnode = aq0;
      const uint32_t* clab = aq0 + ACAP;
      const uint32_t* crem = (const uint32_t*)lab;
      uint64_t* ivl = (uint64_t*)(aq0 + 2u * ACAP);  // (the table-position queues are idle)
      uint32_t n2 = 2u;
      while (n2 < nA) n2 <<= 1;
      for (uint32_t e = (uint32_t)lane; e < n2; e += 64u) {
        uint64_t key = ~0ull;
        if (e < nA) {
          const uint32_t v = cnode[e];
          const uint32_t steps = min(crem[e], (uint32_t)gd.right_half - clab[e]);
          const uint32_t w0 = v ^ 1u, idx = w0 >> 1;
          const uint32_t lo = (w0 & 1u) ? idx - steps : idx, hi = (w0 & 1u) ? idx : idx + steps;
          key = ((uint64_t)lo << 32) | ((uint64_t)(v & 1u) << 31) | hi;
        }
        ivl[e] = key;
      }
      lds_sync();
      lds_sort64(ivl, n2, lane);
      bool cross = false;
      nM = lds_merge_intervals(ivl, nA, lane, &cross);
      if (cross) flags |= G2S_DEV_Q7_A;
      lds_sync();
    }
    cyc_a_end = __builtin_amdgcn_s_memtime();
    if constexpr (TWO) {  // wave 1 is done: what wave 0 needs to know, then the barrier it waits at
      if (lane == 0) {
        ash[0] = nA; ash[1] = roundsA; ash[2] = flags; ash[3] = overflow ? 1u : 0u; ash[9] = nM;
        ash[4] = (uint32_t)((cyc_a_end - cyc0) >> 8);
      }
      __syncthreads();
      // ...and stays for one more job: the Q7 check between the segments, when wave 0 is through phase B (second
      // barrier), in this wave's own LDS region (the table is free: the right set went into wave 0's registers); the
      // verdict waits for wave 0 behind a third barrier
      __syncthreads();
      // (and, while wave 0 sweeps the closure: do two segments share a k-mer at all? — if not, phase D2 skips its
      // pairwise test of the closure's extents)
      bool f7 = false, apart = false;
      if (ash[7]) seg_cross_check(s_node, s_dl, ash[6], (uint64_t*)lab, CAP / 2u, lane, !A.skip_confident && ash[6] <= 192u, &f7, &apart);
      if (lane == 0) { ash[5] = f7 ? 1u : 0u; ash[8] = apart ? 1u : 0u; }
      __syncthreads();
      return;
    }
   }
   if constexpr (!TWO) load_right_set();
  } else {
    // ---- (BIG) the same label-correcting search with room for G2S_SEGX_EA entries.
    // LDS: tab[AS] u64 (node << 32 | label); the two queues live in the workgroup's scratch.
    constexpr uint32_t AS = G2S_SEGX_AS, EA = G2S_SEGX_EA, QCAP = G2S_SEGX_QCAP;
    uint64_t* tab = (uint64_t*)lds;
    uint32_t* gq = scr + 6u * G2S_SEGX_CAP;
    bool stuck = false;  // (per lane) a probe ran past its bound
    for (uint32_t i = (uint32_t)lane; i < AS; i += 64u) tab[i] = SEGX_EMPTY64;
    lds_sync();
    auto a_hash = [&](uint32_t p) -> uint32_t { uint32_t x = p; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x & (AS - 1u); };
    auto relabel = [&](bool active, uint32_t p, uint32_t dp) -> bool {
      bool improved = false, fresh = false;
      if (active) {
        const uint64_t key = ((uint64_t)p << 32) | dp;
        uint32_t h = a_hash(p);
        uint32_t guard = 0;
        while (true) {
          if (++guard > 4u * AS) { stuck = true; break; }
          const uint64_t c = tab[h];
          if ((uint32_t)(c >> 32) == p) {
            improved = atomicMin((unsigned long long*)&tab[h], (unsigned long long)key) > key;
            break;
          }
          if (c == SEGX_EMPTY64) {
            const unsigned long long prev =
                atomicCAS((unsigned long long*)&tab[h], (unsigned long long)SEGX_EMPTY64, (unsigned long long)key);
            if (prev == SEGX_EMPTY64) { improved = true; fresh = true; break; }
            continue;
          }
          h = (h + 1u) & (AS - 1u);
        }
      }
      nA += (uint32_t)__popcll(__ballot(fresh));
      return improved;
    };
    if (!overflow) {
      uint32_t cur = 0, ne = 0;
      {
        const uint32_t sd = (lane <= rmf && lane < 32) ? rseeds[lane] : G2S_DEV_INVALID;
        const bool imp = relabel(sd != G2S_DEV_INVALID && lane <= gd.right_half, sd, (uint32_t)lane);
        const uint64_t m = __ballot(imp);
        if (imp) gq[(uint32_t)__popcll(m & below(lane))] = sd;
        ne = (uint32_t)__popcll(m);
      }
      while (ne > 0 && !overflow) {
        roundsA++;
        if (roundsA > 65535u) { overflow = true; flags |= G2S_DEV_OVERFLOW_A | G2S_DEV_WATCHDOG; break; }
        // (the queue was written by this wave; lines of it this compute unit read for an earlier gap go)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const uint32_t* qc = gq + cur * QCAP;
        uint32_t* qn = gq + (cur ^ 1u) * QCAP;
        uint32_t nn = 0;
        for (uint32_t e0 = 0; e0 < ne && !overflow; e0 += 64u) {
          const bool mine = e0 + (uint32_t)lane < ne;
          const uint32_t v = mine ? qc[e0 + (uint32_t)lane] : 0u;
          uint32_t d = 0;
          if (mine) {
            uint32_t slot = a_hash(v), guard = 0;
            while ((uint32_t)(tab[slot] >> 32) != v) {
              if (++guard > AS) { stuck = true; break; }
              slot = (slot + 1u) & (AS - 1u);
            }
            d = (uint32_t)tab[slot];
          }
          if (__ballot(stuck)) { overflow = true; flags |= G2S_DEV_OVERFLOW_A | G2S_DEV_WATCHDOG; break; }
          uint4 rec = make_uint4(G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID);
          uint32_t r = 0;
          if (mine) {
            const uint4* u = (const uint4*)(urec + (size_t)(v ^ 1u) * 8);
            rec = u[0];
            r = u[1].x;
          }
          const uint32_t steps = min(r, (uint32_t)gd.right_half - d);
          const bool live = mine && d + steps < (uint32_t)gd.right_half;
          if (!live) rec = make_uint4(G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID);
          const uint32_t dchild = d + steps + 1u;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const uint32_t w = q == 0 ? rec.x : q == 1 ? rec.y : q == 2 ? rec.z : rec.w;
            const bool imp = relabel(w != G2S_DEV_INVALID && !overflow, w ^ 1u, dchild);
            const uint64_t m = __ballot(imp);
            if (imp) {
              const uint32_t at = nn + (uint32_t)__popcll(m & below(lane));
              if (at < QCAP) qn[at] = w ^ 1u;
            }
            nn += (uint32_t)__popcll(m);
            // (at most EA + 64 of the AS slots are ever taken, so every probe ends)
            if (nn > QCAP || nA > EA) { overflow = true; flags |= G2S_DEV_OVERFLOW_A | G2S_DEV_WHY_RS; }
            if (__ballot(stuck)) { overflow = true; flags |= G2S_DEV_OVERFLOW_A | G2S_DEV_WATCHDOG; }
          }
        }
        lds_sync();
        cur ^= 1u;
        ne = nn;
      }
    }
    if (!overflow) {
      // the entries, packed to the front of the table (in place: the write cursor never passes the read cursor)
      uint32_t got = 0;
      for (uint32_t c = 0; c < AS; c += 64u) {
        const uint64_t e = tab[c + (uint32_t)lane];
        const bool have = e != SEGX_EMPTY64;
        const uint64_t m = __ballot(have);
        if (have) tab[got + (uint32_t)__popcll(m & below(lane))] = e;
        got += (uint32_t)__popcll(m);
      }
      lds_sync();
      // entry -> k-mer index interval (lo << 32 | orientation << 31 | hi): one record load each
      uint32_t acc_vis = 0, acc_xa = 0;
      uint32_t* o = dbg ? dbg + (size_t)x * dbg_words : nullptr;
#pragma unroll 2
      for (uint32_t e0 = 0; e0 < nA; e0 += 64u) {
        const uint32_t e = e0 + (uint32_t)lane;
        const bool have = e < nA;
        const uint64_t ent = have ? tab[e] : 0ull;
        const uint32_t v = (uint32_t)(ent >> 32), label = (uint32_t)ent;
        const uint32_t r = have ? urec[(size_t)(v ^ 1u) * 8 + 4] : 0u;
        const uint32_t steps = have ? min(r, (uint32_t)gd.right_half - label) : 0u;
        const uint32_t w0 = v ^ 1u, idx = w0 >> 1;
        const uint32_t lo = (w0 & 1u) ? idx - steps : idx, hi = (w0 & 1u) ? idx : idx + steps;
        if (have) {
          tab[e] = ((uint64_t)lo << 32) | ((uint64_t)(v & 1u) << 31) | hi;
          acc_vis += steps + 1u;
          acc_xa += min(steps + 1u, (uint32_t)gd.right_half - label);
          if (o && 9u + 2u * e < dbg_words) { o[8u + 2u * e] = v; o[9u + 2u * e] = label; }
        }
      }
      nvis = wave_sum(acc_vis);
      xa = wave_sum(acc_xa);
      uint32_t n2 = 2;
      while (n2 < nA) n2 <<= 1;
      for (uint32_t i = nA + (uint32_t)lane; i < n2; i += 64u) tab[i] = SEGX_EMPTY64;
      lds_sync();
      lds_sort64(tab, n2, lane);
      // Q7 in the right set, conservatively as above: an entry of each orientation with overlapping intervals
      bool cross = false;
      M = lds_merge_intervals(tab, nA, lane, &cross);
      if (cross) flags |= G2S_DEV_Q7_A;
      ivP = M ? 1u << (31 - __builtin_clz(M)) : 0u;
    }
  }
  // k-mer index x in the right set?  (wave-uniform; BIG: iv_find below)
  auto contains = [&](uint32_t x) -> bool {
    uint64_t m = 0;
#pragma unroll
    for (int s = 0; s < G2S_SEG_ASETS; s++)
      if ((uint32_t)s * 64u < nM) m |= __ballot(alo[s] <= x) & __ballot(x <= ahi[s]);
    return m != 0;
  };
  // largest y <= xmax with [x0, y] inside the right set (x0 - 1 when x0 is not in it): the merged interval that holds
  // x0 ends where the set's coverage ends
  auto covered_up = [&](uint32_t x0, uint32_t xmax) -> uint32_t {
    uint32_t h = 0;
    bool hit = false;
#pragma unroll
    for (int s = 0; s < G2S_SEG_ASETS; s++) {
      if ((uint32_t)s * 64u < nM) {
        const uint64_t m = __ballot(alo[s] <= x0) & __ballot(x0 <= ahi[s]);
        if (m) { h = rl(ahi[s], __builtin_ctzll(m)); hit = true; }
      }
    }
    return hit ? min(h, xmax) : x0 - 1u;
  };
  // smallest y >= xmin with [y, x0] inside the right set (x0 + 1 when x0 is not in it)
  auto covered_down = [&](uint32_t x0, uint32_t xmin) -> uint32_t {
    uint32_t l = 0;
    bool hit = false;
#pragma unroll
    for (int s = 0; s < G2S_SEG_ASETS; s++) {
      if ((uint32_t)s * 64u < nM) {
        const uint64_t m = __ballot(alo[s] <= x0) & __ballot(x0 <= ahi[s]);
        if (m) { l = rl(alo[s], __builtin_ctzll(m)); hit = true; }
      }
    }
    return hit ? max(l, xmin) : x0 + 1u;
  };
  const unsigned long long cyc1 = TWO ? cyc0 : __builtin_amdgcn_s_memtime();
  uint32_t cyc_a_kc = (uint32_t)((cyc_a_end - cyc0) >> 8);
  // (two waves) phase B starts without the right set; the first round that needs it waits for wave 1
  bool have_rs = !TWO;
#ifdef G2S_SEG_PROFILE
  uint32_t prof_wait = 0, prof_load = 0;  // (two waves) cycles at the barrier, and loading the right set behind it
#endif
  auto take_right_set = [&]() {
    if constexpr (TWO) {
#ifdef G2S_SEG_PROFILE
      const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
      __syncthreads();
#ifdef G2S_SEG_PROFILE
      const unsigned long long tw1 = __builtin_amdgcn_s_memtime();
#endif
      nA = ash[0]; roundsA = ash[1]; flags |= ash[2]; nM = ash[9];
      if (ash[3]) overflow = true;
      cyc_a_kc = ash[4];
      load_right_set();
      have_rs = true;
#ifdef G2S_SEG_PROFILE
      prof_wait = (uint32_t)(tw1 - tw0);
      prof_load = (uint32_t)(__builtin_amdgcn_s_memtime() - tw1);
#endif
    }
  };

  // ---------------- phase B: entry events in registers, segments into LDS ----------------------
  uint32_t en = G2S_DEV_INVALID, ec = 0, es = 1, ep01 = 0xFFFFFFFFu, ep23 = 0xFFFFFFFFu;
  uint4 erec = make_uint4(G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID);  // where the event's segment leaves
  constexpr int SEG_NOEV = 0x3FFFFFFF;  // depth of a lane without an event
  int ed = SEG_NOEV;
  uint64_t ev = 0, efx = 0;  // pending events, and which of them have an assigned count (left seeds)
  uint32_t nseg = 0, gen = 0, xb = 0, sb = 0;
#ifdef G2S_SEG_PROFILE
  unsigned long long prof_t[5] = {0, 0, 0, 0, 0}, prof_tail[5] = {0, 0, 0, 0, 0};
  uint32_t prof_acc[4] = {0, 0, 0, 0};
#endif
  uint32_t best = SEG_INF, c1 = 0, c2 = 0;  // phase C: (found level << 6 | j) and the counts of its two lengths
  // Stop depths of the traceback (:1455-1462): a traceback that passes through an entry stops at one of
  // the depths of the left-flank k-mers reachable backwards from it.  est = lowest | highest << 16 of
  // them per pending event: a source entry (depth <= lmf, the left-flank k-mer of that offset) has its
  // own depth, any other the union over its parents.  (0x7FFF, 0) = no parent yet, (0, 0x7FFF) = unknown;
  // both never read as "one depth".  s1 / s2 keep the values of the entries behind c1 / c2.
  uint32_t est = 0x7FFFu, s1 = 0x7FFFu, s2 = 0x7FFFu;
  auto stop_join = [](uint32_t a, uint32_t b) -> uint32_t { return min(a & 0xFFFFu, b & 0xFFFFu) | (max(a >> 16, b >> 16) << 16); };
  auto is_source = [&](uint32_t w, int dw) -> bool {  // (wave-uniform) :1270, k-mer comparison only
    if (dw > lmf) return false;
    const uint32_t ls = uni(l_seed[dw]);
    return ls != G2S_DEV_INVALID && (w >> 1) == (ls >> 1);
  };
  // One child (w, dw) with count c from segment par: merged into the pending event of its state, or a new event in a
  // free lane.  Which of the two it is, and which lane changes, are wave-uniform: two scalar branches, and inside them
  // every register of the event takes its new value through a select (the version with `if (lane == l) { ... }`
  // around the updates saved and restored the execution mask around three levels of cases and copied every event
  // register at every join).  A ballot is taken of ONE comparison at a time and the masks are combined as scalars: a
  // ballot of a conjunction goes through a vector register (0 / 1, compared again).  (A lane without an event carries
  // depth SEG_NOEV: it matches no child.)
  auto add_event = [&](uint32_t w, int dw, uint32_t c, uint32_t par, uint32_t pstop) {
    const uint64_t md = __ballot(ed == dw);
    const uint64_t m = md & __ballot(en == w);
    const bool src = is_source(w, dw);
    if (m) {
      const bool me = lane == __builtin_ctzll(m);
      if (!src) est = me ? stop_join(est, pstop) : est;
      ec = me ? min(ec + c, (uint32_t)G2S_DEV_MAX_PATHS) : ec;  // saturating add is associative (:1058-1060)
      const bool free1 = (ep01 >> 16) == SEG_NOPAR, free2 = (ep23 & 0xFFFFu) == SEG_NOPAR;
      const uint32_t a01 = (ep01 & 0xFFFFu) | (par << 16);
      const uint32_t a23 = free2 ? ((ep23 & 0xFFFF0000u) | par) : ((ep23 & 0xFFFFu) | (par << 16));
      ep01 = (me && free1) ? a01 : ep01;
      ep23 = (me && !free1) ? a23 : ep23;
      return;
    }
    if (md & __ballot(en == (w ^ 1u))) flags |= G2S_DEV_Q7_B;  // the other strand at this depth
    const uint64_t fr = ~ev;
    if (!fr) { overflow = true; flags |= G2S_DEV_OVERFLOW_B | G2S_DEV_WHY_FRONTIER; return; }
    const int l = __builtin_ctzll(fr);
    const bool me = lane == l;
    en = me ? w : en;
    ed = me ? dw : ed;
    ec = me ? c : ec;
    ep01 = me ? (0xFFFF0000u | par) : ep01;
    ep23 = me ? 0xFFFFFFFFu : ep23;
    // states up to the end of the unitig: loaded with the exit record for all new events at once, at the head of the next
    // round.  (Round 6, measured and not kept: the record asked for HERE, by the event's lane.  The compiler waits for
    // it on the spot — the loaded registers are event state that every join of this lambda copies — and the launch got
    // slower, g2s_fill_seg2 0.102 -> 0.104 ms, g2s_fill_seg on 10 000 gaps 0.281 -> 0.320 ms; and a load the compiler
    // does not see could only pass under the round's remaining children: the next round begins with the horizon over
    // this very value.  LAB_NOTES, round 6.)
    es = me ? 0u : es;
    est = me ? (src ? ((uint32_t)dw | ((uint32_t)dw << 16)) : pstop) : est;
    ev |= 1ull << l;
  };
  if constexpr (!BIG) {
    if (!overflow) {
      // left seeds: left.substr(d, k) enters at depth d with the value 1 ASSIGNED (:995-1015, :1082-1105)
      const uint32_t sd = f_l;
      const uint32_t s0 = rl(sd, 0);
      // The usual flank is a stretch of ONE unitig: seed d is the d-th node after seed 0 and the walk
      // from seed 0 stays unitig-internal for lmf steps.  Levels 0 .. lmf-1 are then that chain with
      // count 1 (every state has the next seed as its only successor, and a seed's value is 1 anyway):
      // one segment, and the seed at depth lmf as the only pending event, instead of lmf rounds.
      bool chain = lmf >= 1 && s0 != G2S_DEV_INVALID && __ballot(lane <= lmf && sd != seg_node(s0, (uint32_t)lane)) == 0ull;
      if (chain) chain = uni(urec[(size_t)s0 * 8 + 4]) >= (uint32_t)lmf;
      // (a target k-mer inside the chain could make one of its states a sink or a traceback start: every
      // state there is a source of its own, which only single-state segments express: the general path then)
      if (chain) chain = __ballot(seg_pos(s0, (uint32_t)lmf, tg) >= 0) == 0ull;
      if (chain) {
        if (lane == 0) {
          s_node[0] = s0; s_dl[0] = (uint32_t)lmf << 16; s_cnt[0] = 1u; s_p01[0] = s_p23[0] = 0xFFFFFFFFu; s_aux[0] = 0u;
          s_t[0] = 0x7FFFu;  // (holds no target k-mer: checked above)
        }
        nseg = 1; gen = 1;
        sb += (uint32_t)lmf;
        xb += (uint32_t)lmf;
        ev = efx = 1ull << lmf;
        if (lane == lmf) { en = sd; ed = lmf; ec = 1; ep01 = 0xFFFF0000u; ep23 = 0xFFFFFFFFu; es = 0u; est = (uint32_t)lmf | ((uint32_t)lmf << 16); }
      } else {
        ev = efx = __ballot(sd != G2S_DEV_INVALID && lane <= D);
        if ((ev >> lane) & 1ull) { en = sd; ed = lane; ec = 1; ep01 = ep23 = 0xFFFFFFFFu; es = 0u; est = (uint32_t)lane | ((uint32_t)lane << 16); }
      }
    }
    // (two waves: the rounds in two stretches — up to the round that first needs the right set, and from there on — so
    // that the 21 registers it lives in change between two loops, not inside one: with take_right_set() in the loop the
    // compiler copied them at the head and at the end of EVERY round, 40-odd moves of a round's ~800 instructions)
    for (int stretch = 0; stretch < (TWO ? 2 : 1); stretch++) {
    bool want_rs = false;
    while (ev && !overflow) {
      SEG_PROF_T(0);
      if (((ev >> lane) & 1ull) && es == 0u) {  // one round trip for all events created last round
        if (ed < lmf) { es = 1u; erec = *(const uint4*)(succ + (size_t)en * 4); }  // above the flank: one state, leaves at once
        else { const uint4* u = (const uint4*)(urec + (size_t)en * 8); erec = u[0]; es = u[1].x + 1u; }
      }
      SEG_PROF_WAIT();
      SEG_PROF_T(1);
      // ---- which events are final: depth below the horizon
      const bool valid = (ev >> lane) & 1ull;
      const uint32_t H = wave_min(valid ? (uint32_t)ed + es : SEG_INF);
      const uint64_t sel = ev & __ballot((uint32_t)ed < H);
      const uint32_t nsel = (uint32_t)__popcll(sel);
      if (nseg + nsel > G2S_SEG_CAP) { overflow = true; flags |= G2S_DEV_OVERFLOW_B | G2S_DEV_WHY_LOG; break; }
      const bool mine = (sel >> lane) & 1ull;
      const uint32_t cnt = ((efx >> lane) & 1ull) ? 1u : ec;
      const uint32_t lcap = min(es, (uint32_t)(D - ed + 1));
      uint32_t elen = lcap;
      const uint32_t esid = nseg + (uint32_t)__popcll(sel & below(lane));
      if constexpr (TWO) {  // a state or a child at or beyond the depth the pruning rule starts at: the right set now
        // (nothing of this round has been written yet: it starts again behind the barrier)
        if (!have_rs && (sel & __ballot(ed + (int)lcap >= gd.prune_from))) { want_rs = true; break; }
      }
      SEG_PROF_T(2);
      // ---- their lengths under the pruning rule, their target hits (one segment at a time, wave-uniform)
      for (uint64_t m = sel; m; m &= m - 1) {
        const int l = __builtin_ctzll(m);
        const uint32_t node = rl(en, l), lc = rl(lcap, l), c = rl(cnt, l), stl = rl(est, l);
        const int depth = (int)rl((uint32_t)ed, l);
        uint32_t L = lc;
        if (lc > 1u && depth + (int)lc - 1 >= gd.prune_from) {  // interior states are entered under :1050
          const uint32_t t1 = (uint32_t)max(1, gd.prune_from - depth);
          const uint32_t idx0 = node >> 1;
          if (!(node & 1u)) L = covered_up(idx0 + t1, idx0 + lc - 1u) - idx0 + 1u;
          else L = idx0 - covered_down(idx0 - t1, idx0 - (lc - 1u)) + 1u;
          if (lane == l) elen = L;
        }
        sb += L;
        xb += min(L, (uint32_t)(D - depth));
        (void)c; (void)stl;  // (the target hits of the segments are looked for behind the search, all segments at once)
      }
      if (mine) {
        s_node[esid] = en;
        s_dl[esid] = (uint32_t)ed | (elen << 16);
        s_cnt[esid] = cnt;
        s_p01[esid] = ep01;
        s_p23[esid] = ep23;
        s_aux[esid] = gen;
        s_t[esid] = est;  // (stop depths of the entry, until the hits have been looked for: phase D1 fills s_t later)
      }
      nseg += nsel;
      SEG_PROF_T(3);
      // ---- segments that reached the end of their stretch leave through the successor table
      const uint64_t exits_m = sel & ballot_and(elen == es, ed + (int)elen - 1 < D);
      const uint4 rec = erec;  // elen == lcap == es: the walk reached the node the record belongs to
      const uint32_t xd = (uint32_t)ed + elen;  // depth of the children
      const uint32_t est_sel = est;             // (add_event below may reuse a selected lane for a new event)
      ev &= ~sel;
      efx &= ~sel;
      ed = mine ? SEG_NOEV : ed;  // (the selected events' lanes are free: they match no child)
      // (successor slot by successor slot, only the lanes whose slot holds a node: the order in which the
      // children arrive does not matter — counts add up, the host puts parents into GATB order)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t wv = q == 0 ? rec.x : q == 1 ? rec.y : q == 2 ? rec.z : rec.w;
        for (uint64_t m = exits_m & __ballot(wv != G2S_DEV_INVALID); m && !overflow; m &= m - 1) {
          const int l = __builtin_ctzll(m);
          const uint32_t w = rl(wv, l);
          const int dw = (int)rl(xd, l);
          if (dw < gd.prune_from || contains(w >> 1)) add_event(w, dw, rl(cnt, l), rl(esid, l), rl(est_sel, l));  // :1050
        }
      }
      SEG_PROF_T(4);
      SEG_PROF_ACC();
      gen++;
    }
    if (!want_rs) break;
    take_right_set();
    }
    // ---- phase C's hits (:1107-1159): target k-mer j at position t of a segment is a hit at depth + t.  Behind the
    // search, lane = segment and a loop over the <= 32 targets: inside the rounds the same look — one selected event
    // at a time, wave-uniform — was a tenth of a round.  A hit (error, j) exists at most once above and once below its
    // base depth (a DP state is in one segment), so the smallest key and its two counts are a reduction.
    if (!overflow) {
      lds_sync();
      uint32_t mybest = SEG_INF, myc1 = 0, myc2 = 0, mys1 = 0x7FFFu, mys2 = 0x7FFFu;
      for (uint32_t b0 = 0; b0 < nseg; b0 += 64u) {
        const uint32_t b = b0 + (uint32_t)lane;
        const bool hb = b < nseg;
        const uint32_t node = hb ? s_node[b] : 0u, dl = hb ? s_dl[b] : 0u, c = hb ? s_cnt[b] : 0u, st = hb ? s_t[b] : 0x7FFFu;
        const uint32_t L = dl >> 16;
        const int depth = (int)(dl & 0xFFFFu);
        for (int j = 0; j <= rmf && j < 32; j++) {
          const int t = hb ? seg_pos(node, L, rl(tg, j)) : -1;
          if (t < 0) continue;
          const int td = depth + t, base = gd.g + lmf + j;
          const int err = td >= base ? td - base : base - td;
          if (err > gd.e) continue;
          const uint32_t key = ((uint32_t)(err + gd.g + lmf + rmf) << 6) | (uint32_t)j;
          if (key < mybest) { mybest = key; myc1 = 0; myc2 = 0; }
          if (key == mybest) { if (td >= base) { myc1 = c; mys1 = st; } else { myc2 = c; mys2 = st; } }
        }
      }
      best = wave_min(mybest);
      if (best != SEG_INF) {
        const uint64_t m1 = __ballot(mybest == best && myc1 != 0u), m2 = __ballot(mybest == best && myc2 != 0u);
        if (m1) { c1 = rl(myc1, __builtin_ctzll(m1)); s1 = rl(mys1, __builtin_ctzll(m1)); }
        if (m2) { c2 = rl(myc2, __builtin_ctzll(m2)); s2 = rl(mys2, __builtin_ctzll(m2)); }
      }
    }
  } else {
    // ---- (BIG) the same search with the pending events in LDS.  Lane = event only while a chunk of them is
    // looked at; an event lives in a SLOT (fields below), found by (node, depth) through an open-addressing
    // table ht (node << 32 | depth << 16 | slot).  Children of a whole chunk of final events are inserted
    // by all lanes at once: claim by compare-and-swap, counts merged with atomic adds (at most four
    // parents of at most 2^30 - 1 each: no wrap; clamped when read), stop depths with atomic min / max.
    // A selected event leaves a tombstone behind (no later proposal can carry its key: all have depths
    // at or above the horizon); the table is rebuilt from the pending list when tombstones pile up.
    // LDS (words): iv 2 EA | slots: node, depth|fixed<<15|parents<<16, count, p01, p23, stop lo, stop hi,
    //   states to the unitig's end, table position [PE each], exit record [4 PE] | ht [2 HS] |
    //   pending list x 2, free slots, new slots, final events of the round [PE each]
    constexpr uint32_t PE = G2S_SEGX_PE, HS = G2S_SEGX_HS;
    uint32_t* e_node = lds + 2u * G2S_SEGX_EA;
    uint32_t* e_dp = e_node + PE;
    uint32_t* e_cnt = e_dp + PE;
    uint32_t* e_p01 = e_cnt + PE;
    uint32_t* e_p23 = e_p01 + PE;
    uint32_t* e_slo = e_p23 + PE;
    uint32_t* e_shi = e_slo + PE;
    uint32_t* e_es = e_shi + PE;
    uint32_t* e_hpos = e_es + PE;
    uint4* e_rec = (uint4*)(e_hpos + PE);
    uint64_t* ht = (uint64_t*)(e_rec + PE);
    uint32_t* plist = (uint32_t*)(ht + HS);
    uint32_t* fstack = plist + 2u * PE;
    uint32_t* newl = fstack + PE;
    uint32_t* sell = newl + PE;
    static_assert(2u * G2S_SEGX_EA + 9u * PE + 4u * PE + 2u * HS + 5u * PE + 64u <= SEGX_LDS_WORDS, "LDS layout of the large variant");
    uint32_t np = 0, npn = 0, nnew = 0, nfree = PE, ntomb = 0, pcur = 0;
    uint32_t acc_sb = 0, acc_xb = 0;
    bool stuckb = false;  // (per lane) a probe ran past its bound
    for (uint32_t i = (uint32_t)lane; i < HS; i += 64u) ht[i] = SEGX_EMPTY64;
    for (uint32_t i = (uint32_t)lane; i < PE; i += 64u) fstack[i] = PE - 1u - i;  // slot 0 on top
    lds_sync();
    auto e_hash = [&](uint32_t w, uint32_t dw) -> uint32_t {
      uint32_t h = w * 0x9E3779B1u ^ dw * 0x85EBCA6Bu;
      h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
      return h & (HS - 1u);
    };
    // index of the last interval that begins at or before k-mer index q (-1: none), per lane
    auto iv_find = [&](uint32_t q) -> int {
      uint32_t pos = 0;
      for (uint32_t st = ivP; st; st >>= 1) {
        const uint32_t pp = pos + st;
        if (pp <= M && ivw[2u * (pp - 1u) + 1u] <= q) pos = pp;
      }
      return (int)pos - 1;
    };
    // proposals of all lanes: event (w, dw) gains count c from segment par (0xFFFF: a seed, no parent)
    auto ev_insert = [&](bool act, uint32_t w, uint32_t dw, uint32_t c, uint32_t par, uint32_t pslo, uint32_t pshi, bool fixed) {
      const uint64_t am = __ballot(act);
      const uint32_t na = (uint32_t)__popcll(am);
      if (na == 0u) return;
      if (nfree < na) { overflow = true; flags |= G2S_DEV_OVERFLOW_B | G2S_DEV_WHY_FRONTIER; return; }
      uint32_t slot = 0;
      if (act) slot = fstack[nfree - 1u - (uint32_t)__popcll(am & below(lane))];
      nfree -= na;
      bool src = false;
      if (act && dw <= (uint32_t)lmf) { const uint32_t ls = l_seed[dw]; src = ls != G2S_DEV_INVALID && (w >> 1) == (ls >> 1); }  // :1270
      const uint64_t kpart = ((uint64_t)w << 32) | ((uint64_t)dw << 16);
      if (act) {  // the candidate slot is complete before it can be seen through the table
        e_node[slot] = w;
        e_dp[slot] = dw | (fixed ? 0x8000u : 0u) | (par != SEG_NOPAR ? 0x10000u : 0u);
        e_cnt[slot] = c;
        e_p01[slot] = 0xFFFF0000u | par;
        e_p23[slot] = 0xFFFFFFFFu;
        e_slo[slot] = src ? dw : pslo;
        e_shi[slot] = src ? dw : pshi;
        e_es[slot] = 0u;
      }
      bool fresh = false, merged = false;
      uint32_t mslot = 0;
      uint32_t pos = act ? e_hash(w, dw) : 0u;
      int freepos = -1;
      uint32_t guard = 0;
      if (act) {  // among what was there before this call; the first free position of the probe sequence
        while (true) {
          if (++guard > 2u * HS) { stuckb = true; break; }
          const uint64_t cc = ht[pos];
          if (cc == SEGX_EMPTY64) break;
          if (cc == SEGX_TOMB64) { if (freepos < 0) freepos = (int)pos; }
          else if ((cc & ~0xFFFFull) == kpart) { merged = true; mslot = (uint32_t)cc & 0xFFFFu; break; }
          pos = (pos + 1u) & (HS - 1u);
        }
        if (freepos >= 0) pos = (uint32_t)freepos;
      }
      __builtin_amdgcn_wave_barrier();
      if (act && !merged && !stuckb) {  // claim it, or meet the lane that did with the same key
        while (true) {
          if (++guard > 8u * HS) { stuckb = true; break; }
          const uint64_t cc = ht[pos];
          if (cc == SEGX_EMPTY64 || cc == SEGX_TOMB64) {
            const unsigned long long prev = atomicCAS((unsigned long long*)&ht[pos], (unsigned long long)cc, (unsigned long long)(kpart | slot));
            if (prev == cc) { fresh = true; e_hpos[slot] = pos; break; }
            continue;
          }
          if ((cc & ~0xFFFFull) == kpart) { merged = true; mslot = (uint32_t)cc & 0xFFFFu; break; }
          pos = (pos + 1u) & (HS - 1u);
        }
      }
      if (merged) {
        atomicAdd(&e_cnt[mslot], c);
        if (par != SEG_NOPAR) {
          const uint32_t kk = (atomicAdd(&e_dp[mslot], 0x10000u) >> 16) & 0xFu;
          if (kk == 0u) atomicAnd(&e_p01[mslot], 0xFFFF0000u | par);
          else if (kk == 1u) atomicAnd(&e_p01[mslot], 0x0000FFFFu | (par << 16));
          else if (kk == 2u) atomicAnd(&e_p23[mslot], 0xFFFF0000u | par);
          else if (kk == 3u) atomicAnd(&e_p23[mslot], 0x0000FFFFu | (par << 16));
        }
        if (!src) { atomicMin(&e_slo[mslot], pslo); atomicMax(&e_shi[mslot], pshi); }
      }
      const uint64_t mm = __ballot(merged);
      if (merged) fstack[nfree + (uint32_t)__popcll(mm & below(lane))] = slot;  // the candidate slot was not needed
      nfree += (uint32_t)__popcll(mm);
      const uint64_t fm = __ballot(fresh);
      if (fresh) {
        const uint32_t at = (uint32_t)__popcll(fm & below(lane));
        plist[(pcur ^ 1u) * PE + npn + at] = slot;
        newl[nnew + at] = slot;
      }
      npn += (uint32_t)__popcll(fm);
      nnew += (uint32_t)__popcll(fm);
      lds_sync();
      // Q7: the other strand pending at this depth (looked up after the insertions of this call)
      bool other = false;
      if (fresh) {
        const uint64_t okey = ((uint64_t)(w ^ 1u) << 32) | ((uint64_t)dw << 16);
        uint32_t p2 = e_hash(w ^ 1u, dw), g2 = 0;
        while (true) {
          if (++g2 > 2u * HS) { stuckb = true; break; }
          const uint64_t cc = ht[p2];
          if (cc == SEGX_EMPTY64) break;
          if (cc != SEGX_TOMB64 && (cc & ~0xFFFFull) == okey) { other = true; break; }
          p2 = (p2 + 1u) & (HS - 1u);
        }
      }
      if (__ballot(other)) flags |= G2S_DEV_Q7_B;
      if (__ballot(stuckb)) { overflow = true; flags |= G2S_DEV_OVERFLOW_B | G2S_DEV_WATCHDOG; }
    };
    if (!overflow) {
      // left seeds: left.substr(d, k) enters at depth d with the value 1 ASSIGNED (:995-1015, :1082-1105)
      const uint32_t sd = lane <= lmf ? lseeds[lane] : G2S_DEV_INVALID;
      const uint32_t s0 = rl(sd, 0);
      bool chain = lmf >= 1 && s0 != G2S_DEV_INVALID && __ballot(lane <= lmf && sd != seg_node(s0, (uint32_t)lane)) == 0ull;
      if (chain) chain = uni(urec[(size_t)s0 * 8 + 4]) >= (uint32_t)lmf;
      if (chain) chain = __ballot(seg_pos(s0, (uint32_t)lmf, tg) >= 0) == 0ull;
      if (chain) {  // (see above: the flank as one segment, the seed at depth lmf the only pending event)
        if (lane == 0) {
          s_node[0] = s0; s_dl[0] = (uint32_t)lmf << 16; s_cnt[0] = 1u; s_p01[0] = s_p23[0] = 0xFFFFFFFFu; s_gen[0] = 0u;
        }
        nseg = 1; gen = 1;
        sb += (uint32_t)lmf;
        xb += (uint32_t)lmf;
        ev_insert(lane == lmf, sd, (uint32_t)lmf, 1u, 0u, (uint32_t)lmf, (uint32_t)lmf, true);
      } else {
        ev_insert(sd != G2S_DEV_INVALID && lane <= D, sd, (uint32_t)lane, 1u, SEG_NOPAR, (uint32_t)lane, (uint32_t)lane, true);
      }
      pcur ^= 1u; np = npn; npn = 0;
    }
    while (np > 0 && !overflow) {
      SEG_PROF_T(0);
      // ---- one round trip for all events created last round: states to the end of the unitig, exit record
      for (uint32_t i0 = 0; i0 < nnew; i0 += 64u) {
        if (i0 + (uint32_t)lane < nnew) {
          const uint32_t slot = newl[i0 + (uint32_t)lane];
          const uint32_t node = e_node[slot], dd = e_dp[slot] & 0x7FFFu;
          if ((int)dd < lmf) { e_rec[slot] = *(const uint4*)(succ + (size_t)node * 4); e_es[slot] = 1u; }  // above the flank: one state
          else { const uint4* u = (const uint4*)(urec + (size_t)node * 8); e_rec[slot] = u[0]; e_es[slot] = u[1].x + 1u; }
        }
      }
      nnew = 0;
      lds_sync();
      SEG_PROF_WAIT();
      SEG_PROF_T(1);
      // ---- the horizon
      const uint32_t* pl = plist + pcur * PE;
      uint32_t hmin = SEG_INF;
      for (uint32_t i0 = 0; i0 < np; i0 += 64u)
        if (i0 + (uint32_t)lane < np) { const uint32_t slot = pl[i0 + (uint32_t)lane]; hmin = min(hmin, (e_dp[slot] & 0x7FFFu) + e_es[slot]); }
      for (int o = 32; o > 0; o >>= 1) hmin = min(hmin, (uint32_t)__shfl_xor((int)hmin, o));
      const uint32_t H = uni(hmin);
      // ---- final events: gathered first (they are scattered over the pending list), then worked on in
      // full chunks — every chunk below costs a few thousand cycles whatever the number of its events
      uint32_t nsl = 0;
      for (uint32_t i0 = 0; i0 < np; i0 += 64u) {
        const bool have = i0 + (uint32_t)lane < np;
        const uint32_t slot = have ? pl[i0 + (uint32_t)lane] : 0u;
        const bool fin = have && (e_dp[slot] & 0x7FFFu) < H;
        const uint64_t fm = __ballot(fin), km = __ballot(have && !fin);
        if (fin) sell[nsl + (uint32_t)__popcll(fm & below(lane))] = slot;
        if (have && !fin) plist[(pcur ^ 1u) * PE + npn + (uint32_t)__popcll(km & below(lane))] = slot;  // the others stay pending
        nsl += (uint32_t)__popcll(fm);
        npn += (uint32_t)__popcll(km);
      }
      lds_sync();
      SEG_PROF_T(2);
#ifdef G2S_SEG_PROFILE
      unsigned long long prof_children = 0;
#endif
      for (uint32_t i0 = 0; i0 < nsl && !overflow; i0 += 64u) {
        const bool mine = i0 + (uint32_t)lane < nsl;
        const uint32_t slot = mine ? sell[i0 + (uint32_t)lane] : 0u;
        const uint32_t dp = mine ? e_dp[slot] : 0u;
        const int ed = (int)(dp & 0x7FFFu);
        const uint64_t sel = __ballot(mine);
        const uint32_t nsel = (uint32_t)__popcll(sel);
        if (nseg + nsel > G2S_SEGX_CAP) { overflow = true; flags |= G2S_DEV_OVERFLOW_B | G2S_DEV_WHY_LOG; break; }
        const uint32_t esid = nseg + (uint32_t)__popcll(sel & below(lane));
        const uint32_t en = mine ? e_node[slot] : 0u, es = mine ? e_es[slot] : 1u;
        const uint4 rec = mine ? e_rec[slot] : make_uint4(G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID);
        const uint32_t cnt = (dp & 0x8000u) ? 1u : min(mine ? e_cnt[slot] : 0u, (uint32_t)G2S_DEV_MAX_PATHS);
        const uint32_t slo = mine ? e_slo[slot] : 0x7FFFu, shi = mine ? e_shi[slot] : 0u;
        const uint32_t p01 = mine ? e_p01[slot] : 0xFFFFFFFFu, p23 = mine ? e_p23[slot] : 0xFFFFFFFFu;
        // the slots and table positions of the selected events are free again
        if (mine) { ht[e_hpos[slot]] = SEGX_TOMB64; fstack[nfree + (uint32_t)__popcll(sel & below(lane))] = slot; }
        nfree += nsel;
        ntomb += nsel;
        // ---- lengths under the pruning rule (:1050): every lane searches the intervals for its own event
        const uint32_t lcap = min(es, (uint32_t)(D - ed + 1));
        uint32_t L = lcap;
        {
          const bool pr = mine && lcap > 1u && ed + (int)lcap - 1 >= gd.prune_from;
          const uint32_t t1 = (uint32_t)max(1, gd.prune_from - ed);
          const uint32_t idx0 = en >> 1;
          const uint32_t q0 = (en & 1u) ? idx0 - t1 : idx0 + t1;  // first state entered under the rule
          const int f = iv_find(pr ? q0 : 0u);
          if (pr) {
            const bool in = f >= 0 && q0 <= ivw[2u * (uint32_t)f];
            if (!(en & 1u)) {
              const uint32_t y = in ? min(ivw[2u * (uint32_t)f], idx0 + lcap - 1u) : q0 - 1u;   // last covered index
              L = y - idx0 + 1u;
            } else {
              const uint32_t y = in ? max(ivw[2u * (uint32_t)f + 1u], idx0 - (lcap - 1u)) : q0 + 1u;  // first covered index
              L = idx0 - y + 1u;
            }
          }
        }
        if (mine) { acc_sb += L; acc_xb += min(L, (uint32_t)(D - ed)); }
        // ---- phase C: target k-mer j at position t of a segment is a hit at depth + t
        for (int j = 0; j <= rmf; j++) {
          const uint32_t tj = rl(tg, j);
          const int t = mine ? seg_pos(en, L, tj) : -1;
          for (uint64_t hm = __ballot(t >= 0); hm; hm &= hm - 1) {
            const int l = __builtin_ctzll(hm);
            const int td = (int)rl((uint32_t)ed, l) + (int)rl((uint32_t)t, l), base = gd.g + lmf + j;
            const int err = td >= base ? td - base : base - td;
            if (err > gd.e) continue;
            const uint32_t key = ((uint32_t)(err + gd.g + lmf + rmf) << 6) | (uint32_t)j;
            const uint32_t c = rl(cnt, l), st = rl(slo, l) | (rl(shi, l) << 16);
            if (key < best) { best = key; c1 = 0; c2 = 0; }
            if (key == best) { if (td >= base) { c1 = c; s1 = st; } else { c2 = c; s2 = st; } }
          }
        }
        if (mine) {
          s_node[esid] = en;
          s_dl[esid] = (uint32_t)ed | (L << 16);
          s_cnt[esid] = cnt;
          s_p01[esid] = p01;
          s_p23[esid] = p23;
          s_gen[esid] = gen;
        }
        nseg += nsel;
        // ---- segments that reached the end of their stretch leave through the successor table
        const bool exits = mine && L == es && ed + (int)L - 1 < D;
        const uint32_t xd = (uint32_t)ed + L;  // depth of the children
        // lane = (segment, successor slot): sixteen segments' children per pass, one search and one insertion each
#ifdef G2S_SEG_PROFILE
        const unsigned long long prof_c0 = __builtin_amdgcn_s_memtime();
#endif
        for (uint32_t g0 = 0; g0 < nsel && !overflow; g0 += 16u) {
          const int from = (int)(g0 + ((uint32_t)lane >> 2));
          const uint32_t q = (uint32_t)lane & 3u;
          const uint32_t wx = (uint32_t)__shfl((int)rec.x, from), wy = (uint32_t)__shfl((int)rec.y, from);
          const uint32_t wz = (uint32_t)__shfl((int)rec.z, from), ww = (uint32_t)__shfl((int)rec.w, from);
          const uint32_t w = q == 0u ? wx : q == 1u ? wy : q == 2u ? wz : ww;
          const bool ex = __shfl((int)exits, from) != 0;
          const uint32_t xdl = (uint32_t)__shfl((int)xd, from), cl = (uint32_t)__shfl((int)cnt, from);
          const uint32_t parl = (uint32_t)__shfl((int)esid, from);
          const uint32_t slol = (uint32_t)__shfl((int)slo, from), shil = (uint32_t)__shfl((int)shi, from);
          const int f = iv_find(w >> 1);
          const bool inset = f >= 0 && (w >> 1) <= ivw[2u * (uint32_t)f];
          const bool ok = ex && w != G2S_DEV_INVALID && ((int)xdl < gd.prune_from || inset);  // :1050
          ev_insert(ok, w, xdl, cl, parl, slol, shil, false);
        }
#ifdef G2S_SEG_PROFILE
        prof_children += __builtin_amdgcn_s_memtime() - prof_c0;
#endif
      }
#ifdef G2S_SEG_PROFILE
      // (sections: records wait | horizon + gather | lengths, hits, segment records | children)
      prof_t[3] = __builtin_amdgcn_s_memtime() - prof_children;
      prof_t[4] = prof_t[3] + prof_children;
      SEG_PROF_ACC();
#endif
      pcur ^= 1u;
      np = npn;
      npn = 0;
      gen++;
      // ---- tombstones pile up: rebuild the table from the pending list
      if (ntomb > HS / 4u && !overflow) {
        for (uint32_t i = (uint32_t)lane; i < HS; i += 64u) ht[i] = SEGX_EMPTY64;
        lds_sync();
        const uint32_t* pn = plist + pcur * PE;
        for (uint32_t i0 = 0; i0 < np; i0 += 64u) {
          if (i0 + (uint32_t)lane < np) {
            const uint32_t slot = pn[i0 + (uint32_t)lane];
            const uint32_t w = e_node[slot], dw = e_dp[slot] & 0x7FFFu;
            const uint64_t ent = ((uint64_t)w << 32) | ((uint64_t)dw << 16) | slot;
            uint32_t pos = e_hash(w, dw), g3 = 0;
            while (atomicCAS((unsigned long long*)&ht[pos], (unsigned long long)SEGX_EMPTY64, (unsigned long long)ent) != SEGX_EMPTY64) {
              if (++g3 > 2u * HS) { stuckb = true; break; }
              pos = (pos + 1u) & (HS - 1u);
            }
            e_hpos[slot] = pos;
          }
        }
        ntomb = 0;
        lds_sync();
        if (__ballot(stuckb)) { overflow = true; flags |= G2S_DEV_OVERFLOW_B | G2S_DEV_WATCHDOG; }
      }
    }
    sb += wave_sum(acc_sb);
    xb += wave_sum(acc_xb);
    // (the segment arrays in the scratch are read back below; lines this compute unit read for an earlier gap go)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  if (!have_rs) take_right_set();  // (the search ended before the pruning depth: wave 1 is met here)
  if (overflow && !(flags & G2S_DEV_OVERFLOW_A)) flags |= G2S_DEV_OVERFLOW_B;
  lds_sync();
  const unsigned long long cyc2 = __builtin_amdgcn_s_memtime();
  SEG_PROF_TAIL(0);
  // (two waves per gap — a short list, whose launch lasts as long as its slowest gap: a guessed traceback at the end
  // of this wave — 20-40 k cycles — is worth its while only under the searches still running.  The gaps count themselves
  // as their searches end, here, where the answer has the whole tail to arrive; a search that took longer than most
  // searches take altogether does not ask: it is one of those the launch waits for.)
  uint32_t guess_order = 0xFFFFFFFFu;
  if constexpr (TWO) {
    if (A.tr_spec_text != nullptr && A.tr_guess_until != 0u && !overflow && cyc2 - cyc0 < (unsigned long long)G2S_GUESS_LATE_CYCLES && lane == 0)
      guess_order = (uint32_t)atomicAdd(out_counter + 12, 1ull);
  }

  // ---------------- Q7: an upward and a downward segment of one unitig meeting on a k-mer ------
  // All pairs (every downward segment against every upward one) cost the slowest gaps of a 10 000-gap list a third
  // of their tail — 130-265 k cycles at 230-310 segments, measured with the instrumented build
  // (profiles/r03_segprof_*.txt).  Almost no pair can meet: the two must share k-mers, i.e. overlap as index
  // intervals.  So the upward segments' index intervals are sorted and merged in LDS (the space of s_t, which
  // phase D1 fills later; the large variant: the start of its LDS), a downward segment that touches none of them
  // (the usual case) is done after one binary search, and only the few others are checked against every upward
  // segment.  Lists of a few dozen segments keep the direct all-pairs pass.
  if constexpr (BIG) {
    if (!overflow && !(flags & G2S_DEV_Q7_B) && nseg > 1) {
      bool f7 = false;
      (void)seg_q7_sorted(s_node, s_dl, nseg, (uint64_t*)lds, G2S_SEGX_CAP, lane, &f7);
      if (f7) flags |= G2S_DEV_Q7_B;
    }
  } else if constexpr (TWO) {
    // two waves: wave 1, idle since phase A, makes this check (in its own LDS region: the right set is in wave 0's
    // registers by now) while this wave goes on with phases C, D1 and D2; its verdict is collected where the gap's
    // flags are written for the last time (q7_collect)
    if (lane == 0) { ash[6] = nseg; ash[7] = (!overflow && !(flags & G2S_DEV_Q7_B) && nseg > 1) ? 1u : 0u; }
    __syncthreads();
  }
  bool collected = false, segs_apart = false;  // no two segments share a k-mer
  if constexpr (!BIG && !TWO) {
    if (!overflow && !(flags & G2S_DEV_Q7_B) && nseg > 1) {
      bool f7 = false;
      // (one wave: the second verdict costs this wave what it saves on a few dozen segments; beyond, the sorted pass
      // gives it for the price of sorting the other half of the segments as well)
      seg_cross_check(s_node, s_dl, nseg, (uint64_t*)s_t, CAP / 2u, lane, !skip_confident && nseg > 64u && nseg <= 192u, &f7, &segs_apart);
      if (f7) flags |= G2S_DEV_Q7_B;
    }
  }
  auto q7_collect = [&]() {  // (two waves) wave 1's verdicts: Q7 into the flags, the record and the diagnostics
    if constexpr (TWO) {
      if (collected) return;
      collected = true;
      __syncthreads();
      segs_apart = ash[8] != 0u;
      if (ash[5]) {
        flags |= G2S_DEV_Q7_B;
        if (lane == 0) {
          go->flags |= G2S_DEV_Q7_B;
          if (dbg) dbg[(size_t)x * dbg_words + 3] |= G2S_DEV_Q7_B;
        }
      }
    }
  };

  // ---------------- phase C in closed form (:1107-1159) ------------------------------------------
  const bool found = best != SEG_INF;
  int c_count = 0, n_len = 0, len0 = 0, len1 = 0, reached_j = 0;
  int d_last = D, final_d = D + 1;
  if (found && !overflow) {
    const int dfound = (int)(best >> 6);
    reached_j = (int)(best & 63u);
    const int err = dfound - (gd.g + lmf + rmf);
    const int l1 = gd.g + lmf + reached_j + err, l2 = gd.g + lmf + reached_j - err;
    c_count = (int)min(c1 + c2, (uint32_t)G2S_DEV_MAX_PATHS);
    if (c1 > 0) { len0 = l1; n_len = 1; if (c2 > 0) { len1 = l2; n_len = 2; } }
    else { len0 = l2; n_len = 1; }
    if (!gd.all_paths) {  // -best-only: the DP stops after the level of the find (:1156-1158)
      d_last = dfound;
      final_d = dfound;
      xb = 0; sb = 0;
      for (uint32_t b0 = 0; b0 < nseg; b0 += 64u) {
        const uint32_t b = b0 + (uint32_t)lane;
        const uint32_t dl = b < nseg ? s_dl[b] : 0u;
        const int d0 = (int)(dl & 0xFFFFu), len = (int)(dl >> 16);
        sb += wave_sum(b < nseg ? (uint32_t)max(0, min(len, d_last - d0 + 1)) : 0u);
        xb += wave_sum(b < nseg ? (uint32_t)max(0, min(len, d_last - d0)) : 0u);
      }
    }
  }
  if (dbg) {  // diagnostics (tests): the entries of phase A and the segments of phase B
    uint32_t* o = dbg + (size_t)x * dbg_words;
    if (lane == 0) { o[0] = gi; o[1] = nA; o[2] = nseg; o[3] = flags; o[4] = roundsA; o[5] = gen; o[6] = (uint32_t)c_count; o[7] = best; }
#ifdef G2S_SEG_PROFILE
    if (lane == 0) for (int pi = 0; pi < 4; pi++) o[dbg_words - 4u + pi] = prof_acc[pi];
    if (lane == 0 && !TWO) for (int pi = 0; pi < 4; pi++) o[dbg_words - 14u + pi] = prof_a[pi];
#endif
    if constexpr (!BIG) {  // (BIG: written while the entries were turned into intervals)
#pragma unroll
      for (int s = 0; s < G2S_SEG_ASETS; s++) {
        const uint32_t e = (uint32_t)s * 64u + (uint32_t)lane;
        if (e < nA && 8u + 2u * e + 1u < dbg_words) { o[8u + 2u * e] = an[s]; o[9u + 2u * e] = al[s]; }
      }
    }
    const uint32_t sb0 = 8u + 2u * (BIG ? G2S_SEGX_EA : 64u * G2S_SEG_ASETS);
    for (uint32_t b = (uint32_t)lane; b < nseg; b += 64u)
      if (sb0 + 6u * b + 5u < dbg_words) {
        o[sb0 + 6u * b] = s_node[b]; o[sb0 + 6u * b + 1] = s_dl[b]; o[sb0 + 6u * b + 2] = s_cnt[b];
        o[sb0 + 6u * b + 3] = s_p01[b]; o[sb0 + 6u * b + 4] = s_p23[b]; o[sb0 + 6u * b + 5] = BIG ? s_gen[b] : s_aux[b];
      }
  }
  if (lane == 0) {
    go->flags = flags;
    go->n_right = nvis;
    go->x_right = xa;
    go->n_states = sb;
    go->x_left = xb;
    go->final_d = final_d;
    go->c_count = c_count;
    go->n_len = n_len;
    go->len[0] = len0;
    go->len[1] = len1;
    go->reached_j = reached_j;
    go->n_xl = 0;
    go->top_level = 0;
    go->stat[0] = roundsA; go->stat[1] = nA; go->stat[2] = gen; go->stat[3] = nseg;
    go->stat[4] = cyc_a_kc; go->stat[5] = (uint32_t)((cyc2 - cyc1) >> 8);
  }
  if (overflow || !(c_count > 0 && n_len > 0)) {  // :1169
    q7_collect();
    publish();
    return;
  }

  // ---------------- phase D1: backward closure over the segments ---------------------------------
  SEG_PROF_TAIL(1);
  if constexpr (BIG) {  // generations into LDS: s_aux collects the closure marks there
    for (uint32_t b = (uint32_t)lane; b < nseg; b += 64u) s_aux[b] = s_gen[b];
    lds_sync();
  }
  const bool want_s = !skip_confident;
  const uint32_t sinknode = (want_s && gd.all_paths && rmf >= 1) ? rl(tg, rmf - 1) : G2S_DEV_INVALID;  // Q3/Q4
  const int lo_sink = max(0, lmf + gd.g - gd.e);  // :1196
  const uint32_t reached = rl(tg, (int)uni((uint32_t)reached_j));
  const bool t_is_s = want_s && !gd.all_paths;  // -best-only: the traceback starts are the sinks (:1245-1259)
  uint32_t start_b0 = SEG_NOPAR, start_b1 = SEG_NOPAR, start_t0 = 0, start_t1 = 0;  // segments and positions of the traceback starts
  bool choice = false;  // some entry of the traceback closure has more than one parent
  // What a segment holds by itself — a sink, a traceback start, a left-flank k-mer at its first state — does not
  // depend on its children: all segments at once, in front of the sweep (inside it, each of the sweep's ~80
  // dependent steps carried these comparisons: phase D1 was half of the tail of a config-2 gap).  s_t: ts | source
  // << 15 | tt << 16, rewritten by the sweep.
  for (uint32_t b0 = 0; b0 < nseg; b0 += 64u) {
    const uint32_t b = b0 + (uint32_t)lane;
    const bool hb = b < nseg;
    const uint32_t v0 = hb ? s_node[b] : 0u, dl = hb ? s_dl[b] : 0u;
    const int d0 = (int)(dl & 0xFFFFu);
    const int len = max(0, min((int)(dl >> 16), d_last - d0 + 1));
    int ts = -1, tt = -1, pstart = -1, jstart = 0;
    if (hb && len > 0) {
      const int ps = seg_pos(v0, (uint32_t)len, sinknode);
      if (ps >= 0 && d0 + ps >= lo_sink) ts = ps;
      const int pt = seg_pos(v0, (uint32_t)len, reached);
      if (pt >= 0 && (d0 + pt == len0 || (n_len > 1 && d0 + pt == len1))) {
        tt = pt;
        if (t_is_s) ts = max(ts, pt);
        pstart = pt;
        jstart = d0 + pt == len0 ? 0 : 1;
      }
    }
    const uint32_t ls = (hb && d0 <= lmf) ? l_seed[d0] : G2S_DEV_INVALID;
    const bool source = ls != G2S_DEV_INVALID && (v0 >> 1) == (ls >> 1);  // :1270, k-mer comparison only
    if (hb) s_t[b] = enc15(ts) | (source ? 0x8000u : 0u) | (enc15(tt) << 16);
    for (uint64_t sm = __ballot(pstart >= 0); sm; sm &= sm - 1) {  // (state (reached, len_j) is unique)
      const int l = __builtin_ctzll(sm);
      if (rl((uint32_t)jstart, l) == 0u) { start_b0 = b0 + (uint32_t)l; start_t0 = rl((uint32_t)pstart, l); }
      else { start_b1 = b0 + (uint32_t)l; start_t1 = rl((uint32_t)pstart, l); }
    }
  }
  lds_sync();
  {
    // the sweep, in reverse: a segment in the closure tells its parents — one addition per parent and mark:
    // children on paths to a sink in bits 20..22 of the parent's word (the out-degree of its last state in the
    // subgraph, for the branch rule below), children in the traceback closure in bits 24..26 (a state has at most
    // four successors); a parent with either count above zero is in that closure whole.  A parent has a lower id
    // than its children, so by the time a chunk of 64 segments is reached every child outside it has spoken; a pass
    // over the chunk is final unless a segment that spoke in it has a parent INSIDE the chunk — then another pass.
    // (The sweep used to take the generations — rounds of phase B — one by one: 80 dependent steps on a config-2 gap;
    // a parent is mostly many rounds older than its child, and most chunks are done in one pass.)
    uint32_t hi = nseg;
    while (hi > 0) {
      const uint32_t lo = hi > 64u ? hi - 64u : 0u;
      const uint32_t b = lo + (uint32_t)lane;
      const bool hb = b < hi;
      const uint32_t pre = hb ? s_t[b] : 0x7FFF7FFFu, dl = hb ? s_dl[b] : 0u;
      const uint32_t p01 = hb ? s_p01[b] : 0xFFFFFFFFu, p23 = hb ? s_p23[b] : 0xFFFFFFFFu;
      const int d0 = (int)(dl & 0xFFFFu);
      const int len = max(0, min((int)(dl >> 16), d_last - d0 + 1));
      const uint32_t q0 = p01 & 0xFFFFu, q1 = p01 >> 16, q2 = p23 & 0xFFFFu, q3 = p23 >> 16;
      const bool speaks = hb && d0 > 0 && !(pre & 0x8000u);
      // (a parent's id is below the segment's own: inside the chunk = at or above its first id; SEG_NOPAR is above all)
      const uint64_t inside_m = (ballot_and(q0 >= lo, q0 != SEG_NOPAR) | ballot_and(q1 >= lo, q1 != SEG_NOPAR) |
                                 ballot_and(q2 >= lo, q2 != SEG_NOPAR) | ballot_and(q3 >= lo, q3 != SEG_NOPAR));
      uint32_t sent = 0u;  // marks this segment has passed on: bit 0 to a sink, bit 1 traceback closure
      int ts = -1, tt = -1;
#pragma nounroll
      while (true) {
        const uint32_t aux = hb ? s_aux[b] : 0u;
        ts = dec15(pre); tt = dec15(pre >> 16);
        if (len > 0) {
          if (aux & (7u << 20)) ts = len - 1;
          if (aux & (7u << 24)) tt = len - 1;
        }
        const uint32_t want = speaks ? ((ts >= 0 ? 1u : 0u) | (tt >= 0 ? 2u : 0u)) & ~sent : 0u;
        const uint32_t mk = ((want & 1u) << 20) | ((want & 2u) << 23);
        if (mk) {
          if (q0 != SEG_NOPAR) atomicAdd(&s_aux[q0], mk);
          if (q1 != SEG_NOPAR) atomicAdd(&s_aux[q1], mk);
          if (q2 != SEG_NOPAR) atomicAdd(&s_aux[q2], mk);
          if (q3 != SEG_NOPAR) atomicAdd(&s_aux[q3], mk);
        }
        sent |= want;
        const bool again = (__ballot(mk != 0u) & inside_m) != 0ull;
        lds_sync();
        if (!again) break;
      }
      if (hb) s_t[b] = enc15(ts) | (enc15(tt) << 16);
      if (__ballot(speaks && tt >= 0 && q1 != SEG_NOPAR)) choice = true;
      hi = lo;
    }
  }
  SEG_PROF_TAIL(2);
  // ---------------- phase D2 for closures without a repeated k-mer (:1314-1435) -----------------
  // When every k-mer occurs at one depth of the S closure the subgraph is a DAG whose vertices are
  // the states: nothing to contract, and the branch rule (:1411-1434: walk the vertices in
  // topological order with a running count, -(in-1) before a vertex, +(out-1) after it; safe <=> the
  // count is 1 at the vertex) only moves at a segment's entry, at a sink inside it and at its last
  // state.  Ascending segment id is a topological order (parents were created first), so the count
  // in front of every segment is a prefix sum.  Otherwise (overlapping index intervals, or more
  // segments than the pairwise check is worth) the host analyses the gap (post.cpp).
  bool analysed = false, sink_safe = false;
  uint32_t sub_vertices = 0, sub_edges = 0;
  int count_s = 0;
  if (want_s && nseg <= 192u) {
    uint32_t n_s = 0, edges = 0, src_out = 0, sink_in = 0;
    bool dag = true;
    q7_collect();  // (two waves: the second wave may have found that no two segments share a k-mer at all)
    for (uint32_t b0 = 0; b0 < nseg; b0 += 64u) {  // totals, and no two S intervals may overlap
      const uint32_t b = b0 + (uint32_t)lane;
      const bool hb = b < nseg;
      const uint32_t st = hb ? s_t[b] : 0x7FFF7FFFu;
      const int ts = dec15(st);
      const bool in_s = ts >= 0;
      const uint32_t v0 = hb ? s_node[b] : 0u, dl = hb ? s_dl[b] : 0u;
      const int d0 = (int)(dl & 0xFFFFu);
      const int len = max(0, min((int)(dl >> 16), d_last - d0 + 1));
      const uint32_t ls = (hb && d0 <= lmf) ? l_seed[d0] : G2S_DEV_INVALID;
      const bool source = ls != G2S_DEV_INVALID && (v0 >> 1) == (ls >> 1);
      const uint32_t p01 = hb ? s_p01[b] : 0xFFFFFFFFu, p23 = hb ? s_p23[b] : 0xFFFFFFFFu;
      const uint32_t npar = ((p01 & 0xFFFFu) != SEG_NOPAR) + ((p01 >> 16) != SEG_NOPAR) + ((p23 & 0xFFFFu) != SEG_NOPAR) + ((p23 >> 16) != SEG_NOPAR);
      int sp = -1;
      if (in_s) {
        const int pk = seg_pos(v0, (uint32_t)len, sinknode);
        if (pk >= 0 && d0 + pk >= lo_sink) sp = pk;
        if (t_is_s) { const int pt = seg_pos(v0, (uint32_t)len, reached); if (pt >= 0 && (d0 + pt == len0 || (n_len > 1 && d0 + pt == len1))) sp = pt; }
        if (sp > ts) sp = -1;
      }
      n_s += wave_sum(in_s ? (uint32_t)ts + 1u : 0u);
      edges += wave_sum(in_s ? (uint32_t)ts + (source ? 1u : (d0 > 0 ? npar : 0u)) + (sp >= 0 ? 1u : 0u) : 0u);
      src_out += (uint32_t)__popcll(__ballot(in_s && source));
      sink_in += (uint32_t)__popcll(__ballot(sp >= 0));
      for (uint64_t m = __ballot(sp >= 0); m; m &= m - 1)
        count_s = (int)min((uint32_t)count_s + rl(hb ? s_cnt[b] : 0u, __builtin_ctzll(m)), (uint32_t)G2S_DEV_MAX_PATHS);
      // no two S intervals may overlap: this chunk against itself and the chunks before it, both sides
      // in registers (a pass over LDS per pair of segments was 15 % of the kernel)
      const uint32_t idx = v0 >> 1;
      const uint32_t ilo = (v0 & 1u) ? idx - (uint32_t)max(ts, 0) : idx, ihi = (v0 & 1u) ? idx : idx + (uint32_t)max(ts, 0);
      for (uint32_t a0 = 0; a0 <= b0 && dag && !segs_apart; a0 += 64u) {
        uint32_t alo_ = ilo, ahi_ = ihi;
        bool a_in = in_s;
        if (a0 != b0) {
          const uint32_t a = a0 + (uint32_t)lane;  // (a < nseg: an earlier chunk is full)
          const int tsa = dec15(s_t[a]);
          const uint32_t va = s_node[a], ia = va >> 1;
          a_in = tsa >= 0;
          alo_ = (va & 1u) ? ia - (uint32_t)max(tsa, 0) : ia;
          ahi_ = (va & 1u) ? ia : ia + (uint32_t)max(tsa, 0);
        }
        for (uint64_t am = __ballot(a_in); am && dag; am &= am - 1) {
          const int al = __builtin_ctzll(am);
          const uint32_t lo_a = rl(alo_, al), hi_a = rl(ahi_, al);
          if (ballot_and(in_s, b > a0 + (uint32_t)al, ilo <= hi_a, lo_a <= ihi)) dag = false;
        }
      }
    }
    if (dag) {
      int bc = 1 + (src_out > 1u ? (int)src_out - 1 : 0);  // the source pseudo-vertex comes first
      for (uint32_t b0 = 0; b0 < nseg; b0 += 64u) {
        const uint32_t b = b0 + (uint32_t)lane;
        const bool hb = b < nseg;
        const uint32_t st = hb ? s_t[b] : 0x7FFF7FFFu;
        const int ts = dec15(st);
        const bool in_s = ts >= 0;
        const uint32_t v0 = hb ? s_node[b] : 0u, dl = hb ? s_dl[b] : 0u;
        const int d0 = (int)(dl & 0xFFFFu);
        const int len = max(0, min((int)(dl >> 16), d_last - d0 + 1));
        const uint32_t ls = (hb && d0 <= lmf) ? l_seed[d0] : G2S_DEV_INVALID;
        const bool source = ls != G2S_DEV_INVALID && (v0 >> 1) == (ls >> 1);
        const uint32_t p01 = hb ? s_p01[b] : 0xFFFFFFFFu, p23 = hb ? s_p23[b] : 0xFFFFFFFFu;
        const int npar = ((p01 & 0xFFFFu) != SEG_NOPAR) + ((p01 >> 16) != SEG_NOPAR) + ((p23 & 0xFFFFu) != SEG_NOPAR) + ((p23 >> 16) != SEG_NOPAR);
        int sp = -1;
        if (in_s) {
          const int pk = seg_pos(v0, (uint32_t)len, sinknode);
          if (pk >= 0 && d0 + pk >= lo_sink) sp = pk;
          if (t_is_s) { const int pt = seg_pos(v0, (uint32_t)len, reached); if (pt >= 0 && (d0 + pt == len0 || (n_len > 1 && d0 + pt == len1))) sp = pt; }
          if (sp > ts) sp = -1;
        }
        const int din = source ? 1 : npar;
        const int outs = hb ? (int)((s_aux[b] >> 20) & 7u) : 0;
        const int d_in = (in_s && din > 1) ? -(din - 1) : 0;
        const int d_mid = (in_s && sp >= 0 && sp < ts) ? 1 : 0;  // out-degree 2: the next state and the sink
        const int dout = in_s ? (ts == len - 1 ? outs : 0) + (sp == ts ? 1 : 0) : 0;
        const int d_out = dout > 1 ? dout - 1 : 0;
        const int total = d_in + d_mid + d_out;
        const int incl = (int)wave_scan((uint32_t)total, lane);
        const int at_entry = bc + incl - total + d_in;
        if (in_s) s_t[b] = st | (at_entry == 1 ? 0x8000u : 0u) | ((at_entry + d_mid == 1) ? 0x80000000u : 0u);
        bc += (int)rl((uint32_t)incl, 63);
      }
      if (sink_in >= 1u) { if (sink_in > 1u) bc -= (int)sink_in - 1; sink_safe = bc == 1; }
      analysed = true;
      sub_vertices = n_s + 2u;
      sub_edges = edges;
      lds_sync();
    }
  }
  if (want_s && nseg > 192u) {
    // (closures the host analyses: the all-paths recount alone — the sum of the counts of the sink states,
    // :1189-1226 — so that phase D3 on the device knows whether the gap counts as filled, :369)
    for (uint32_t b0 = 0; b0 < nseg; b0 += 64u) {
      const uint32_t b = b0 + (uint32_t)lane;
      const bool hb = b < nseg;
      const int ts = dec15(hb ? s_t[b] : 0x7FFF7FFFu);
      const uint32_t v0 = hb ? s_node[b] : 0u, dl = hb ? s_dl[b] : 0u;
      const int d0 = (int)(dl & 0xFFFFu);
      const int len = max(0, min((int)(dl >> 16), d_last - d0 + 1));
      int sp = -1;
      if (ts >= 0) {
        const int pk = seg_pos(v0, (uint32_t)len, sinknode);
        if (pk >= 0 && d0 + pk >= lo_sink) sp = pk;
        if (t_is_s) { const int pt = seg_pos(v0, (uint32_t)len, reached); if (pt >= 0 && (d0 + pt == len0 || (n_len > 1 && d0 + pt == len1))) sp = pt; }
        if (sp > ts) sp = -1;
      }
      for (uint64_t m = __ballot(sp >= 0); m; m &= m - 1)
        count_s = (int)min((uint32_t)count_s + rl(hb ? s_cnt[b] : 0u, __builtin_ctzll(m)), (uint32_t)G2S_DEV_MAX_PATHS);
    }
  }
  SEG_PROF_TAIL(3);
  // ---- the closure leaves as SEGMENTS (32 bytes each, SegRec), children before parents = descending
  // segment id; the host expands them into per-state records (post.cpp: seg_expand).  Writing the
  // ~730 16-byte state records of a gap over the link instead made the launch PCIe-bound: 5.8 MB per
  // 500 gaps, 117 MB per 10 000 (measured: config 3's kernel 2.6 ms = the time the link takes).
  // s_aux becomes the segment's index among the emitted ones.
  uint32_t nrec = 0, nsub = 0, nxp = 0;
  for (uint32_t top = nseg; top > 0; top = top > 64u ? top - 64u : 0u) {
    const bool hb = (uint32_t)lane < top;
    const uint32_t b = hb ? top - 1u - (uint32_t)lane : 0u;
    const uint32_t st = hb ? s_t[b] : 0x7FFF7FFFu;
    const int ts = dec15(st), tt = dec15(st >> 16);
    const bool in = max(ts, tt) >= 0;
    const uint64_t m = __ballot(in);
    if (in) s_aux[b] = nrec + (uint32_t)__popcll(m & below(lane));
    nrec += (uint32_t)__popcll(m);
    nsub += wave_sum(in ? (uint32_t)(max(ts, tt) + 1) : 0u);
  }
  lds_sync();
  const uint32_t nres = 2u * nrec;  // in 16-byte units of the output buffer
  unsigned long long hbase = 0;
  if (lane == 0) hbase = atomicAdd(out_counter, (unsigned long long)nres);
  hbase = __shfl(hbase, 0);
  if (hbase + nres > out_cap) {  // the host buffer is full: the gap runs again in the LDS tier
    if (lane == 0) go->flags = flags | G2S_DEV_OVERFLOW_B | G2S_DEV_WHY_LOG;
    q7_collect();
    publish();
    return;
  }
  // (resident mode, lists that are not deep: a closure the host will analyse — more than 192 segments, a k-mer at two
  // depths — goes to pinned host memory as well, the moment its gap ends, as the large variant's do on deep lists
  // (SegArgs.early_*, fill_segw.hip): the thread that waits for the list analyses it under the rest of the launch instead
  // of behind phase D3's hand-over, where two such closures were 25 us of a 500-gap list's step beside a 24 us kernel)
  SegRec* edst = nullptr;
  uint32_t eslot = 0xFFFFFFFFu, eoff = 0xFFFFFFFFu;
  if (want_s && !analysed) nodes_to_host();  // (in front of the early item's release below, and of everything phase D3 hands over)
  if constexpr (!BIG) {
    if (A.early_items != nullptr && want_s && !analysed && nrec > 0u) {
      unsigned long long slot = 0, so = 0;
      if (lane == 0) {
        slot = atomicAdd(&A.early_ctr[0], 1ull);
        if (slot < (unsigned long long)A.early_cap_items) so = atomicAdd(&A.early_ctr[1], (unsigned long long)nrec);
      }
      slot = __shfl(slot, 0); so = __shfl(so, 0);
      if (slot < (unsigned long long)A.early_cap_items) {
        eslot = (uint32_t)slot;
        if (so + nrec <= (unsigned long long)A.early_cap_segs) { eoff = (uint32_t)so; edst = A.early_segs + eoff; }  // (no room: the item says so)
      }
    }
  }
  {
    SegRec* dst = (SegRec*)(sub_out + hbase);
    for (uint32_t b0 = 0; b0 < nseg; b0 += 64u) {
      const uint32_t b = b0 + (uint32_t)lane;
      if (b >= nseg) continue;
      const uint32_t st = s_t[b];
      const int ts = dec15(st), tt = dec15(st >> 16);
      if (max(ts, tt) < 0) continue;
      const uint32_t dl = s_dl[b];
      const int d0 = (int)(dl & 0xFFFFu);
      const uint32_t v0 = s_node[b];
      int split = ts;  // states t <= split carry safe bit a, the others b: a sink inside the S part
      if (analysed && ts >= 0) {
        const int len = max(0, min((int)(dl >> 16), d_last - d0 + 1));
        int sp = -1;
        const int pk = seg_pos(v0, (uint32_t)len, sinknode);
        if (pk >= 0 && d0 + pk >= lo_sink) sp = pk;
        if (t_is_s) { const int pt = seg_pos(v0, (uint32_t)len, reached); if (pt >= 0 && (d0 + pt == len0 || (n_len > 1 && d0 + pt == len1))) sp = pt; }
        if (sp >= 0 && sp < ts) split = sp;
      }
      const uint32_t ls = d0 <= lmf ? l_seed[d0] : G2S_DEV_INVALID;
      const bool source = ls != G2S_DEV_INVALID && (v0 >> 1) == (ls >> 1);  // :1270, k-mer comparison only
      SegRec r;
      r.node = v0;
      r.depth_len = (uint32_t)d0 | ((uint32_t)(max(ts, tt) + 1) << 16);
      r.cnt = s_cnt[b];
      r.ts_tt = st;
      r.par01 = r.par23 = 0xFFFFFFFFu;
      r.flags = source ? G2S_SUB_SOURCE : 0u;
      r.pad = (uint32_t)max(split, 0);
      if (!source && d0 > 0) {  // parents as indices among the emitted segments (they are all in the closure)
        const uint32_t p01 = s_p01[b], p23 = s_p23[b];
        const uint32_t ps[4] = {p01 & 0xFFFFu, p01 >> 16, p23 & 0xFFFFu, p23 >> 16};
        const uint32_t k = (ps[0] != SEG_NOPAR) + (ps[1] != SEG_NOPAR) + (ps[2] != SEG_NOPAR) + (ps[3] != SEG_NOPAR);
        // Several parents: the traceback picks predecessors(v)[rand() % n] (:1476-1513), i.e. in GATB's
        // neighbour order.  predecessors(v)[i] = succ(v^1)[i] ^ 1, so the slot of a parent is where its last
        // node shows up in v's record of the table; sorted here, the host's sequential in-order pass and
        // the tracebacks do not have to look the order up per segment (G2S_SEG_ORDERED).
        uint32_t key[4] = {0u, 1u, 2u, 3u}, id[4];
        bool ordered = true;
        uint4 sr = make_uint4(G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID);
        if (k > 1u) sr = *(const uint4*)(succ + (size_t)(v0 ^ 1u) * 4);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          id[q] = SEG_NOPAR;
          if (ps[q] == SEG_NOPAR) { key[q] = 8u + (uint32_t)q; continue; }
          id[q] = s_aux[ps[q]];
          if (k > 1u) {
            const uint32_t pn = s_node[ps[q]];
            const uint32_t pl = seg_node(pn, (s_dl[ps[q]] >> 16) - 1u) ^ 1u;  // the parent's last node, flipped
            key[q] = sr.x == pl ? 0u : sr.y == pl ? 1u : sr.z == pl ? 2u : sr.w == pl ? 3u : 4u;
            if (key[q] == 4u) ordered = false;  // (cannot happen on a consistent table: the host sorts then)
          }
        }
#define SEG_CSWAP(a, c) do { if (key[a] > key[c]) { const uint32_t tk = key[a], ti = id[a]; key[a] = key[c]; id[a] = id[c]; key[c] = tk; id[c] = ti; } } while (0)
        SEG_CSWAP(0, 1); SEG_CSWAP(2, 3); SEG_CSWAP(0, 2); SEG_CSWAP(1, 3); SEG_CSWAP(1, 2);
#undef SEG_CSWAP
        r.par01 = id[0] | (id[1] << 16);
        r.par23 = id[2] | (id[3] << 16);
        if (ordered) r.flags |= G2S_SEG_ORDERED;
        nxp += k > 1u ? k - 1u : 0u;
      }
      dst[s_aux[b]] = r;
      if (edst) edst[s_aux[b]] = r;
    }
    if (edst) __threadfence_system();  // (this lane's records are in host memory before the item says so)
    nxp = wave_sum(nxp);
  }
  if (lane == 0) {
    // a traceback from start j consumes 1 + (length - stop depth) draws whenever every path back from it
    // stops at one depth (s1 / s2: lowest | highest << 16 stop depth behind the hits of phase C)
    const uint32_t sa = c1 > 0 ? s1 : s2, sb2 = s2;
    go->fixed_draws[0] = ((sa & 0xFFFFu) == (sa >> 16)) ? 1 + len0 - (int)(sa & 0xFFFFu) : -1;
    go->fixed_draws[1] = (n_len > 1 && (sb2 & 0xFFFFu) == (sb2 >> 16)) ? 1 + len1 - (int)(sb2 & 0xFFFFu) : -1;
    go->stop[0] = sa;
    go->stop[1] = sb2;
    go->start_seg = (start_b0 != SEG_NOPAR ? s_aux[start_b0] : 0xFFFFu) | ((start_b1 != SEG_NOPAR ? s_aux[start_b1] : 0xFFFFu) << 16);
    go->start_t = start_t0 | (start_t1 << 16);
    go->sub_vertices = sub_vertices;
    go->sub_edges = sub_edges;
    go->count_s = count_s;
    const bool to_d2 = A.d2_list && want_s && !analysed;  // (d2_device.hip takes it)
    go->dflags = (analysed || !want_s ? G2S_DEVA_ANALYSED : 0u) | (choice ? G2S_DEVA_CHOICE : 0u) | (sink_safe ? G2S_DEVA_SINK_SAFE : 0u) |
                 (to_d2 ? G2S_DEVA_D2_PENDING : 0u);
    go->flags = flags | G2S_DEV_COMPACT;
    go->n_sub = nsub;
    go->n_xp = nxp;
    go->n_xl = nrec;
    go->sub_off = hbase;
    go->x_sub = nsub;
    go->stat[6] = gen;
    go->stat[7] = (uint32_t)((__builtin_amdgcn_s_memtime() - cyc2) >> 8);
    if (to_d2) {  // (record and closure are in device memory in front of the entry: g2s_d2_small may be polling the list)
      if (A.d2_tag) __threadfence();  // (a polling launch reads them while this kernel runs; a launch behind it needs no fence)
      const unsigned long long at = atomicAdd(out_counter + 4, 1ull);
      __hip_atomic_store(&A.d2_list[at], gi | A.d2_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (A.d2_ticks) A.d2_list[A.d2_ticks + gi] = (uint32_t)__builtin_amdgcn_s_memrealtime();  // (tools: when the closure was listed)
    }
  }
#ifdef G2S_SEG_PROFILE
  SEG_PROF_TAIL(4);
  if (dbg && lane == 0) {
    uint32_t* o = dbg + (size_t)x * dbg_words;
    for (int pi = 0; pi < 4; pi++) o[dbg_words - 8u + pi] = (uint32_t)(prof_tail[pi + 1] - prof_tail[pi]);
    o[dbg_words - 10u] = prof_wait;
    o[dbg_words - 9u] = prof_load;
  }
#endif
  q7_collect();
  if constexpr (!BIG) {
    if (eslot != 0xFFFFFFFFu && lane == 0) {  // the early item: the gap's finished record, then what says it is complete
      if (edst) {
        static_assert(sizeof(GapOut) % 16 == 0, "GapOut is copied in 16-byte words");
        __threadfence();  // (the record's words above: read back below)
        const uint4* gs = (const uint4*)go;
        uint4* gd4 = (uint4*)&A.early_outs[eslot];
        for (uint32_t q = 0; q < sizeof(GapOut) / 16u; q++) gd4[q] = gs[q];
      }
      uint32_t* it = A.early_items + 8u * (size_t)eslot;
      it[0] = gi; it[1] = edst ? nrec : 0u; it[2] = eoff; it[3] = 0u;
      __threadfence_system();
      __hip_atomic_store(&it[4], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // ---------------- the traceback of a gap whose path is its only one, here (round 6) -----------------------------
  // :1437-1522 draws rand() for the path length and at every traced base, but what a traceback WRITES depends on the
  // values only where it has a choice: one path length and no entry of the traceback closure with several parents
  // (most gaps of a list) leave one path, whatever is drawn — its fill text, case, fuz values and number of draws are
  // known now.  Resident mode's trace kernel (d3_device.hip) waits for every gap of the list — the offsets into the one
  // rand() stream are prefix sums — and then pushes the whole list's text through the link at once: 214 us for the
  // 8 MB of a 10 000-gap list, behind everything else.  Such a gap's wave writes text and result record itself, while
  // the other gaps are still being searched (the link is idle then); phase D3 still counts the gap's draws, its wave
  // of the trace kernel finds G2S_DEVA_TRACED and leaves.  Everything here follows g2s_d3_trace step by step (the walk
  // along the parents, the safe bit of every base, lower case = not safe and k or more below the nearest safe base
  // above, :1466-1468); anything out of the ordinary leaves the gap to that kernel.
  if constexpr (!BIG) {
    const bool no_over = sb <= A.tr_max_states && nvis <= A.tr_max_states;  // (else the -max-mem verdict: phase D3's)
    const uint32_t sa_w = c1 > 0 ? s1 : s2;
    // (round 6, second step: a traceback that HAS choices is written here too — as a GUESS: the first path length, the
    // first parent at every choice — with a copy of text and record in device memory.  The bench's genome has a second
    // haplotype every ~500 bp: nine tracebacks in ten cross a bubble, but the two ways through one differ in a base or
    // two.  Phase D3's trace kernel still traces such a gap for real, compares 64 bases at a time with what was guessed,
    // and sends through the link only what differs: G2S_DEVA_SPEC.)
    const bool sure = !choice && n_len == 1 && (sa_w & 0xFFFFu) == (sa_w >> 16);
    bool may_guess = A.tr_spec_text != nullptr;
    if constexpr (TWO) {
      if (may_guess && A.tr_guess_until != 0u) may_guess = uni(guess_order) < A.tr_guess_until;  // (asked at the head of the tail)
    } else {
      // (a chip-filling list: its launch ends with its slowest gaps too — a few dozen of 10 000, whose searches took several
      // times the mean: their guesses, 20-40 k cycles each, would be the launch's last; the trace kernel has them.
      // Config 3: g2s_fill_seg 0.317 -> 0.313 ms)
      if (may_guess && cyc2 - cyc0 > (unsigned long long)G2S_GUESS_LATE_CYCLES_LONG) may_guess = false;
    }
    if (A.tr_results != nullptr && (analysed || !want_s) && (sure || may_guess) && start_b0 != SEG_NOPAR && no_over &&
        gd.rlog_cap == 0u /* no skip rule on this gap */ && len0 >= 1 && len0 <= 4096 && nrec <= 256u) {
      const int len = len0, k = A.tr_k;
      uint2* hop = (uint2*)s_cnt;           // by hop: the depth at which it is entered | the segment | entry state << 16
      unsigned char* cb = (unsigned char*)s_p23;  // by fill-buffer index: safe bit, then the character (s_p23 and s_aux: 4 096 bytes)
      lds_sync();
      // (i) the chain of segments: from the start along the (first) parent to a source.  As in g2s_d3_trace: what the
      // walk needs of a segment — the parent it goes on to, its emitted id (for the chain's hash), whether it is a source
      // or has no way on — is packed into ONE word per segment by all lanes first, so that the walk itself, a chain of
      // dependent steps on one wave, reads one LDS word a segment (with five reads and the tests a step it was 400
      // cycles a segment, 10 k cycles of a guess); then all lanes check that the hops' depths follow each other.
      uint32_t* pk2 = s_p23;  // parent | emitted id << 16 | source << 30 | no way on << 31   (s_p23: free until the characters go there)
      for (uint32_t b0 = 0; b0 < nseg; b0 += 64u) {
        const uint32_t b = b0 + (uint32_t)lane;
        if (b >= nseg) continue;
        const uint32_t dl = s_dl[b], v0 = s_node[b], par = s_p01[b] & 0xFFFFu;
        const int d0 = (int)(dl & 0xFFFFu);
        const uint32_t ls = d0 <= lmf ? l_seed[d0] : G2S_DEV_INVALID;
        const bool source = ls != G2S_DEV_INVALID && (v0 >> 1) == (ls >> 1);  // :1455-1462
        const bool stuck = !source && (d0 < 1 || par == SEG_NOPAR || par >= nseg);  // (:1493-1510: the trace kernel's, and the host's)
        pk2[b] = (par & 0xFFFFu) | ((s_aux[b] & 0x3FFu) << 16) | (source ? 0x40000000u : 0u) | (stuck ? 0x80000000u : 0u);
      }
      lds_sync();
      bool bad = false;
      int nh = 0, d_end = -1;
      unsigned long long chain_hash = 14695981039346656037ull;  // (of the emitted ids of the segments entered, in order: the trace kernel's walk makes the same)
      {
        uint32_t si = start_b0, w = 0u;
        for (;;) {
          if (nh >= 256) { bad = true; break; }
          w = uni(pk2[si]);
          chain_hash = (chain_hash ^ (unsigned long long)((w >> 16) & 0x3FFu)) * 1099511628211ull;
          if (lane == 0) hop[nh].y = si;
          nh++;
          if (w & 0xC0000000u) break;
          si = w & 0xFFFFu;
        }
        if (!bad && !(w & 0x40000000u)) bad = true;
      }
      lds_sync();
      // (ii) all lanes, a hop each: the walk enters the first segment at the start's state and every other at its last
      // closure state, and steps from a segment's first state to the depth below: the depth at which hop h is entered is
      // len less the states passed before it — a scan — and has to be the segment's own depth at that state
      {
        int carry = 0;
        for (int h0 = 0; !bad && h0 < nh; h0 += 64) {
          const int h = h0 + lane;
          const bool in = h < nh;
          const uint32_t sq = in ? hop[h].y : 0u;
          const uint32_t dl = in ? s_dl[sq] : 0u, st = in ? s_t[sq] : 0x7FFF7FFFu;
          const int d0 = (int)(dl & 0xFFFFu);
          const int t = h == 0 ? (int)start_t0 : max(dec15(st), dec15(st >> 16));  // (a child in the closure puts the whole parent there)
          const int inc = in ? t + 1 : 0;
          const int incl = (int)wave_scan((uint32_t)inc, lane);
          const int at = len - carry - (incl - inc);
          if (__ballot(in && (t < 0 || d0 + t != at)) != 0ull) { bad = true; break; }
          if (in) hop[h] = make_uint2((uint32_t)at, sq | ((uint32_t)t << 16));
          if (__ballot(in && h + 1 == nh)) d_end = (int)rl((uint32_t)d0, (nh - 1) & 63);
          carry += (int)rl((uint32_t)incl, 63);
        }
      }
      lds_sync();
      if (!bad && sure && d_end != (int)(sa_w & 0xFFFFu)) bad = true;  // (the stop depth the search itself found)
      lds_sync();
      if (!bad) {
        const int stop0 = d_end, left_fuz = lmf - d_end, draws = 1 + len - d_end;
        const int npos = len - stop0, per = (npos + 63) / 64;
        const int hi_d = len - lane * per, cnt = max(0, min(per, hi_d - stop0));  // this lane: depths (hi_d - cnt, hi_d]
        // the safe bit of the k-mer of state q of segment b (g2s_d3_trace: outside_safe, and the rule in front of it)
        auto split_of = [&](uint32_t b, int ts) -> int {
          int split = ts;
          if (analysed && ts >= 0) {
            const uint32_t dl = s_dl[b], v0 = s_node[b];
            const int d0 = (int)(dl & 0xFFFFu);
            const int ln = max(0, min((int)(dl >> 16), d_last - d0 + 1));
            int sp = -1;
            const int pk = seg_pos(v0, (uint32_t)ln, sinknode);
            if (pk >= 0 && d0 + pk >= lo_sink) sp = pk;
            if (t_is_s) { const int pt = seg_pos(v0, (uint32_t)ln, reached); if (pt >= 0 && (d0 + pt == len0 || (n_len > 1 && d0 + pt == len1))) sp = pt; }
            if (sp >= 0 && sp < ts) split = sp;
          }
          return max(split, 0);
        };
        auto safe_of = [&](uint32_t b, int q) -> bool {
          if (!want_s) return true;
          const uint32_t st = s_t[b];
          const int ts = dec15(st);
          if (q > ts) {  // in the traceback closure only: the first segment in emission order that holds the k-mer on a path to a sink
            const uint32_t v0 = s_node[b];
            const uint32_t x = (v0 & 1u) ? (v0 >> 1) - (uint32_t)q : (v0 >> 1) + (uint32_t)q;
            for (int o = (int)nseg - 1; o >= 0; o--) {
              const uint32_t ost = s_t[o];
              if (max(dec15(ost), dec15(ost >> 16)) < 0) continue;  // (not emitted)
              const int ots = dec15(ost);
              const uint32_t on = s_node[o], oidx = on >> 1;
              const int tq = (on & 1u) ? (int)oidx - (int)x : (int)x - (int)oidx;
              if (tq >= 0 && tq <= ots) return tq <= split_of((uint32_t)o, ots) ? (ost & 0x8000u) != 0u : (ost & 0x80000000u) != 0u;
            }
            return sink_safe;
          }
          return q > split_of(b, ts) ? (st & 0x80000000u) != 0u : (st & 0x8000u) != 0u;
        };
        auto hop_find = [&](int d) -> int {  // the last hop entered at or above depth d
          int lo = 0, hi = nh;
          while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)hop[mid].x >= d) lo = mid; else hi = mid; }
          return lo;
        };
        // pass 1: the safe bits of this lane's stretch, from the top of the fill downwards.  What a segment says about
        // its states — where its part on a path to a sink ends, where the bit changes inside it, the two bits — is
        // worked out when the stretch enters the segment, not per base (with it per base this pass was most of the
        // 19 k cycles a guess cost a gap's wave)
        int lowest_safe = 0x7FFFFFFF;
        if (cnt > 0) {
          int h = hop_find(hi_d);
          uint2 hr = hop[h];
          int d0 = (int)hr.x - (int)(hr.y >> 16);
          int h_ts = -1, h_split = 0;
          bool h_a = true, h_b = true;
          auto enter = [&]() {
            if (!want_s) return;
            const uint32_t b = hr.y & 0xFFFFu, st = s_t[b];
            h_ts = dec15(st);
            h_split = split_of(b, h_ts);
            h_a = (st & 0x8000u) != 0u; h_b = (st & 0x80000000u) != 0u;
          };
          enter();
          for (int c = 0; c < cnt; c++) {
            const int p = hi_d - c;
            if (p < d0) { h++; hr = hop[h]; d0 = (int)hr.x - (int)(hr.y >> 16); enter(); }
            const int q = p - d0;
            const bool sf = !want_s ? true : q > h_ts ? safe_of(hr.y & 0xFFFFu, q) : (q > h_split ? h_b : h_a);
            if (sf) lowest_safe = p;
            cb[p - 1] = sf ? 1u : 0u;
          }
        }
        // the nearest safe depth above each lane's stretch (the walk begins with the top of the fill counting as safe)
        int above = lowest_safe;
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(above, o); if (lane >= o) above = min(above, y); }
        above = __shfl_up(above, 1);
        if (lane == 0) above = 0x7FFFFFFF;
        int last_solid = min(above, len);
        // pass 2: case, and the characters (the last base of the k-mer, by orientation) — sixteen loads in flight: a lane's
        // stretch of a 1 000-base fill in one round trip (with eight, the second round trip was a fifth of a guess's cost)
        if (cnt > 0) {
          int h = hop_find(hi_d);
          uint2 hr = hop[h];
          int d0 = (int)hr.x - (int)(hr.y >> 16);
          uint32_t v0 = s_node[hr.y & 0xFFFFu];
          for (int c0 = 0; c0 < cnt; c0 += 16) {
            uint32_t xs[16];
            bool lowc[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
              const int c = c0 + u;
              xs[u] = 0u; lowc[u] = false;
              if (c < cnt) {
                const int p = hi_d - c;
                if (p < d0) { h++; hr = hop[h]; d0 = (int)hr.x - (int)(hr.y >> 16); v0 = s_node[hr.y & 0xFFFFu]; }
                const int q = p - d0;
                const bool up = (v0 & 1u) == 0u;
                xs[u] = (up ? (v0 >> 1) + (uint32_t)q : (v0 >> 1) - (uint32_t)q) | (up ? 0u : 0x80000000u);
                if (cb[p - 1]) last_solid = p;
                else lowc[u] = p <= last_solid - k;
              }
            }
            char ch[16];
#pragma unroll
            for (int u = 0; u < 16; u++) ch[u] = (c0 + u < cnt) ? ((xs[u] >> 31) ? A.tr_chd[xs[u] & 0x7FFFFFFFu] : A.tr_chu[xs[u]]) : (char)0;
#pragma unroll
            for (int u = 0; u < 16; u++)
              if (c0 + u < cnt) cb[hi_d - (c0 + u) - 1] = (unsigned char)(lowc[u] ? (ch[u] | 0x20) : ch[u]);
          }
        }
        lds_sync();
        // the text, 64 consecutive bytes an instruction, and the record, a word a lane (g2s_d3_trace: finish)
        const uint64_t abs_off = A.tr_arena_base + gd.rlog_off;
        char* buf = A.tr_arena + abs_off;
        for (int p = stop0 + lane; p < len; p += 64) buf[p] = (char)cb[p];
        if (lane == 0) buf[len] = '\0';
        if (!sure) {  // (the guess, where the trace kernel can compare with it)
          char* sbuf = A.tr_spec_text + abs_off;
          for (int p = stop0 + lane; p < len; p += 64) sbuf[p] = (char)cb[p];
        }
        const uint64_t fo = abs_off + (uint64_t)stop0;
        const uint32_t rflags = ((flags & (G2S_DEV_Q7_A | G2S_DEV_Q7_B | G2S_DEV_Q7_D)) ? G2S_GAP_Q7 : 0u) | G2S_GAP_PHASE_D;
        const uint32_t cnt_out = (uint32_t)((want_s && gd.all_paths) ? count_s : c_count);
        uint32_t w = 0u;
        switch (lane) {
          case 0: w = cnt_out; break;
          case 1: w = (uint32_t)left_fuz; break;
          case 2: w = (uint32_t)reached_j; break;
          case 3: w = rflags; break;
          case 4: w = (uint32_t)fo; break;
          case 5: w = (uint32_t)(fo >> 32); break;
          case 6: w = (uint32_t)(len - stop0); break;
          case 7: w = (uint32_t)draws; break;
          case 8: case 16: w = want_s ? sub_vertices : 0u; break;
          case 10: case 18: w = want_s ? sub_edges : 0u; break;
          case 20: w = (uint32_t)c_count; break;
          case 21: w = (uint32_t)n_len; break;
          case 22: w = (uint32_t)len0; break;
          case 23: w = (uint32_t)len1; break;
          default: break;
        }
        if (lane < 28) A.tr_results[(size_t)gi * 28u + (uint32_t)lane] = w;
        if (!sure && lane < 28) A.tr_spec_res[(size_t)gi * 28u + (uint32_t)lane] = w;
        if (lane == 0) {
          go->top_level = (uint32_t)stop0 | ((uint32_t)len << 16);  // (where the text written here begins and ends)
          go->dflags |= sure ? G2S_DEVA_TRACED : G2S_DEVA_SPEC;
          // (a guess: which chain of segments it followed — the trace kernel's wave that finds its own chain to be the same
          // one has nothing to compare or send; the two diagnostic words of the record carry it)
          if (!sure) { go->stat[6] = (uint32_t)chain_hash; go->stat[7] = (uint32_t)(chain_hash >> 32); }
        }
      }
    }
  }
  publish();
}

// dynamic LDS: 7 arrays of G2S_SEG_CAP words + left seeds
__global__ __launch_bounds__(64) void g2s_fill_seg(const SegArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  seg_fill_one<false, false>(lds, A, blockIdx.x, nullptr);
}

// Two waves per gap (see seg_fill_one): for lists short enough to be latency-bound — the launch ends with its
// slowest gap, and that gap's phase A (a quarter of its cycles) runs beside the first half of its phase B.
// dynamic LDS: the segment arrays and seeds, then phase A's table, queues and hand-over words
__global__ __launch_bounds__(128) void g2s_fill_seg2(const SegArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  seg_fill_one<false, true>(lds, A, blockIdx.x, nullptr);
}

// The large variant: one workgroup per compute unit (it takes nearly all of the LDS), each working
// through the list by an atomic counter so that the longest searches (the list is sorted) start first.
__global__ __launch_bounds__(64) void g2s_fill_segx(const SegArgs A, uint32_t* scratch, uint32_t ngaps,
                                                     unsigned long long* next_gap) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint32_t* scr = scratch + (size_t)blockIdx.x * SEGX_SCR_WORDS;
  while (true) {
    unsigned long long x = 0;
    if (threadIdx.x == 0) x = atomicAdd(next_gap, 1ull);
    x = __shfl(x, 0);
    if (x >= (unsigned long long)ngaps) break;
    seg_fill_one<true, false>(lds, A, (uint32_t)x, scr);
    lds_sync();
    __threadfence_block();
  }
}

namespace g2s {

uint32_t d2_ticks_offset = 0u;


size_t fill_seg_lds_bytes() { return 4u * (7u * G2S_SEG_CAP + 32u); }
size_t fill_seg2_lds_bytes() {  // (... + the next round's records: 2 x 64 x sets x (16 + 4) bytes)
  return 4u * (7u * G2S_SEG_CAP + 32u + 2u * 128u * G2S_SEG_ASETS + 128u * G2S_SEG_ASETS + 4u * 64u * G2S_SEG_ASETS + 16u + 2u * 64u * G2S_SEG_ASETS * 5u);
}
uint32_t fill_seg_dbg_words() { return 8u + 2u * 64u * G2S_SEG_ASETS + 6u * G2S_SEG_CAP + 14u; }  // (+14: profile words)
size_t fill_segx_lds_bytes() { return 4u * SEGX_LDS_WORDS; }
size_t fill_segx_scratch_bytes(uint32_t workgroups) { return (size_t)workgroups * SEGX_SCR_WORDS * 4u; }
uint32_t fill_segx_dbg_words() { return 8u + 2u * G2S_SEGX_EA + 6u * G2S_SEGX_CAP + 14u; }

hipError_t launch_fill_seg(hipStream_t st, uint32_t ngaps, const uint32_t* succ, const uint32_t* urec, const GapDev* gaps,
                           const uint32_t* gap_ids, const uint32_t* flank_nodes, SubRec* sub_out,
                           unsigned long long out_cap, unsigned long long* out_counter, GapOut* outs, GapOut* outs_host,
                           uint32_t* done_list, int skip_confident, uint32_t* dbg, bool two_waves, unsigned long long* xcd_tickets,
                           uint32_t* xcd_list, uint32_t xcd_stride, uint32_t pub_batch, bool resident, uint32_t* ovf_list,
                           uint32_t* d2_list, uint32_t d2_tag, const SegInline* inl, const SegEarly* early, const SegTrace* tr,
                           const GapLite* lite, int lite_e, int lite_all_paths, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (ngaps == 0) return hipSuccess;
  size_t bytes = two_waves ? fill_seg2_lds_bytes() : fill_seg_lds_bytes();
  // (G2S_SEG_LDS_PAD=BYTES, measurements only: a larger LDS request per gap = fewer gaps resident per compute unit)
  static const size_t lds_pad = getenv("G2S_SEG_LDS_PAD") ? (size_t)atoi(getenv("G2S_SEG_LDS_PAD")) : 0;
  bytes += lds_pad;
  {  // (the attribute is per device and kernel: set when a device sees a kernel for the first time, not per launch)
    static std::mutex mu;
    static std::set<std::pair<int, int>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (done.insert(std::make_pair(dev, two_waves ? 1 : 0)).second) {
      const hipError_t e = hipFuncSetAttribute(two_waves ? (const void*)g2s_fill_seg2 : (const void*)g2s_fill_seg,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e != hipSuccess) { done.erase(std::make_pair(dev, two_waves ? 1 : 0)); return e; }
    }
  }
  if (!xcd_tickets || !xcd_list || pub_batch < 2u || pub_batch > 64u || (pub_batch & (pub_batch - 1u))) pub_batch = 1u;
  const SegArgs A = {succ, urec, GapSrc{gaps, resident ? lite : nullptr, lite_e, lite_all_paths}, gap_ids, flank_nodes, sub_out, out_cap, out_counter, outs, outs_host, done_list,
                     skip_confident, dbg, fill_seg_dbg_words(), xcd_tickets, xcd_list, xcd_stride, pub_batch, resident ? 1u : 0u,
                     (resident && d2_list) ? g2s::d2_ticks_offset : 0u, resident ? ovf_list : nullptr,
                     (resident && early) ? early->segs : nullptr, (resident && early) ? early->items : nullptr,
                     (resident && early) ? early->outs : nullptr, (resident && early) ? early->ctr : nullptr,
                     (resident && early) ? early->cap_items : 0u, (resident && early) ? early->cap_segs : 0u,
                     resident ? d2_list : nullptr, resident ? d2_tag : 0u,
                     (resident && inl) ? inl->text : nullptr, inl ? inl->nodes_dev : nullptr, inl ? inl->nodes_host : nullptr,
                     inl ? inl->text_stride : 0u, inl ? inl->lk : FlankLookup(),
                     (resident && tr) ? tr->results : nullptr, tr ? tr->arena : nullptr, tr ? tr->arena_base : 0ull, tr ? tr->chu : nullptr,
                     tr ? tr->chd : nullptr, tr ? tr->max_states : 0ull, tr ? tr->k : 0, tr ? tr->spec_text : nullptr, tr ? tr->spec_res : nullptr,
                     (tr && two_waves) ? tr->guess_until : 0u};
  if (ev_start != nullptr || ev_stop != nullptr) {  // (a bracketed launch: the dispatch's own times in the two events)
    if (two_waves) hipExtLaunchKernelGGL(g2s_fill_seg2, dim3(ngaps), dim3(128), (uint32_t)bytes, st, ev_start, ev_stop, 0u, A);
    else hipExtLaunchKernelGGL(g2s_fill_seg, dim3(ngaps), dim3(64), (uint32_t)bytes, st, ev_start, ev_stop, 0u, A);
    return hipGetLastError();
  }
  if (two_waves) hipLaunchKernelGGL(g2s_fill_seg2, dim3(ngaps), dim3(128), bytes, st, A);
  else hipLaunchKernelGGL(g2s_fill_seg, dim3(ngaps), dim3(64), bytes, st, A);
  return hipGetLastError();
}

hipError_t launch_fill_segx(hipStream_t st, uint32_t ngaps, uint32_t workgroups, const uint32_t* succ, const uint32_t* urec,
                            const GapDev* gaps, const uint32_t* gap_ids, const uint32_t* flank_nodes, SubRec* sub_out,
                            unsigned long long out_cap, unsigned long long* out_counter, GapOut* outs, GapOut* outs_host,
                            uint32_t* done_list, int skip_confident, uint32_t* dbg, uint32_t* scratch,
                            unsigned long long* next_gap) {
  if (ngaps == 0) return hipSuccess;
  const size_t bytes = fill_segx_lds_bytes();
  hipError_t e = hipFuncSetAttribute((const void*)g2s_fill_segx, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  const SegArgs A = {succ, urec, GapSrc{gaps, nullptr, 0, 0}, gap_ids, flank_nodes, sub_out, out_cap, out_counter, outs, outs_host, done_list,
                     skip_confident, dbg, fill_segx_dbg_words(), nullptr, nullptr, 0u, 1u, 0u, 0u, nullptr};
  hipLaunchKernelGGL(g2s_fill_segx, dim3(workgroups), dim3(64), bytes, st, A, scratch, ngaps, next_gap);
  return hipGetLastError();
}

}  // namespace g2s
