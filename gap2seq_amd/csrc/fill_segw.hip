// gap2seq_amd/csrc/fill_segw.hip — the LARGE VARIANT of the segment tier on a WORKGROUP OF EIGHT WAVES per gap
// (g2s_fill_segw): phases A, B, C, D1 and (small closures) D2 of /root/reference/src/Gap2Seq.cpp:858-1435 for the
// gaps that outgrow the capacities of g2s_fill_seg (-dist-error 2000: thousands of segments, hundreds of pending
// events, thousands of right-set entries).  Same algorithm as fill_seg.hip (read its header first; tests/seg_model.py
// is the executable restatement), other containers, and every pass over them spread over the 512 threads of the
// workgroup that owns the compute unit's LDS.
//
// Why: round 3's large variant (g2s_fill_segx) ran ONE wave per compute unit — 256 waves on a chip with 8 192 wave
// slots — and its slowest gap (877 rounds of phase B at 17 k cycles, then 3.6 M cycles of tail) was the launch:
// 8.1 ms for 422 gaps of BASELINE config 5.  A round there was a chain of wave-wide passes: two passes over a list
// of up to 1 024 pending events, a chunk of selected events, four insertion passes per chunk.  Here
//   * a pending event lives in an LDS SLOT owned (for the scans) by one thread: new events fetch their exit
//     record, the horizon is a minimum over the slots, final events are selected where they sit — no lists;
//   * a selected event is worked on by ITS thread (length under the pruning rule: a binary search over the sorted
//     right-set intervals in LDS), its children are dealt to the lanes of its wave (lane = child) for the pruning
//     test and the insertion: claim by compare-and-swap in an open-addressing table, counts merged with atomic
//     adds, parents through a slot counter, stop depths with atomic min/max — the protocol of round 3's variant,
//     which is safe across waves because the LDS executes every wave's operations in order;
//   * a round has TWO workgroup barriers: behind the horizon, and behind the insertions;
//   * free slots are a ring: pops take what was free when the round began, pushes append — no ordering between
//     the two inside a round; dead table entries are left in place (no later key can equal them: all carry depths
//     at or above the horizon) and swept when a quarter of the table is dead;
//   * phase A's rounds, the packing/sorting of the right set, phase C's hits, the Q7 check between segments, the
//     per-segment facts of phase D1 and the emission run on all eight waves; phase D1's sweep (a chain of
//     generations) and phase D2 of small closures on wave 0.
// Segments within one generation are numbered in the order the waves reserve them: nothing downstream depends on
// that order (any order is topological; parents leave in GATB's order; tools/seg_check.py compares sets).
// Integer work only: no MFMA.
#include "sync_debug.h"
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "fill_device.h"
#include "fill_seg.h"
#include "seg_device.h"

#define SEGW_NW 8u
#define SEGW_NT (64u * SEGW_NW)
#define SEGW_LDS_WORDS 40960u  // all 160 KB of the compute unit
// global scratch of one workgroup (words): seven segment arrays, two queues of phase A (also: the packed entries)
#define SEGW_SCR_WORDS (7u * G2S_SEGX_CAP + 2u * G2S_SEGX_QCAP)
#define SEGW_CL_CAP 4096u      // downward segments that touch an upward one (Q7 between segments)
// -DG2S_SEGW_PROFILE: cycles of the sections of the kernel as wave 0 sees them, summed per gap into the last words of
// the gap's diagnostics row (G2S_SEG_DUMP; tools/segw_profile.py)
#ifdef G2S_SEGW_PROFILE
#define WPROF(i) pw_t[i] = __builtin_amdgcn_s_memtime()
#define WPROF_ADD(acc, a, b) acc += (uint32_t)(pw_t[b] - pw_t[a])
#else
#define WPROF(i) do {} while (0)
#define WPROF_ADD(acc, a, b) do {} while (0)
#endif
// -DG2S_SEGW_PROFILE=2: eight sections of the lane = (event, successor slot) pass instead of phase A's and the tail's words
#if defined(G2S_SEGW_PROFILE) && G2S_SEGW_PROFILE == 2
#define WFINE(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pf_acc[i] += (uint32_t)(t_ - pf_last); pf_last = t_; } while (0)
#define WFINE_START() pf_last = __builtin_amdgcn_s_memtime()
#else
#define WFINE(i) do {} while (0)
#define WFINE_START() do {} while (0)
#endif

namespace {

// shared words of the workgroup (LDS)
enum {
  SH_NA = 0, SH_NQ0, SH_NQ1, SH_FLAGS, SH_OVF, SH_H0, SH_H1, SH_NSEG, SH_QHEAD, SH_QTAIL, SH_NPEND, SH_NDEAD,  // 0-11
  SH_SB, SH_XB, SH_BEST, SH_C1, SH_C2, SH_S1, SH_S2, SH_NVIS, SH_XA, SH_M, SH_X, SH_CHAIN,                      // 12-23
  SH_CS_LO, SH_CS_HI,                                                                                           // 24-25 (one 64-bit sum)
  SH_PK, SH_BO_SB, SH_BO_XB, SH_NU, SH_ANYDN, SH_MU, SH_NC, SH_D2F, SH_D2V, SH_D2E,                             // 26-35
  SH_START0, SH_START1, SH_ST0, SH_ST1, SH_CHOICE, SH_NREC, SH_NSUB, SH_NXP, SH_HBASE_LO, SH_HBASE_HI,          // 36-45
  SH_NSEL, SH_NSEL1,                                                                                            // 46-47
  SH_ESLOT, SH_EOFF,                                                                                            // 48-49 (the gap's item of the early hand-over, its segments' offset)
  SH_WORDS = 64
};
static_assert(SH_CS_LO % 2 == 0, "the 64-bit sum is 8-byte aligned");

// position in a shared list for every lane with p: one atomic per wave
__device__ __forceinline__ uint32_t wave_reserve(bool p, uint32_t* ctr, int lane) {
  const uint64_t m = __ballot(p);
  if (!m) return 0u;
  const int first = __builtin_ctzll(m);
  uint32_t base = 0;
  if (lane == first) base = atomicAdd(ctr, (uint32_t)__popcll(m));
  base = rl(base, first);
  return base + (uint32_t)__popcll(m & below(lane));
}

// ascending bitonic sort of n2 (a power of two >= 2) 64-bit keys in LDS by the whole workgroup
__device__ __forceinline__ void block_sort64(uint64_t* a, uint32_t n2, uint32_t tid) {
  for (uint32_t k = 2; k <= n2; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = tid; t < (n2 >> 1); t += SEGW_NT) {
        const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));
        const uint32_t l = i | j;
        const bool up = (i & k) == 0u;
        const uint64_t p = a[i], q = a[l];
        if ((p > q) == up) { a[i] = q; a[l] = p; }
      }
      __syncthreads();
    }
}

}  // namespace

// One gap, one workgroup of SEGW_NT threads.  lds: SEGW_LDS_WORDS words; scr: SEGW_SCR_WORDS words of global scratch.
__device__ __forceinline__ void segw_fill_one(uint32_t* lds, const SegArgs& A, const uint32_t x, uint32_t* scr) {
  constexpr uint32_t NT = SEGW_NT, CAP = G2S_SEGX_CAP, EA = G2S_SEGX_EA, AS = G2S_SEGX_AS, QCAP = G2S_SEGX_QCAP;
  constexpr uint32_t PE = G2S_SEGX_PE, HS = G2S_SEGX_HS;
  const uint32_t* __restrict__ succ = A.succ;
  const uint32_t* __restrict__ urec = A.urec;
  SubRec* sub_out = A.sub_out;
  const unsigned long long out_cap = A.out_cap;
  unsigned long long* out_counter = A.out_counter;
  const int skip_confident = A.skip_confident;
  uint32_t* dbg = A.dbg;
  const uint32_t dbg_words = A.dbg_words;

  // segment arrays in the scratch
  uint32_t* s_node = scr;            // entry node
  uint32_t* s_dl = s_node + CAP;     // entry depth | length << 16
  uint32_t* s_cnt = s_dl + CAP;      // path count of every state of the segment
  uint32_t* s_p01 = s_cnt + CAP;     // parents (segment ids, 16 bits each, 0xFFFF = none)
  uint32_t* s_p23 = s_p01 + CAP;
  uint32_t* s_gen = s_p23 + CAP;     // generation (= round of phase B)
  uint32_t* s_stop = s_gen + CAP;    // stop depths of the entry: lowest | highest << 16
  uint32_t* gq = s_stop + CAP;       // phase A: two queues of QCAP nodes; then the packed entries (u64)
  // LDS
  uint32_t* sh = lds + (SEGW_LDS_WORDS - SH_WORDS);
  uint32_t* l_seed = sh - 32;        // left-flank seeds by depth
  uint32_t* s_aux = lds;             // (phase D1) generation | closure marks of the children << 16; later: emit offset
  uint32_t* s_t = lds + CAP;         // (phase D1) last closure state: towards a sink | from a traceback start << 16

  const uint32_t tid = threadIdx.x;
  const int lane = (int)(tid & 63u);
  const uint32_t wave = tid >> 6;
  const uint32_t gi = uni(A.gap_ids[x]);
  const GapDev gd = A.gaps.load(gi);
  GapOut* go = &A.outs[gi];
  const uint32_t* lseeds = A.flank_nodes + gd.flank_off;
  const uint32_t* rseeds = lseeds + (uint32_t)(gd.lmf + 1);
  const uint32_t* targets = rseeds + (uint32_t)(gd.rmf + 1);
  const int D = gd.D, lmf = gd.lmf, rmf = gd.rmf;
  const unsigned long long cyc0 = __builtin_amdgcn_s_memtime();

  uint32_t lflags = 0;      // flags this thread raises (joined at the end of a phase)
#ifdef G2S_SEGW_PROFILE
  unsigned long long pw_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // phase A: records | label + proposals | barriers;  phase B: scan + records | horizon barrier | selected events |
  // children | end barrier;  tail: hits + Q7 | D1 | D2 + recount | emission
  uint32_t pa_rec = 0, pa_prop = 0, pa_bar = 0, pb_scan = 0, pb_b0 = 0, pb_sel = 0, pb_child = 0, pb_b3 = 0, pb_pack = 0;
#if G2S_SEGW_PROFILE == 2
  uint32_t pf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long pf_last = 0;
#endif
#endif
  bool stuck = false;       // (per lane) a probe ran past its bound
  const bool overflow0 = D >= 32767 || rmf + 1 > 32 || lmf + 1 > 32;

  // every wave keeps the target k-mers in its lanes: lane j = target j
  const uint32_t tg = (lane <= rmf && lane < 32) ? targets[lane] : G2S_DEV_INVALID;
  if (tid < SH_WORDS) sh[tid] = (tid == SH_H0 || tid == SH_H1 || tid == SH_BEST) ? SEG_INF : 0u;
  if (tid >= 64u && tid < 96u) l_seed[tid - 64u] = (int)(tid - 64u) <= lmf ? lseeds[tid - 64u] : G2S_DEV_INVALID;

  // The publication of a gap's results on the host path (see fill_seg.hip, `publish`): every wave's stores are
  // waited for, then wave 0 copies the record and releases it at system scope.  Nothing is announced in resident mode.
  auto publish = [&]() {
    if (A.resident) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave != 0u) return;
    __threadfence();
    if ((uint32_t)lane < sizeof(GapOut) / 4u)
      ((uint32_t*)&A.outs_host[gi])[lane] = __hip_atomic_load(&((const uint32_t*)go)[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    if (lane == 0) {
      const unsigned long long at = atomicAdd(out_counter + 1, 1ull);
      __hip_atomic_store(&A.done_list[at], gi, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  };

  // ---------------- phase A: the right set (:871-982) as (entry node, depth label) pairs ----------------
  // Label-correcting search over unitigs, thread = queued entry: one exit record and the entry's label per entry,
  // four proposals through the table with 64-bit atomic min on (node << 32 | label).
  uint64_t* tab = (uint64_t*)lds;
  for (uint32_t i = tid; i < AS; i += NT) tab[i] = SEGX_EMPTY64;
  __syncthreads();
  auto a_hash = [&](uint32_t p) -> uint32_t { uint32_t h = p; h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16; return h & (AS - 1u); };
  // propose label dp for node p; true when the label improved (the caller queues p)
  auto relabel = [&](bool active, uint32_t p, uint32_t dp) -> bool {
    bool improved = false, fresh = false;
    if (active) {
      const uint64_t key = ((uint64_t)p << 32) | dp;
      uint32_t h = a_hash(p), guard = 0;
      while (true) {
        if (++guard > 4u * AS) { stuck = true; break; }
        const uint64_t c = tab[h];
        if ((uint32_t)(c >> 32) == p) {
          improved = atomicMin((unsigned long long*)&tab[h], (unsigned long long)key) > key;
          break;
        }
        if (c == SEGX_EMPTY64) {
          const unsigned long long prev = atomicCAS((unsigned long long*)&tab[h], (unsigned long long)SEGX_EMPTY64, (unsigned long long)key);
          if (prev == SEGX_EMPTY64) { improved = true; fresh = true; break; }
          continue;  // somebody took the slot: look at it again
        }
        h = (h + 1u) & (AS - 1u);
      }
    }
    const uint64_t fm = __ballot(fresh);
    if (fm && lane == __builtin_ctzll(fm)) atomicAdd(&sh[SH_NA], (uint32_t)__popcll(fm));
    return improved;
  };
  uint32_t roundsA = 0;
  if (!overflow0) {
    if (wave == 0u) {  // seeds: right.substr(len-k-j, k) enters at depth j (:878-884, :953-976)
      const uint32_t sd = (lane <= rmf && lane < 32) ? rseeds[lane] : G2S_DEV_INVALID;
      const bool imp = relabel(sd != G2S_DEV_INVALID && lane <= gd.right_half, sd, (uint32_t)lane);
      const uint32_t pos = wave_reserve(imp, &sh[SH_NQ0], lane);
      if (imp) gq[pos] = sd;
    }
    __syncthreads();
    uint32_t cur = 0;
    while (true) {
      const uint32_t ne = sh[SH_NQ0 + cur];
      if (ne == 0u || sh[SH_OVF]) break;
      if (++roundsA > 65535u) { lflags |= G2S_DEV_OVERFLOW_A | G2S_DEV_WATCHDOG; if (tid == 0) sh[SH_OVF] = 1u; break; }
      const uint32_t* qc = gq + cur * QCAP;
      uint32_t* qn = gq + (cur ^ 1u) * QCAP;
      uint32_t* nq = &sh[SH_NQ0 + (cur ^ 1u)];
      // thread = (entry, predecessor slot): a round has a dozen entries on average, four proposals each
      for (uint32_t i0 = 0; i0 < 4u * ne; i0 += NT) {
        WPROF(0);
        const uint32_t e = (i0 + tid) >> 2, q = tid & 3u;
        const bool mine = e < ne;
        const uint32_t v = mine ? qc[e] : 0u;
        // walking back from v = walking on from v^1: steps left in the unitig and the successor record of the
        // walk's last node (graph.predecessors(last)[i] = succ(last^1)[i] ^ 1) in one record
        uint32_t w = G2S_DEV_INVALID, r = 0;
        if (mine) {
          const uint32_t* u = urec + (size_t)(v ^ 1u) * 8;
          w = u[q];
          r = u[4];
        }
#ifdef G2S_SEGW_PROFILE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        WPROF(1);
        WPROF_ADD(pa_rec, 0, 1);
        uint32_t d = 0;
        if (mine) {  // the entry's current label (the record travels meanwhile)
          uint32_t slot = a_hash(v), guard = 0;
          while ((uint32_t)(tab[slot] >> 32) != v) {
            if (++guard > AS) { stuck = true; break; }
            slot = (slot + 1u) & (AS - 1u);
          }
          d = (uint32_t)tab[slot];
        }
        const uint32_t steps = min(r, (uint32_t)gd.right_half - d);
        const bool live = mine && !stuck && d + steps < (uint32_t)gd.right_half;  // (then steps == r: the record is the last node's)
        const uint32_t dchild = d + steps + 1u;
        const bool imp = relabel(live && w != G2S_DEV_INVALID, w ^ 1u, dchild);
        const uint32_t pos = wave_reserve(imp, nq, lane);
        if (imp && pos < QCAP) qn[pos] = w ^ 1u;
        WPROF(2);
        WPROF_ADD(pa_prop, 1, 2);
      }
      WPROF(3);
      __syncthreads();
      if (tid == 0) {
        sh[SH_NQ0 + cur] = 0u;
        // (at most EA + NT * 4 of the AS slots are ever taken, so every probe ends)
        if (sh[SH_NQ0 + (cur ^ 1u)] > QCAP || sh[SH_NA] > EA) { sh[SH_OVF] = 1u; atomicOr(&sh[SH_FLAGS], G2S_DEV_OVERFLOW_A | G2S_DEV_WHY_RS); }
      }
      if (__ballot(stuck) && lane == 0) { sh[SH_OVF] = 1u; atomicOr(&sh[SH_FLAGS], G2S_DEV_OVERFLOW_A | G2S_DEV_WATCHDOG); }
      cur ^= 1u;
      __syncthreads();
      WPROF(4);
      WPROF_ADD(pa_bar, 3, 4);
    }
  }
  __syncthreads();
  WPROF(5);
  bool overflow = overflow0 || sh[SH_OVF] != 0u;
  uint32_t nA = sh[SH_NA];
  uint32_t M = 0;  // the right set as M disjoint, sorted index intervals (merged in place: ivw[2 i] = last index, ivw[2 i + 1] = first)
  uint32_t* ivw = lds;
  if (!overflow) {
    // the entries, packed (through the scratch: the queues are idle), then entry -> k-mer index interval
    // (lo << 32 | orientation << 31 | hi): one record load each; sorted and merged into disjoint intervals
    uint64_t* pk = (uint64_t*)gq;
    for (uint32_t i = tid; i < AS; i += NT) {
      const uint64_t e = tab[i];
      const bool have = e != SEGX_EMPTY64;
      const uint32_t pos = wave_reserve(have, &sh[SH_PK], lane);
      if (have) pk[pos] = e;
    }
    __syncthreads();
    nA = sh[SH_PK];
    uint32_t n2 = 2;
    while (n2 < nA) n2 <<= 1;
    uint32_t acc_vis = 0, acc_xa = 0;
    uint32_t* o = dbg ? dbg + (size_t)x * dbg_words : nullptr;
    for (uint32_t e = tid; e < n2; e += NT) {
      if (e < nA) {
        const uint64_t ent = pk[e];
        const uint32_t v = (uint32_t)(ent >> 32), label = (uint32_t)ent;
        const uint32_t r = urec[(size_t)(v ^ 1u) * 8 + 4];
        const uint32_t steps = min(r, (uint32_t)gd.right_half - label);
        const uint32_t w0 = v ^ 1u, idx = w0 >> 1;
        const uint32_t lo = (w0 & 1u) ? idx - steps : idx, hi = (w0 & 1u) ? idx : idx + steps;
        tab[e] = ((uint64_t)lo << 32) | ((uint64_t)(v & 1u) << 31) | hi;
        acc_vis += steps + 1u;  // (intervals of one unitig may overlap: an upper bound of the set's size)
        acc_xa += min(steps + 1u, (uint32_t)gd.right_half - label);
        if (o && 9u + 2u * e < dbg_words) { o[8u + 2u * e] = v; o[9u + 2u * e] = label; }
      } else {
        tab[e] = SEGX_EMPTY64;
      }
    }
    acc_vis = wave_sum(acc_vis);
    acc_xa = wave_sum(acc_xa);
    if (lane == 0) { atomicAdd(&sh[SH_NVIS], acc_vis); atomicAdd(&sh[SH_XA], acc_xa); }
    __syncthreads();
    block_sort64(tab, n2, tid);
    if (wave == 0u) {
      // Q7 in the right set, conservatively as in the regular tier: an entry of each orientation with overlapping intervals
      bool cross = false;
      const uint32_t m = lds_merge_intervals(tab, nA, lane, &cross);
      if (lane == 0) { sh[SH_M] = m; if (cross) atomicOr(&sh[SH_FLAGS], G2S_DEV_Q7_A); }
    }
    __syncthreads();
    M = sh[SH_M];
  }
  const uint32_t nvis = sh[SH_NVIS], xa = sh[SH_XA];
  const unsigned long long cyc1 = __builtin_amdgcn_s_memtime();
  WPROF(6);
  WPROF_ADD(pb_pack, 5, 6);

  // ---------------- phase B: pending events in LDS slots, segments into the scratch ----------------------
  // LDS (words): right set: firsts [EA + 16] | lasts [EA] | sampled levels of the firsts (every 8th, 64th, 512th,
  //   4096th) [1024] | slots: node, depth|fixed<<15|states to the unitig's end<<16, count, p01, p23, stop lo, stop hi,
  //   parents, right-set interval of the entry k-mer lo, hi [PE each], exit record [4 PE] | ht [2 HS] | ring of free
  //   slots [PE] | the round's final events [PE]
  constexpr uint32_t IVF = 0u, IVL = EA + 16u, LV1 = IVL + EA, LV2 = LV1 + 784u, LV3 = LV2 + 112u, LV4 = LV3 + 32u, SLOT0 = LV1 + 1024u;
  static_assert((EA % 8u) == 0u && EA <= 8u * 776u && LV4 + 16u <= SLOT0 && (SLOT0 % 4u) == 0u, "levels of the right set");
  uint32_t* ivF = lds + IVF;
  uint32_t* ivL = lds + IVL;
  uint32_t* e_node = lds + SLOT0;  // G2S_DEV_INVALID = the slot is free
  uint32_t* e_de = e_node + PE;    // depth | fixed << 15 | min(states to the unitig's end, 0x7FFF) << 16 (0: record not loaded yet)
  uint32_t* e_cnt = e_de + PE;
  uint32_t* e_p01 = e_cnt + PE;
  uint32_t* e_p23 = e_p01 + PE;
  uint32_t* e_slo = e_p23 + PE;
  uint32_t* e_shi = e_slo + PE;
  uint32_t* e_np = e_shi + PE;     // parents so far
  uint32_t* e_clo = e_np + PE;     // the right-set interval that holds the entry k-mer (0xFFFFFFFF: not looked up)
  uint32_t* e_chi = e_clo + PE;
  uint4* e_rec = (uint4*)(e_chi + PE);
  uint64_t* ht = (uint64_t*)(e_rec + PE);
  uint32_t* fq = (uint32_t*)(ht + HS);
  uint32_t* sell = fq + PE;  // the final events of the round (slots)
  static_assert(SLOT0 + 10u * PE + 4u * PE + 2u * HS + PE + PE + 32u + SH_WORDS <= SEGW_LDS_WORDS, "LDS layout of the large variant");
  static_assert((PE & (PE - 1u)) == 0u && (HS & (HS - 1u)) == 0u, "ring and table sizes are powers of two");
  if (!overflow) {
    // ---- the merged intervals (interleaved, at the start of the LDS) as two dense arrays, and the sampled levels:
    // a search reads one block of eight firsts per level and counts those at or below the key
    uint32_t* tmp = lds + 16384u;  // (free: the sort used the first 16 384 words at most)
    for (uint32_t i = tid; i < M; i += NT) { tmp[i] = ivw[2u * i + 1u]; tmp[8192u + i] = ivw[2u * i]; }
    __syncthreads();
    for (uint32_t i = tid; i < EA + 16u; i += NT) ivF[i] = i < M ? tmp[i] : 0x7FFFFFFFu;
    for (uint32_t i = tid; i < M; i += NT) ivL[i] = tmp[8192u + i];
    __syncthreads();
    for (uint32_t i = tid; i < 784u; i += NT) lds[LV1 + i] = 8u * i + 7u < EA + 16u ? ivF[8u * i + 7u] : 0x7FFFFFFFu;
    __syncthreads();
    for (uint32_t i = tid; i < 112u; i += NT) lds[LV2 + i] = 8u * i + 7u < 784u ? lds[LV1 + 8u * i + 7u] : 0x7FFFFFFFu;
    __syncthreads();
    if (tid < 32u) lds[LV3 + tid] = 8u * tid + 7u < 112u ? lds[LV2 + 8u * tid + 7u] : 0x7FFFFFFFu;
    __syncthreads();
    if (tid < 16u) lds[LV4 + tid] = 8u * tid + 7u < 32u ? lds[LV3 + 8u * tid + 7u] : 0x7FFFFFFFu;
    __syncthreads();
  }
  // number of firsts at or below q among the eight at p (16-byte aligned).  k-mer indices and the padding
  // (0x7FFFFFFF) are below 2^31, so a first above q leaves the sign bit of (q - first) set: a subtraction and a
  // shift per sample, the sums as three-operand adds — a comparison per sample was three instructions (the compare,
  // a wait state for its mask, the add with carry).
  auto cnt8 = [&](const uint32_t* p, uint32_t q) -> uint32_t {
    const uint4 a = ((const uint4*)p)[0], b = ((const uint4*)p)[1];
    const uint32_t above = ((q - a.x) >> 31) + ((q - a.y) >> 31) + ((q - a.z) >> 31) + ((q - a.w) >> 31) +
                           ((q - b.x) >> 31) + ((q - b.y) >> 31) + ((q - b.z) >> 31) + ((q - b.w) >> 31);
    return 8u - above;
  };
  // The first level a search reads is one block of eight, the same for every key: kept in registers for the gap
  // (one LDS round trip less per search: a round is a chain of them).
  uint4 top_a = make_uint4(0x7FFFFFFFu, 0x7FFFFFFFu, 0x7FFFFFFFu, 0x7FFFFFFFu), top_b = top_a;
  if (!overflow) {
    const uint32_t* tp = M > 4096u ? lds + LV4 : M > 512u ? lds + LV3 : M > 64u ? lds + LV2 : M > 8u ? lds + LV1 : ivF;
    top_a = ((const uint4*)tp)[0];
    top_b = ((const uint4*)tp)[1];
  }
  // number of intervals that begin at or before k-mer index q, per lane (wave-uniform control flow: M is uniform)
  auto iv_rank = [&](uint32_t q) -> uint32_t {
    q = min(q, 0x7FFFFFFEu);  // (cnt8 wants keys below the padding)
    uint32_t blk = 8u - (((q - top_a.x) >> 31) + ((q - top_a.y) >> 31) + ((q - top_a.z) >> 31) + ((q - top_a.w) >> 31) +
                         ((q - top_b.x) >> 31) + ((q - top_b.y) >> 31) + ((q - top_b.z) >> 31) + ((q - top_b.w) >> 31));
    if (M > 4096u) blk = 8u * blk + cnt8(lds + LV3 + 8u * blk, q);
    if (M > 512u) blk = 8u * blk + cnt8(lds + LV2 + 8u * blk, q);
    if (M > 64u) blk = 8u * blk + cnt8(lds + LV1 + 8u * blk, q);
    if (M > 8u) blk = 8u * blk + cnt8(ivF + 8u * blk, q);
    return blk;
  };
  uint32_t acc_sb = 0, acc_xb = 0;
  uint32_t gen = 0;
  // (both strands of a k-mer at one depth share a probe sequence: the walk that inserts one passes the other — Q7)
  auto e_hash = [&](uint32_t w, uint32_t dw) -> uint32_t {
    uint32_t h = (w >> 1) * 0x9E3779B1u ^ dw * 0x85EBCA6Bu;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h & (HS - 1u);
  };
  uint32_t tail_snap = 0;  // the ring's tail when the round began: pops stay below it
  // Proposals of the lanes of this wave: event (w, dw) gains count c from segment par (0xFFFF: a seed, no parent);
  // (clo, chi): the right-set interval that holds w's k-mer (0xFFFFFFFF: not looked up); `at`: the lane's position in
  // the ring of free slots (reserved by the caller).  Written for the instruction count — a wave alone on its SIMD
  // issues an instruction every four to five cycles, and a round of the search is a chain: loops are wave-uniform
  // with a per-lane STATE word (0 pending, 1 new event, 2 merged), the rare cases behind wave-uniform branches.
  auto ev_insert = [&](bool act, uint32_t w, uint32_t dw, uint32_t c, uint32_t par, uint32_t pslo, uint32_t pshi, bool fixed,
                       uint32_t clo, uint32_t chi, uint32_t at) {
    if (ballot_and(act, (int32_t)(tail_snap - at) <= 0)) {  // the ring has nothing left from before this round
      if (act && (int32_t)(tail_snap - at) <= 0) { act = false; lflags |= G2S_DEV_OVERFLOW_B | G2S_DEV_WHY_FRONTIER; sh[SH_OVF] = 1u; }
    }
    const uint32_t slot = fq[at & (PE - 1u)];
    uint32_t slo = pslo, shi = pshi;
    bool src = false;
    if (ballot_and(act, dw <= (uint32_t)lmf)) {  // :1270 (the first rounds only)
      const uint32_t ls = (act && dw <= (uint32_t)lmf) ? l_seed[dw] : G2S_DEV_INVALID;
      src = ls != G2S_DEV_INVALID && (w >> 1) == (ls >> 1);
      if (src) { slo = dw; shi = dw; }
    }
    const uint32_t klo = dw << 16;  // the key: node in the high word, depth in the upper half of the low word (the slot below)
    // The exit record of the event this child may become (states to the end of its unitig, where the segment leaves)
    // is asked for NOW — before the walk along the table tells whether the child is a new event or merges into a
    // pending one: the record travels while the walk's LDS round trips go on, and lands in the slot at the end of this
    // call (the next round's scan finds it there instead of waiting for memory).  A merged child's record is dropped.
    uint4 nrec = make_uint4(G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID, G2S_DEV_INVALID);
    uint32_t nes = 1u;
    if (act) {
      if ((int)dw < lmf) nrec = *(const uint4*)(succ + (size_t)w * 4);  // above the flank: one state
      else { const uint4* u = (const uint4*)(urec + (size_t)w * 8); nrec = u[0]; nes = u[1].x; }
    }
    if (act) {
      // (the node last: the slot's owner may be looking at it right now, and a slot whose node it can see must show the
      // new depth — at or above the horizon — whichever of the two words it reads first: freed slots keep depth 0x7FFF)
      e_de[slot] = dw | (fixed ? 0x8000u : 0u);
      e_cnt[slot] = c;
      e_p01[slot] = 0xFFFF0000u | par;
      e_p23[slot] = 0xFFFFFFFFu;
      e_slo[slot] = slo;
      e_shi[slot] = shi;
      e_np[slot] = par != SEG_NOPAR ? 1u : 0u;
      e_clo[slot] = clo;
      e_chi[slot] = chi;
      e_node[slot] = w;
    }
    WFINE(4);
    // (dead entries stay in the table until it is swept: they match no key — all of those carry depths at or above
    // the horizon their events were selected under — so a key sits at or before the first empty position of its
    // probe sequence, and two lanes with one key meet at that position)
    uint32_t st = act ? 0u : 3u, mslot = 0, pos = e_hash(w, dw);
    bool q7 = false;
#pragma nounroll
    for (uint32_t guard = 0; guard < 2u * HS; guard++) {
      uint2 cc = *(const uint2*)&ht[pos];  // (.x: depth << 16 | slot, .y: node)
      if (st == 0u && (cc.x & cc.y) == 0xFFFFFFFFu) {
        const unsigned long long prev = atomicCAS((unsigned long long*)&ht[pos], (unsigned long long)SEGX_EMPTY64,
                                                  ((unsigned long long)w << 32) | (klo | slot));
        cc.x = (uint32_t)prev; cc.y = (uint32_t)(prev >> 32);  // (taken meanwhile: what the other lane put there is looked at now)
        if (prev == SEGX_EMPTY64) st = 1u;
      }
      const bool samed = (cc.x & 0xFFFF0000u) == klo;
      const bool hit = st == 0u && cc.y == w && samed;
      // Q7: the other strand pending at this depth.  A new event's walk passes every entry of the sequence up to the
      // first empty position, so of two lanes that insert the two strands the one that comes second sees the other
      // (its own claim included: a lost compare-and-swap returns what the winner put there).
      q7 |= cc.y == (w ^ 1u) && samed;
      mslot = hit ? (cc.x & 0xFFFFu) : mslot;
      st = hit ? 2u : st;
      pos = st == 0u ? ((pos + 1u) & (HS - 1u)) : pos;
      if (!__ballot(st == 0u)) break;
    }
    if (__ballot(st == 0u)) { if (st == 0u) { stuck = true; st = 3u; e_node[slot] = G2S_DEV_INVALID; } }
    WFINE(5);
    const uint64_t mm = __ballot(st == 2u);
    if (mm) {
      const bool merged = st == 2u;
      if (merged) {
        atomicAdd(&e_cnt[mslot], c);  // (at most four parents of at most 2^30 - 1 each: no wrap; clamped when read)
        if (par != SEG_NOPAR) {
          const uint32_t kk = atomicAdd(&e_np[mslot], 1u);  // (a state has at most four predecessors)
          uint32_t* pw = (kk & 2u) ? &e_p23[mslot] : &e_p01[mslot];
          atomicAnd(pw, (kk & 1u) ? (0x0000FFFFu | (par << 16)) : (0xFFFF0000u | par));
        }
        if (!src) { atomicMin(&e_slo[mslot], pslo); atomicMax(&e_shi[mslot], pshi); }
        e_node[slot] = G2S_DEV_INVALID;  // the candidate slot was not needed
      }
      uint32_t back = 0;
      if (lane == 0) back = atomicAdd(&sh[SH_QTAIL], (uint32_t)__popcll(mm));
      back = uni(back) + (uint32_t)__popcll(mm & below(lane));
      if (merged) fq[back & (PE - 1u)] = slot;
    }
    const uint64_t fm = __ballot(st == 1u);
    if (fm && lane == 0) atomicAdd(&sh[SH_NPEND], (uint32_t)__popcll(fm));
    // (nothing is computed on the record before this point: a use further up makes the wave wait for the load where it
    // was asked for)
    asm volatile("" : "+v"(nes) : : "memory");
    WFINE(6);
    if (q7 && st == 1u) lflags |= G2S_DEV_Q7_B;
    if (st == 1u) {
      e_rec[slot] = nrec;
      e_de[slot] = dw | (fixed ? 0x8000u : 0u) | (((int)dw < lmf ? 1u : min(nes + 1u, 0x7FFFu)) << 16);
    }
    WFINE(7);
  };
  if (!overflow) {
    for (uint32_t i = tid; i < HS; i += NT) ht[i] = SEGX_EMPTY64;
    for (uint32_t i = tid; i < PE; i += NT) { fq[i] = i; e_node[i] = G2S_DEV_INVALID; e_de[i] = 0x7FFFu; }
    if (tid == 0) { sh[SH_QHEAD] = 0u; sh[SH_QTAIL] = PE; }
    __syncthreads();
    tail_snap = PE;
    if (wave == 0u) {
      // left seeds: left.substr(d, k) enters at depth d with the value 1 ASSIGNED (:995-1015, :1082-1105)
      const uint32_t sd = lane <= lmf ? lseeds[lane] : G2S_DEV_INVALID;
      const uint32_t s0 = rl(sd, 0);
      // the usual flank is a stretch of ONE unitig: one segment, the seed at depth lmf the only pending event
      // (see fill_seg.hip)
      bool chain = lmf >= 1 && s0 != G2S_DEV_INVALID && __ballot(lane <= lmf && sd != seg_node(s0, (uint32_t)lane)) == 0ull;
      if (chain) chain = uni(urec[(size_t)s0 * 8 + 4]) >= (uint32_t)lmf;
      if (chain) chain = __ballot(seg_pos(s0, (uint32_t)lmf, tg) >= 0) == 0ull;
      if (chain) {
        if (lane == 0) {
          s_node[0] = s0; s_dl[0] = (uint32_t)lmf << 16; s_cnt[0] = 1u; s_p01[0] = s_p23[0] = 0xFFFFFFFFu; s_gen[0] = 0u;
          s_stop[0] = 0x7FFFu;  // (holds no target k-mer: checked above)
          sh[SH_NSEG] = 1u;
          sh[SH_CHAIN] = (uint32_t)lmf;  // the chain's states and expansions
        }
        sh[SH_QHEAD] = 1u;  // (the ring hands out slots 0, 1, ... at first)
        ev_insert(lane == lmf, sd, (uint32_t)lmf, 1u, 0u, (uint32_t)lmf, (uint32_t)lmf, true, 0xFFFFFFFFu, 0u, 0u);
      } else {
        const bool sa = sd != G2S_DEV_INVALID && lane <= D;
        const uint64_t sm = __ballot(sa);
        sh[SH_QHEAD] = (uint32_t)__popcll(sm);
        ev_insert(sa, sd, (uint32_t)lane, 1u, SEG_NOPAR, (uint32_t)lane, (uint32_t)lane, true, 0xFFFFFFFFu, 0u, (uint32_t)__popcll(sm & below(lane)));
      }
    }
    __syncthreads();
    gen = sh[SH_NSEG] ? 1u : 0u;
    uint32_t par = 0;  // which of the two horizon words (and list counters) this round uses
    while (true) {
      // ---- (nothing is in flight here: the counters are stable)
      const uint32_t npend = sh[SH_NPEND];
      if (npend == 0u || sh[SH_OVF]) break;
      tail_snap = sh[SH_QTAIL];
      if (gen > 65000u) { lflags |= G2S_DEV_OVERFLOW_B | G2S_DEV_WATCHDOG; if (tid == 0) sh[SH_OVF] = 1u; break; }
      if (sh[SH_NDEAD] > HS / 8u) {  // dead entries pile up: the table again from the live slots
        __syncthreads();
        for (uint32_t i = tid; i < HS; i += NT) ht[i] = SEGX_EMPTY64;
        if (tid == 0) sh[SH_NDEAD] = 0u;
        __syncthreads();
        for (uint32_t s = tid; s < PE; s += NT) {
          const uint32_t w = e_node[s];
          const uint32_t dw = e_de[s] & 0x7FFFu;
          const uint64_t ent = ((uint64_t)w << 32) | ((uint64_t)dw << 16) | s;
          uint32_t pos = e_hash(w, dw);
          bool pend = w != G2S_DEV_INVALID;
          for (uint32_t g3 = 0; __ballot(pend); g3++) {
            if (g3 > 2u * HS) { if (pend) stuck = true; break; }
            if (pend && atomicCAS((unsigned long long*)&ht[pos], (unsigned long long)SEGX_EMPTY64, (unsigned long long)ent) == SEGX_EMPTY64) pend = false;
            pos = (pos + 1u) & (HS - 1u);
          }
        }
        __syncthreads();
      }
      // ---- own slots: one round trip for the events created last round (states to the end of the unitig, exit
      // record), and the horizon: min over pending events of (depth + states to the end of the unitig)
      WPROF(0);
      static_assert(PE == 2u * SEGW_NT, "a thread owns two slots");
      uint32_t own_d[2];  // depth of the live event in own slot k (0x7FFF: none)
      uint32_t hmin = SEG_INF;
#pragma unroll
      for (int k2 = 0; k2 < 2; k2++) {
        const uint32_t s = tid + (uint32_t)k2 * NT;
        const uint32_t node = e_node[s];
        uint32_t de = e_de[s];
        const bool live = node != G2S_DEV_INVALID;
        const uint32_t dd = de & 0x7FFFu;
        if (live && (de >> 16) == 0u) {
          uint4 r0;
          uint32_t es;
          if ((int)dd < lmf) { r0 = *(const uint4*)(succ + (size_t)node * 4); es = 1u; }  // above the flank: one state
          else { const uint4* u = (const uint4*)(urec + (size_t)node * 8); r0 = u[0]; es = min(u[1].x + 1u, 0x7FFFu); }
          e_rec[s] = r0;
          de |= es << 16;
          e_de[s] = de;
        }
        own_d[k2] = live ? dd : 0x7FFFu;
        if (live) hmin = min(hmin, dd + (de >> 16));
      }
      hmin = wave_min(hmin);
      if (lane == 0 && hmin != SEG_INF) atomicMin(&sh[SH_H0 + par], hmin);
      WPROF(1);
      __syncthreads();
      WPROF(2);
      WPROF_ADD(pb_scan, 0, 1);
      WPROF_ADD(pb_b0, 1, 2);
      const uint32_t H = sh[SH_H0 + par];
      // ---- final events (depth below the horizon): gathered into a list by their slots' owners
      if (tid == 0) { sh[SH_H0 + (par ^ 1u)] = SEG_INF; sh[SH_NSEL + (par ^ 1u)] = 0u; }
      {
        const bool m0 = own_d[0] < H, m1 = own_d[1] < H;
        const uint64_t b0 = __ballot(m0), b1 = __ballot(m1);
        if (b0 | b1) {
          uint32_t at = 0;
          if (lane == 0) at = atomicAdd(&sh[SH_NSEL + par], (uint32_t)(__popcll(b0) + __popcll(b1)));
          at = rl(at, 0);
          if (m0) sell[at + (uint32_t)__popcll(b0 & below(lane))] = tid;
          if (m1) sell[at + (uint32_t)__popcll(b0) + (uint32_t)__popcll(b1 & below(lane))] = tid + NT;
        }
      }
      WPROF(8);
      __syncthreads();
      WPROF(3);
      WPROF_ADD(pb_b0, 8, 3);
      const uint32_t nsel = sh[SH_NSEL + par];
      if (tid == 0) { atomicSub(&sh[SH_NPEND], nsel); atomicAdd(&sh[SH_NDEAD], nsel); }
      // ---- lane = (final event, successor slot): sixteen events per wave and pass.  The four lanes of an event read
      // its fields together; lane 0 of the four writes the segment and frees the slot; every lane tests its child
      // against the pruning rule (:1050) and inserts it.
#pragma unroll 1
      for (uint32_t base = 16u * wave; base < nsel; base += 16u * SEGW_NW) {
        WFINE_START();
        const uint32_t ei = base + ((uint32_t)lane >> 2), q = (uint32_t)lane & 3u;
        const bool mine = ei < nsel;
        const uint32_t s = sell[mine ? ei : base];  // (lanes without an event read the pass's first: their results are not used)
        const uint32_t en = e_node[s], de = e_de[s];
        const uint32_t w = mine ? ((const uint32_t*)e_rec)[4u * s + q] : G2S_DEV_INVALID;
        const uint32_t cnt = (de & 0x8000u) ? 1u : min(e_cnt[s], (uint32_t)G2S_DEV_MAX_PATHS);
        const uint32_t slo = e_slo[s], shi = e_shi[s];
        const uint32_t p01 = e_p01[s], p23 = e_p23[s];
        const uint32_t clo = e_clo[s], chi = e_chi[s];
        const int ed = (int)(de & 0x7FFFu);
        const uint32_t es = de >> 16;
        const bool lead = mine && q == 0u;
        WFINE(0);
        // ---- length under the pruning rule (:1050).  An event that was inserted under the rule knows the right-set
        // interval [clo, chi] around its entry k-mer: the run from its second state on stays inside the set as far as
        // that interval goes (merged intervals have holes between them).  An event from above the rule's depth whose
        // run reaches it (each path has one) searches for its first state under the rule.
        const uint32_t lcap = min(es, (uint32_t)(D - ed + 1));
        const bool pr = lcap > 1u && ed + (int)lcap - 1 >= gd.prune_from;
        const uint32_t idx0 = en >> 1;
        const uint32_t l_up = (idx0 + 1u <= chi ? min(chi, idx0 + lcap - 1u) : idx0) - idx0 + 1u;
        const uint32_t l_dn = idx0 - ((idx0 >= 1u && idx0 - 1u >= clo) ? max(clo, idx0 - (lcap - 1u)) : idx0) + 1u;
        uint32_t L = (pr && clo != 0xFFFFFFFFu) ? ((en & 1u) ? l_dn : l_up) : lcap;  // (known interval: ed >= prune_from, the rule holds from the second state on)
        if (ballot_and(mine, pr, clo == 0xFFFFFFFFu)) {
          const uint32_t t1 = (uint32_t)max(1, gd.prune_from - ed);
          const uint32_t q0 = (en & 1u) ? idx0 - t1 : idx0 + t1;  // first state entered under the rule
          const uint32_t r = iv_rank(q0);
          if (pr && clo == 0xFFFFFFFFu) {
            const uint32_t r1 = r > 0u ? r - 1u : 0u;
            const bool in = r > 0u && q0 <= ivL[r1];
            if (!(en & 1u)) L = (in ? min(ivL[r1], idx0 + lcap - 1u) : q0 - 1u) - idx0 + 1u;
            else L = idx0 - (in ? max(ivF[r1], idx0 - (lcap - 1u)) : q0 + 1u) + 1u;
          }
        }
        WFINE(1);
        // ---- segments that reached the end of their stretch leave through the successor table; a child at or below
        // the rule's depth has to be in the right set (:1050): one search of the sorted intervals per lane
        const bool exits = mine && L == es && ed + (int)L - 1 < D;
        const uint32_t xd = (uint32_t)ed + L;  // depth of the children
        bool cact = exits && w != G2S_DEV_INVALID;
        uint32_t cclo = 0xFFFFFFFFu, cchi = 0u;
        if (ballot_and(cact, (int)xd >= gd.prune_from)) {
          const bool cm = cact && (int)xd >= gd.prune_from;
          const uint32_t r = iv_rank(w >> 1);
          const uint32_t r1 = r > 0u ? r - 1u : 0u;
          const uint32_t fl = ivF[r1], ll = ivL[r1];
          const bool in = r > 0u && (w >> 1) <= ll;
          cclo = cm ? fl : cclo;
          cchi = cm ? ll : cchi;
          cact = cact && (!cm || in);
        }
        WFINE(3);
        // ---- one LDS operation for the pass's three reservations: segment ids (lane 0), the ring's tail for the slots
        // of the final events (lane 1), the ring's head for the candidate slots of the children (lane 2)
        const uint64_t cmk = __ballot(cact);
        const uint32_t npass = min(nsel - base, 16u);
        uint32_t rsv = 0;
        if (lane < 3) rsv = atomicAdd(&sh[lane == 0 ? SH_NSEG : lane == 1 ? SH_QTAIL : SH_QHEAD], lane == 2 ? (uint32_t)__popcll(cmk) : npass);
        const uint32_t esid = rl(rsv, 0) + ((uint32_t)lane >> 2);
        const uint32_t back = rl(rsv, 1) + ((uint32_t)lane >> 2);
        const uint32_t at = rl(rsv, 2) + (uint32_t)__popcll(cmk & below(lane));
        if (ballot_and(mine, esid >= CAP)) {  // (the gap then runs in the LDS tier)
          if (mine && esid >= CAP) { cact = false; lflags |= G2S_DEV_OVERFLOW_B | G2S_DEV_WHY_LOG; sh[SH_OVF] = 1u; }
        }
        if (lead) {
          acc_sb += L; acc_xb += min(L, (uint32_t)(D - ed));
          e_node[s] = G2S_DEV_INVALID;  // the slot is free again (through the ring: not before the next round)
          e_de[s] = 0x7FFFu;
          fq[back & (PE - 1u)] = s;
          if (esid < CAP) {
            s_node[esid] = en;
            s_dl[esid] = (uint32_t)ed | (L << 16);
            s_cnt[esid] = cnt;
            s_p01[esid] = p01;
            s_p23[esid] = p23;
            s_gen[esid] = gen;
            s_stop[esid] = slo | (shi << 16);
          }
        }
        WPROF(4);
        WPROF_ADD(pb_sel, 3, 4);
        WFINE(2);
        if (cmk) ev_insert(cact, w, xd, cnt, esid, slo, shi, false, cclo, cchi, at);
        WPROF(5);
        WPROF_ADD(pb_child, 4, 5);
        WPROF(3);
      }
      if (__ballot(stuck) && lane == 0) { sh[SH_OVF] = 1u; atomicOr(&sh[SH_FLAGS], G2S_DEV_OVERFLOW_B | G2S_DEV_WATCHDOG); }
      gen++;
      par ^= 1u;
      WPROF(6);
      __syncthreads();
      WPROF(7);
      WPROF_ADD(pb_b3, 6, 7);
    }
  }
  {  // the threads' sums and flags
    acc_sb = wave_sum(acc_sb);
    acc_xb = wave_sum(acc_xb);
    uint32_t f = lflags;
    for (int o = 32; o > 0; o >>= 1) f |= (uint32_t)__shfl_xor((int)f, o);
    if (lane == 0) { atomicAdd(&sh[SH_SB], acc_sb); atomicAdd(&sh[SH_XB], acc_xb); if (f) atomicOr(&sh[SH_FLAGS], f); }
    lflags = 0;
  }
  __syncthreads();
  overflow = overflow0 || sh[SH_OVF] != 0u;
  uint32_t flags = sh[SH_FLAGS] | G2S_DEV_BIG;
  if (overflow && !(flags & G2S_DEV_OVERFLOW_A)) flags |= G2S_DEV_OVERFLOW_B;
  const uint32_t nseg = min(sh[SH_NSEG], CAP);
  uint32_t sb = sh[SH_SB] + sh[SH_CHAIN], xb = sh[SH_XB] + sh[SH_CHAIN];
  const unsigned long long cyc2 = __builtin_amdgcn_s_memtime();

  // ---------------- Q7: an upward and a downward segment of one unitig meeting on a k-mer ------
  // The upward segments' index intervals sorted and merged in LDS; a downward segment that touches none of them (the
  // usual case) is done after one binary search; the few others are checked against every upward segment.
  // (for every gap the search finished, filled or not: the reference's outcome is in doubt either way)
  if (!overflow && !(flags & G2S_DEV_Q7_B) && nseg > 1u) {
    uint64_t* sbuf = (uint64_t*)lds;
    uint32_t* clist = lds + 2u * CAP;
    for (uint32_t b0 = 0; b0 < nseg; b0 += NT) {
      const uint32_t b = b0 + tid;
      const bool hb = b < nseg;
      const uint32_t nb_ = hb ? s_node[b] : 0u, lb = hb ? s_dl[b] >> 16 : 0u;
      const bool up = hb && !(nb_ & 1u) && lb > 0u;
      const uint32_t pos = wave_reserve(up, &sh[SH_NU], lane);
      if (up) sbuf[pos] = ((uint64_t)(nb_ >> 1) << 32) | (uint64_t)((nb_ >> 1) + lb - 1u);
      if (__ballot(hb && (nb_ & 1u) && lb > 0u) && lane == 0) sh[SH_ANYDN] = 1u;
    }
    __syncthreads();
    const uint32_t nu = sh[SH_NU];
    const bool anydn = sh[SH_ANYDN] != 0u;
    if (nu > 0u && anydn) {
      uint32_t n2 = 2;
      while (n2 < nu) n2 <<= 1;
      for (uint32_t i = nu + tid; i < n2; i += NT) sbuf[i] = SEGX_EMPTY64;
      __syncthreads();
      block_sort64(sbuf, n2, tid);
      // sorted by first index; the last indices become their running maximum: a downward segment [ib - lb + 1, ib]
      // touches an upward one iff the maximum over the intervals that begin at or before ib reaches ib - lb + 1
      uint32_t* uw = (uint32_t*)sbuf;  // uw[2 i] = last index (then: running maximum), uw[2 i + 1] = first index
      {
        uint32_t* wtot = lds + 2u * CAP + SEGW_CL_CAP;  // [SEGW_NW]
        const uint32_t K = (nu + NT - 1u) / NT;
        const uint32_t i_lo = min(nu, tid * K), i_hi = min(nu, (tid + 1u) * K);
        uint32_t run = 0;
        for (uint32_t i = i_lo; i < i_hi; i++) run = max(run, uw[2u * i]);
        uint32_t inc = run;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)inc, o); if (lane >= o) inc = max(inc, y); }
        if (lane == 63) wtot[wave] = inc;
        uint32_t excl = (uint32_t)__shfl_up((int)inc, 1);
        if (lane == 0) excl = 0u;
        __syncthreads();
        for (uint32_t w2 = 0; w2 < wave; w2++) excl = max(excl, wtot[w2]);
        run = excl;
        for (uint32_t i = i_lo; i < i_hi; i++) { run = max(run, uw[2u * i]); uw[2u * i] = run; }
      }
      __syncthreads();
      const uint32_t Mu = nu;
      const uint32_t Pu = 1u << (31 - __builtin_clz(Mu));
      for (uint32_t b0 = 0; b0 < nseg; b0 += NT) {
        const uint32_t b = b0 + tid;
        const bool hb = b < nseg;
        const uint32_t nb_ = hb ? s_node[b] : 0u, dlb = hb ? s_dl[b] : 0u;
        const int ib = (int)(nb_ >> 1), lb = (int)(dlb >> 16);
        const bool down = hb && (nb_ & 1u) && lb > 0;
        uint32_t pos = 0;
        for (uint32_t st = Pu; st; st >>= 1) {
          const uint32_t pp = pos + st;
          if (pp <= Mu && uw[2u * (pp - 1u) + 1u] <= (uint32_t)ib) pos = pp;
        }
        const bool cand = down && pos > 0u && (int)uw[2u * (pos - 1u)] >= ib - lb + 1;
        const uint32_t at = wave_reserve(cand, &sh[SH_NC], lane);
        if (cand && at < SEGW_CL_CAP) clist[at] = b;
      }
      __syncthreads();
      const uint32_t nc = sh[SH_NC];
      if (nc > SEGW_CL_CAP) {  // (never seen: the gap runs in the LDS tier)
        flags |= G2S_DEV_OVERFLOW_B | G2S_DEV_WHY_LOG;
        overflow = true;
      }
      bool hit = false;
      for (uint32_t c = 0; c < min(nc, SEGW_CL_CAP) && !hit; c++) {
        const uint32_t b = clist[c];
        const uint32_t nb_ = s_node[b], dlb = s_dl[b];
        const int ibl = (int)(nb_ >> 1), dbl = (int)(dlb & 0xFFFFu), lbl = (int)(dlb >> 16);
        for (uint32_t a = tid; a < nseg; a += NT) {
          const uint32_t na = s_node[a], dla = s_dl[a];
          const int ia = (int)(na >> 1), da = (int)(dla & 0xFFFFu), la = (int)(dla >> 16);
          const int sdiff = ibl - ia, ddiff = dbl - da;
          const int t1 = (sdiff + ddiff) >> 1, t2 = (sdiff - ddiff) >> 1;
          if (!(na & 1u) && !((sdiff + ddiff) & 1) && t1 >= 0 && t1 < la && t2 >= 0 && t2 < lbl) { hit = true; break; }
        }
      }
      if (__ballot(hit) && lane == 0) atomicOr(&sh[SH_FLAGS], G2S_DEV_Q7_B);
      __syncthreads();
      flags |= sh[SH_FLAGS] & G2S_DEV_Q7_B;
    }
    __syncthreads();
  }

  WPROF(8);
  // ---------------- phase C's hits (:1107-1159): target k-mer j at position t of a segment is a hit at depth + t --------
  // Behind the search, thread = segment and a loop over the <= 32 targets.  A hit (error, j) exists at most once
  // above and once below its base depth (a DP state is in one segment): the smallest key is a reduction, and its
  // two counts have one holder each.
  uint32_t best = SEG_INF, c1 = 0, c2 = 0, s1 = 0x7FFFu, s2 = 0x7FFFu;
  if (!overflow) {
    uint32_t mybest = SEG_INF, myc1 = 0, myc2 = 0, mys1 = 0x7FFFu, mys2 = 0x7FFFu;
    for (uint32_t b0 = 0; b0 < nseg; b0 += NT) {
      const uint32_t b = b0 + tid;
      const bool hb = b < nseg;
      const uint32_t node = hb ? s_node[b] : 0u, dl = hb ? s_dl[b] : 0u, c = hb ? s_cnt[b] : 0u, st = hb ? s_stop[b] : 0x7FFFu;
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
    const uint32_t wb = wave_min(mybest);
    if (lane == 0 && wb != SEG_INF) atomicMin(&sh[SH_BEST], wb);
    __syncthreads();
    best = sh[SH_BEST];
    if (best != SEG_INF && mybest == best) {
      if (myc1 != 0u) { sh[SH_C1] = myc1; sh[SH_S1] = mys1; }
      if (myc2 != 0u) { sh[SH_C2] = myc2; sh[SH_S2] = mys2; }
    }
    __syncthreads();
    if (best != SEG_INF) { c1 = sh[SH_C1]; c2 = sh[SH_C2]; if (c1) s1 = sh[SH_S1]; if (c2) s2 = sh[SH_S2]; }
  }

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
      uint32_t a_sb = 0, a_xb = 0;
      for (uint32_t b = tid; b < nseg; b += NT) {
        const uint32_t dl = s_dl[b];
        const int d0 = (int)(dl & 0xFFFFu), len = (int)(dl >> 16);
        a_sb += (uint32_t)max(0, min(len, d_last - d0 + 1));
        a_xb += (uint32_t)max(0, min(len, d_last - d0));
      }
      a_sb = wave_sum(a_sb);
      a_xb = wave_sum(a_xb);
      if (lane == 0) { atomicAdd(&sh[SH_BO_SB], a_sb); atomicAdd(&sh[SH_BO_XB], a_xb); }
      __syncthreads();
      sb = sh[SH_BO_SB]; xb = sh[SH_BO_XB];
    }
  }
  if (dbg) {  // diagnostics (tests): the segments of phase B (the entries of phase A were written above)
    uint32_t* o = dbg + (size_t)x * dbg_words;
    if (tid == 0) { o[0] = gi; o[1] = nA; o[2] = nseg; o[3] = flags; o[4] = roundsA; o[5] = gen; o[6] = (uint32_t)c_count; o[7] = best; }
    const uint32_t sb0 = 8u + 2u * EA;
    for (uint32_t b = tid; b < nseg; b += NT)
      if (sb0 + 6u * b + 5u < dbg_words) {
        o[sb0 + 6u * b] = s_node[b]; o[sb0 + 6u * b + 1] = s_dl[b]; o[sb0 + 6u * b + 2] = s_cnt[b];
        o[sb0 + 6u * b + 3] = s_p01[b]; o[sb0 + 6u * b + 4] = s_p23[b]; o[sb0 + 6u * b + 5] = s_gen[b];
      }
  }
  if (tid == 0) {
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
    go->stat[4] = (uint32_t)((cyc1 - cyc0) >> 8); go->stat[5] = (uint32_t)((cyc2 - cyc1) >> 8);
  }
  if (overflow || !(c_count > 0 && n_len > 0)) {  // :1169
    publish();
    return;
  }

  WPROF(9);
  // ---------------- phase D1: backward closure over the segments ---------------------------------
  const bool want_s = !skip_confident;
  const uint32_t sinknode = (want_s && gd.all_paths && rmf >= 1) ? uni(targets[rmf - 1]) : G2S_DEV_INVALID;  // Q3/Q4
  const int lo_sink = max(0, lmf + gd.g - gd.e);  // :1196
  const uint32_t reached = uni(targets[reached_j]);
  const bool t_is_s = want_s && !gd.all_paths;  // -best-only: the traceback starts are the sinks (:1245-1259)
  if (tid == 0) { sh[SH_START0] = SEG_NOPAR; sh[SH_START1] = SEG_NOPAR; sh[SH_ST0] = 0u; sh[SH_ST1] = 0u; sh[SH_CHOICE] = 0u; }
  // (in front of the pass that sets them: without this barrier a wave that found the traceback's start before thread 0
  // got here had its entry overwritten — once in a hundred lists of a campaign leg a gap lost its start segment and the
  // host's traceback stopped at its first state: round 5's fuzz campaign, part 4)
  __syncthreads();
  // what a segment holds by itself — a sink, a traceback start, a left-flank k-mer at its first state: all
  // segments at once, in front of the sweep.  s_t: ts | source << 15 | tt << 16, rewritten by the sweep.
  for (uint32_t b = tid; b < nseg; b += NT) {
    const uint32_t v0 = s_node[b], dl = s_dl[b];
    const int d0 = (int)(dl & 0xFFFFu);
    const int len = max(0, min((int)(dl >> 16), d_last - d0 + 1));
    int ts = -1, tt = -1;
    if (len > 0) {
      const int ps = seg_pos(v0, (uint32_t)len, sinknode);
      if (ps >= 0 && d0 + ps >= lo_sink) ts = ps;
      const int pt = seg_pos(v0, (uint32_t)len, reached);
      if (pt >= 0 && (d0 + pt == len0 || (n_len > 1 && d0 + pt == len1))) {
        tt = pt;
        if (t_is_s) ts = max(ts, pt);
        if (d0 + pt == len0) { sh[SH_START0] = b; sh[SH_ST0] = (uint32_t)pt; }  // (state (reached, len_j) is unique)
        else { sh[SH_START1] = b; sh[SH_ST1] = (uint32_t)pt; }
      }
    }
    const uint32_t ls = d0 <= lmf ? l_seed[d0] : G2S_DEV_INVALID;
    const bool source = ls != G2S_DEV_INVALID && (v0 >> 1) == (ls >> 1);  // :1270, k-mer comparison only
    s_aux[b] = s_gen[b];
    s_t[b] = enc15(ts) | (source ? 0x8000u : 0u) | (enc15(tt) << 16);
  }
  __syncthreads();
  if (wave == 0u) {
    // the sweep (wave 0), in reverse: a segment in the closure tells its parents — one addition per parent and
    // mark: children on paths to a sink in bits 20..22 of the parent's word, children in the traceback closure in
    // bits 24..26 (see fill_seg.hip).  A parent has a lower id than its children, so by the time a chunk of 64
    // segments is reached every child outside it has spoken; a pass over the chunk is final unless a segment that
    // spoke in it has a parent INSIDE the chunk — then another pass (a parent is mostly many rounds older than its
    // child: one pass for most chunks, where the sweep used to take one per generation, five or six a chunk).
    bool choice = false;
    uint32_t hi = nseg;
    while (hi > 0) {
      const uint32_t lo = hi > 64u ? hi - 64u : 0u;
      const uint32_t b = lo + (uint32_t)lane;
      const bool hb = b < hi;
      const uint32_t dl = hb ? s_dl[b] : 0u;
      const uint32_t p01 = hb ? s_p01[b] : 0xFFFFFFFFu, p23 = hb ? s_p23[b] : 0xFFFFFFFFu;
      const uint32_t pre = hb ? s_t[b] : 0x7FFF7FFFu;
      const int d0 = (int)(dl & 0xFFFFu);
      const int len = max(0, min((int)(dl >> 16), d_last - d0 + 1));
      const uint32_t q0 = p01 & 0xFFFFu, q1 = p01 >> 16, q2 = p23 & 0xFFFFu, q3 = p23 >> 16;
      const bool speaks = hb && d0 > 0 && !(pre & 0x8000u);
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
    if (lane == 0 && choice) sh[SH_CHOICE] = 1u;
  }
  __syncthreads();
  WPROF(10);
  const bool choice = sh[SH_CHOICE] != 0u;
  const uint32_t start_b0 = sh[SH_START0], start_b1 = sh[SH_START1], start_t0 = sh[SH_ST0], start_t1 = sh[SH_ST1];

  // sink position of a closure segment (-1: none): a sink state at or below ts
  auto sink_pos = [&](uint32_t v0, int d0, int len, int ts) -> int {
    int sp = -1;
    if (ts >= 0) {
      const int pk = seg_pos(v0, (uint32_t)len, sinknode);
      if (pk >= 0 && d0 + pk >= lo_sink) sp = pk;
      if (t_is_s) { const int pt = seg_pos(v0, (uint32_t)len, reached); if (pt >= 0 && (d0 + pt == len0 || (n_len > 1 && d0 + pt == len1))) sp = pt; }
      if (sp > ts) sp = -1;
    }
    return sp;
  };

  // ---------------- phase D2 for small closures without a repeated k-mer (:1314-1435), wave 0 -----------------
  // (see fill_seg.hip: the branch rule as a prefix sum over segments; anything else is analysed by the host)
  bool analysed = false, sink_safe = false;
  uint32_t sub_vertices = 0, sub_edges = 0;
  if (want_s && nseg <= 192u) {
    if (wave == 0u) {
      uint32_t n_s = 0, edges = 0, src_out = 0, sink_in = 0;
      bool dag = true;
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
        const int sp = sink_pos(v0, d0, len, ts);
        n_s += wave_sum(in_s ? (uint32_t)ts + 1u : 0u);
        edges += wave_sum(in_s ? (uint32_t)ts + (source ? 1u : (d0 > 0 ? npar : 0u)) + (sp >= 0 ? 1u : 0u) : 0u);
        src_out += (uint32_t)__popcll(__ballot(in_s && source));
        sink_in += (uint32_t)__popcll(__ballot(sp >= 0));
        const uint32_t idx = v0 >> 1;
        const uint32_t ilo = (v0 & 1u) ? idx - (uint32_t)max(ts, 0) : idx, ihi = (v0 & 1u) ? idx : idx + (uint32_t)max(ts, 0);
        for (uint32_t a0 = 0; a0 <= b0 && dag; a0 += 64u) {
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
          const int sp = sink_pos(v0, d0, len, ts);
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
        if (lane == 0) { sh[SH_D2F] = 1u | (sink_safe ? 2u : 0u); sh[SH_D2V] = n_s + 2u; sh[SH_D2E] = edges; }
      } else if (lane == 0) {
        sh[SH_D2F] = 0u;
      }
    }
    __syncthreads();
    analysed = (sh[SH_D2F] & 1u) != 0u;
    sink_safe = (sh[SH_D2F] & 2u) != 0u;
    if (analysed) { sub_vertices = sh[SH_D2V]; sub_edges = sh[SH_D2E]; }
  }
  // the all-paths recount: the sum of the counts of the sink states (:1189-1226)
  if (tid == 0) { sh[SH_CS_LO] = 0u; sh[SH_CS_HI] = 0u; sh[SH_NREC] = 0u; sh[SH_NSUB] = 0u; sh[SH_NXP] = 0u; }
  __syncthreads();
  if (want_s) {
    unsigned long long acc = 0;
    for (uint32_t b = tid; b < nseg; b += NT) {
      const int ts = dec15(s_t[b]);
      if (ts < 0) continue;
      const uint32_t v0 = s_node[b], dl = s_dl[b];
      const int d0 = (int)(dl & 0xFFFFu);
      const int len = max(0, min((int)(dl >> 16), d_last - d0 + 1));
      if (sink_pos(v0, d0, len, ts) >= 0) acc += s_cnt[b];
    }
    if (acc) atomicAdd((unsigned long long*)&sh[SH_CS_LO], acc);
  }
  WPROF(11);
  // ---- the closure leaves as SEGMENTS (32 bytes each, SegRec), children before parents = descending segment id.
  // s_aux becomes the segment's index among the emitted ones: per chunk of 64 a count, the counts summed from the top.
  uint32_t* ccnt = lds + 2u * CAP;  // [CAP / 64] counts, then bases
  const uint32_t nchunk = (nseg + 63u) / 64u;
  static_assert(G2S_SEGX_CAP / 64u <= 512u, "one pass of wave 0 per 64 chunks");
  {
    uint32_t a_sub = 0;
    for (uint32_t c = wave; c < nchunk; c += SEGW_NW) {
      const uint32_t b = c * 64u + (uint32_t)lane;
      const uint32_t st = b < nseg ? s_t[b] : 0x7FFF7FFFu;
      const int ts = dec15(st), tt = dec15(st >> 16);
      const bool in = max(ts, tt) >= 0;
      const uint64_t m = __ballot(in);
      if (lane == 0) ccnt[c] = (uint32_t)__popcll(m);
      a_sub += in ? (uint32_t)(max(ts, tt) + 1) : 0u;
    }
    a_sub = wave_sum(a_sub);
    if (lane == 0 && a_sub) atomicAdd(&sh[SH_NSUB], a_sub);
  }
  __syncthreads();
  if (wave == 0u) {  // bases from the top: base[c] = emitted segments of the chunks above c
    uint32_t run = 0;
    for (uint32_t top = nchunk; top > 0; top = top > 64u ? top - 64u : 0u) {
      const bool hb = (uint32_t)lane < top;
      const uint32_t c = hb ? top - 1u - (uint32_t)lane : 0u;
      const uint32_t v = hb ? ccnt[c] : 0u;
      const uint32_t incl = wave_scan(v, lane);
      if (hb) ccnt[c] = run + incl - v;
      run += rl(incl, 63);
    }
    if (lane == 0) sh[SH_NREC] = run;
  }
  __syncthreads();
  for (uint32_t c = wave; c < nchunk; c += SEGW_NW) {
    const uint32_t b = c * 64u + (uint32_t)lane;
    const uint32_t st = b < nseg ? s_t[b] : 0x7FFF7FFFu;
    const bool in = max(dec15(st), dec15(st >> 16)) >= 0;
    const uint64_t m = __ballot(in);
    // (descending ids: the lanes above this one come first)
    if (in) s_aux[b] = ccnt[c] + (uint32_t)__popcll(m >> lane >> 1);
  }
  __syncthreads();
  const uint32_t nrec = sh[SH_NREC], nsub = sh[SH_NSUB];
  const int count_s = (int)min(*(unsigned long long*)&sh[SH_CS_LO], (unsigned long long)G2S_DEV_MAX_PATHS);
  const uint32_t nres = 2u * nrec;  // in 16-byte units of the output buffer
  // (a closure the host will analyse goes to pinned memory as well, now: see SegArgs.early_*)
  const bool early = A.early_items != nullptr && want_s && !analysed && nrec > 0u;
  if (tid == 0) {
    const unsigned long long hb_ = atomicAdd(out_counter, (unsigned long long)nres);
    sh[SH_HBASE_LO] = (uint32_t)hb_; sh[SH_HBASE_HI] = (uint32_t)(hb_ >> 32);
    sh[SH_ESLOT] = 0xFFFFFFFFu;
    if (early) {
      const unsigned long long slot = atomicAdd(&A.early_ctr[0], 1ull);
      if (slot < (unsigned long long)A.early_cap_items) {
        const unsigned long long so = atomicAdd(&A.early_ctr[1], (unsigned long long)nrec);
        sh[SH_ESLOT] = (uint32_t)slot;
        sh[SH_EOFF] = so + nrec <= (unsigned long long)A.early_cap_segs ? (uint32_t)so : 0xFFFFFFFFu;  // (no room: the item says so)
      }
    }
  }
  __syncthreads();
  const uint32_t eslot = sh[SH_ESLOT], eoff = sh[SH_EOFF];
  SegRec* edst = (eslot != 0xFFFFFFFFu && eoff != 0xFFFFFFFFu) ? A.early_segs + eoff : nullptr;
  const unsigned long long hbase = (unsigned long long)sh[SH_HBASE_LO] | ((unsigned long long)sh[SH_HBASE_HI] << 32);
  if (hbase + nres > out_cap) {  // the output buffer is full: the gap runs again in the LDS tier
    if (tid == 0) go->flags = flags | G2S_DEV_OVERFLOW_B | G2S_DEV_WHY_LOG;
    publish();
    return;
  }
  {
    SegRec* dst = (SegRec*)(sub_out + hbase);
    uint32_t nxp = 0;
    for (uint32_t b = tid; b < nseg; b += NT) {
      const uint32_t st = s_t[b];
      const int ts = dec15(st), tt = dec15(st >> 16);
      if (max(ts, tt) < 0) continue;
      const uint32_t dl = s_dl[b];
      const int d0 = (int)(dl & 0xFFFFu);
      const uint32_t v0 = s_node[b];
      int split = ts;  // states t <= split carry safe bit a, the others b: a sink inside the S part
      if (analysed && ts >= 0) {
        const int len = max(0, min((int)(dl >> 16), d_last - d0 + 1));
        const int sp = sink_pos(v0, d0, len, ts);
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
        // several parents: in GATB's predecessor order (see fill_seg.hip)
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
#define SEGW_CSWAP(a, c) do { if (key[a] > key[c]) { const uint32_t tk = key[a], ti = id[a]; key[a] = key[c]; id[a] = id[c]; key[c] = tk; id[c] = ti; } } while (0)
        SEGW_CSWAP(0, 1); SEGW_CSWAP(2, 3); SEGW_CSWAP(0, 2); SEGW_CSWAP(1, 3); SEGW_CSWAP(1, 2);
#undef SEGW_CSWAP
        r.par01 = id[0] | (id[1] << 16);
        r.par23 = id[2] | (id[3] << 16);
        if (ordered) r.flags |= G2S_SEG_ORDERED;
        nxp += k > 1u ? k - 1u : 0u;
      }
      dst[s_aux[b]] = r;
      if (edst) edst[s_aux[b]] = r;
    }
    if (edst) __threadfence_system();  // (this thread's records are in host memory before the item says so)
    nxp = wave_sum(nxp);
    if (lane == 0 && nxp) atomicAdd(&sh[SH_NXP], nxp);
  }
  __syncthreads();
  if (tid == 0) {
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
    go->n_xp = sh[SH_NXP];
    go->n_xl = nrec;
    go->sub_off = hbase;
    go->x_sub = nsub;
    go->stat[6] = gen;
    go->stat[7] = (uint32_t)((__builtin_amdgcn_s_memtime() - cyc2) >> 8);
    if (to_d2) {  // (as in fill_seg.hip: the entry carries the list's tag)
      if (A.d2_tag) __threadfence();  // (a polling launch reads them while this kernel runs; a launch behind it needs no fence)
      const unsigned long long at = atomicAdd(out_counter + 4, 1ull);
      __hip_atomic_store(&A.d2_list[at], gi | A.d2_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (A.d2_ticks) A.d2_list[A.d2_ticks + gi] = (uint32_t)__builtin_amdgcn_s_memrealtime();  // (tools: when the closure was listed)
    }
    if (eslot != 0xFFFFFFFFu) {  // the item: the gap's record, then what says it is complete
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
#ifdef G2S_SEGW_PROFILE
  if (dbg && tid == 0) {
    uint32_t* o = dbg + (size_t)x * dbg_words;
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    o[dbg_words - 14u] = pa_rec; o[dbg_words - 13u] = pa_prop; o[dbg_words - 12u] = pa_bar; o[dbg_words - 11u] = pb_pack;
    o[dbg_words - 10u] = pb_sel; o[dbg_words - 9u] = pb_child;
    o[dbg_words - 8u] = (uint32_t)(pw_t[9] - pw_t[8]); o[dbg_words - 7u] = (uint32_t)(pw_t[10] - pw_t[9]);
    o[dbg_words - 6u] = (uint32_t)(pw_t[11] - pw_t[10]); o[dbg_words - 5u] = (uint32_t)(t_end - pw_t[11]);
    o[dbg_words - 4u] = pb_scan; o[dbg_words - 3u] = pb_b0; o[dbg_words - 2u] = pb_sel + pb_child; o[dbg_words - 1u] = pb_b3;
#if G2S_SEGW_PROFILE == 2
    for (int pi = 0; pi < 4; pi++) { o[dbg_words - 14u + pi] = pf_acc[pi]; o[dbg_words - 8u + pi] = pf_acc[4 + pi]; }
#endif
  }
#endif
  publish();
}

// The large variant: one workgroup of eight waves per compute unit (it takes all of the LDS), each working through
// the list by an atomic counter so that the longest searches (the list is sorted) start first.
__global__ __launch_bounds__(SEGW_NT) void g2s_fill_segw(const SegArgs A, uint32_t* scratch, uint32_t ngaps_max,
                                                          unsigned long long* next_gap, const unsigned long long* ngaps_dev) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t ngaps = ngaps_dev ? min((uint32_t)*ngaps_dev, ngaps_max) : ngaps_max;
  uint32_t* scr = scratch + (size_t)blockIdx.x * SEGW_SCR_WORDS;
  uint32_t* sh = lds + (SEGW_LDS_WORDS - SH_WORDS);
  while (true) {
    __syncthreads();  // (the previous gap's last reads of the shared words)
    if (threadIdx.x == 0) sh[SH_X] = (uint32_t)atomicAdd(next_gap, 1ull);
    __syncthreads();
    const uint32_t x = sh[SH_X];
    __syncthreads();  // (segw_fill_one resets the shared words)
    if (x >= ngaps) break;
    segw_fill_one(lds, A, x, scr);
  }
}

namespace g2s {

size_t fill_segw_lds_bytes() { return 4u * SEGW_LDS_WORDS; }
size_t fill_segw_scratch_bytes(uint32_t workgroups) { return (size_t)workgroups * SEGW_SCR_WORDS * 4u; }

hipError_t launch_fill_segw(hipStream_t st, uint32_t ngaps, uint32_t workgroups, const uint32_t* succ, const uint32_t* urec,
                            const GapDev* gaps, const uint32_t* gap_ids, const uint32_t* flank_nodes, SubRec* sub_out,
                            unsigned long long out_cap, unsigned long long* out_counter, GapOut* outs, GapOut* outs_host,
                            uint32_t* done_list, int skip_confident, uint32_t* dbg, uint32_t* scratch,
                            unsigned long long* next_gap, bool resident, const unsigned long long* ngaps_dev, const SegEarly* early,
                            uint32_t* d2_list, uint32_t d2_tag, const GapLite* lite, int lite_e, int lite_all_paths) {
  if (ngaps == 0) return hipSuccess;
  const size_t bytes = fill_segw_lds_bytes();
  hipError_t e = hipFuncSetAttribute((const void*)g2s_fill_segw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  SegArgs A = {succ, urec, GapSrc{gaps, resident ? lite : nullptr, lite_e, lite_all_paths}, gap_ids, flank_nodes, sub_out, out_cap, out_counter, outs, outs_host, done_list,
               skip_confident, dbg, fill_segx_dbg_words(), nullptr, nullptr, 0u, 1u, resident ? 1u : 0u,
               (resident && d2_list) ? d2_ticks_offset : 0u, nullptr, early ? early->segs : nullptr, early ? early->items : nullptr, early ? early->outs : nullptr,
               early ? early->ctr : nullptr, early ? early->cap_items : 0u, early ? early->cap_segs : 0u, resident ? d2_list : nullptr, resident ? d2_tag : 0u};
  hipLaunchKernelGGL(g2s_fill_segw, dim3(workgroups), dim3(SEGW_NT), bytes, st, A, scratch, ngaps, next_gap, ngaps_dev);
  return hipGetLastError();
}

}  // namespace g2s
