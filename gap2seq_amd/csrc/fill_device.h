// gap2seq_amd/csrc/fill_device.h — structures shared by the HIP kernels and the
// host orchestration of the fill path (layout of the per-gap work areas in HBM).
//
// The path is /root/reference/src/Gap2Seq.cpp:858-1167 (fill_gap phases A-C);
// the per-gap quantities below are the locals of that function.
#pragma once
#include <stdint.h>

#define G2S_DEV_INVALID 0xFFFFFFFFu
#define G2S_DEV_EMPTY64 0xFFFFFFFFFFFFFFFFull
#define G2S_DEV_MAX_PATHS 1073741822u /* Gap2Seq.cpp:38 */

/* GapOut.flags */
#define G2S_DEV_Q7_A 0x1u        /* both strands of a k-mer in the right set       */
#define G2S_DEV_Q7_B 0x2u        /* both strands of a k-mer at one DP level        */
#define G2S_DEV_OVERFLOW_A 0x4u  /* right-set tables too small for this gap        */
#define G2S_DEV_OVERFLOW_B 0x8u  /* state tables too small for this gap            */
#define G2S_DEV_Q7_D 0x10u       /* both strands of a k-mer at one level of the backward sweep (HBM tier; the LDS
                                    tier's closure is a subset of the DP states, covered by Q7_B) */
/* why a gap left the LDS tier (diagnostics, set together with OVERFLOW_B) */
#define G2S_DEV_WHY_FRONTIER 0x100u /* a DP level wider than the LDS frontier buffers */
#define G2S_DEV_WHY_HITS 0x200u     /* more target hits than the LDS list holds       */
#define G2S_DEV_WHY_LOG 0x400u      /* state log full                                 */
#define G2S_DEV_WHY_RS 0x800u       /* (with OVERFLOW_A) the right-set table was full */
/* diagnostics of the LDS tier's pools (no effect on results) */
#define G2S_DEV_LOG_POOL 0x1000u    /* the state log moved to a chunk of the log pool        */
#define G2S_DEV_RS_POOL 0x2000u     /* the right set moved from LDS to a chunk of the spill pool */
/* the closure of this gap was emitted as segments (SegRec, segment tier), not as per-state records */
#define G2S_DEV_COMPACT 0x4000u
/* (with an OVERFLOW bit) a probe loop of the segment tier's large variant ran past its bound: a defect, never
   expected; the gap runs again in the LDS tier instead of holding the GPU */
#define G2S_DEV_WATCHDOG 0x8000u
/* the gap ran in the large variant of the segment tier (fill_segw.hip) */
#define G2S_DEV_BIG 0x10000u
/* LDS tier, lvl[]: bit 31 of the END offset of level L = L was produced by a bulk step
 * (same width as level L-1, state r has the single parent r of level L-1) */
#define G2S_LVL_UNIFORM 0x80000000u

struct GapDev {
  int32_t g;           // gap_len
  int32_t e;           // gap_err
  int32_t lmf, rmf;    // left/right_max_fuz
  int32_t D;           // left_half + right_half = lmf + rmf + g + e   (:862-863,1029)
  int32_t right_half;  // rmf + ceil((g+e)/2)                          (:862)
  int32_t prune_from;  // g/2 + e/2 + lmf                              (:1050)
  int32_t all_paths;
  uint32_t flank_off;  // flank_nodes: [left seeds lmf+1][right seeds rmf+1][targets rmf+1]
  uint32_t rs_mask;    // right-set hash: capacity-1 (u32 slots)
  uint32_t rlog_cap;   // right BFS visit log capacity
  uint32_t st_mask;    // state hash: capacity-1
  uint32_t slog_cap;   // state log capacity (<= (st_mask+1)/2)
  uint32_t pad0;       // LDS tier: capacity of the gap's extra-parent list
  uint64_t rs_off;     // element offsets into the session arrays
  uint64_t rlog_off;
  uint64_t st_off;     // LDS tier: offset of the gap's extra-parent list
  uint64_t slog_off;   // also: offset of the gap's state log / closure scratch in the LDS tier
  uint64_t lvl_off;    // LDS tier: D+2 level offsets into the state log
};

/* Resident mode, segment tier: the 32 bytes of a gap that are not the same for the whole list and cannot be derived —
 * what the host writes per gap and the link carries (GapDev: 96 bytes, most of them the work-area offsets of the tiers
 * that keep their search state in device memory).  The kernels expand it (GapSrc::load). */
struct GapLite {
  int32_t g;            // gap_len
  uint16_t lmf, rmf;
  uint32_t flank_off;   // as GapDev.flank_off
  uint32_t text_off;    // -> GapDev.rs_mask: the gap's flank text (look-ups in the fill kernel)
  uint64_t arena_off;   // -> GapDev.rlog_off: the gap's fill buffer in the batch's share of the arena
  uint32_t has_skip;    // -> GapDev.rlog_cap: a skip rule decides over the gap
  uint32_t pad;
};
#ifdef __HIPCC__
/* where a kernel of the segment tier reads a gap's descriptor from: the full records, or the short ones + the list's constants */
struct GapSrc {
  const GapDev* full;
  const GapLite* lite;
  int32_t e, all_paths;
  __device__ __forceinline__ GapDev load(uint32_t gi) const {
    if (!lite) return full[gi];
    const GapLite l = lite[gi];
    GapDev d;
    d.g = l.g; d.e = e; d.lmf = (int32_t)l.lmf; d.rmf = (int32_t)l.rmf;
    d.D = d.lmf + d.rmf + d.g + e;                 // :862-863,1029
    d.right_half = d.rmf + (d.g + e + 1) / 2;      // :862
    d.prune_from = d.g / 2 + e / 2 + d.lmf;        // :1050
    d.all_paths = all_paths;
    d.flank_off = l.flank_off; d.rs_mask = l.text_off; d.rlog_cap = l.has_skip; d.st_mask = 0u; d.slog_cap = 0u; d.pad0 = 0u;
    d.rs_off = 0ull; d.rlog_off = l.arena_off; d.st_off = 0ull; d.slog_off = 0ull; d.lvl_off = 0ull;
    return d;
  }
};
#endif

/* SubState.flags */
#define G2S_SUB_IN_S 0x1u     /* on a path to a sink of the reference's subgraph (D1/D2)      */
#define G2S_SUB_IN_T 0x2u     /* reachable backwards from a traceback start (D3)              */
#define G2S_SUB_SOURCE 0x4u   /* depth <= lmf and k-mer == left flank k-mer at that offset    */
#define G2S_SUB_SINK 0x8u     /* has an edge to the sink pseudo-vertex                        */
#define G2S_SUB_START_T 0x10u /* (reachedTarget, pathLengths[i])                              */
/* SegRec.flags only: the parents are listed in GATB's predecessor order (sorted on the device) */
#define G2S_SEG_ORDERED 0x100u

/* One state of the backward closure that phase D works on, in discovery order (depth
 * descending), as the HBM tier's kernel emits it: pred[i] = index (within the gap's array) of
 * the state of graph.predecessors(node)[i] at depth-1 when that state is set, else -1.  The
 * host converts these to SubRec. */
struct SubState {
  uint32_t node;
  uint32_t depth;
  uint32_t cnt;
  uint32_t flags;
  int32_t pred[4];
};

/* The closure as the host works on it and as the LDS tier emits it: 16 bytes per state.
 * Almost every state has one parent; the others set G2S_SUB_MORE in `pred` and list their
 * further parents in the gap's side list xp[] = (state << 32 | parent), sorted by state.
 * Parents are a set here; where their GATB order matters (the traceback, :1476-1513) it
 * is recovered from the graph: predecessors(v)[i] is the parent p whose p^1 ends with base i. */
#define G2S_SUB_META_FLAG_SHIFT 27u         /* meta = depth | flags << 27 */
#define G2S_SUB_META_DEPTH_MASK 0x07FFFFFFu
#define G2S_SUB_MORE 0x40000000             /* in pred: further parents in xp[] */
#define G2S_SUB_PRED_MASK 0x3FFFFFFF
struct SubRec {
  uint32_t node;
  uint32_t cnt;
  uint32_t meta;
  int32_t pred;  /* -1: none */
};

/* The closure as the segment tier emits it: one record per unitig segment that holds closure
 * states, children before parents.  The states of a segment are (node +- 2t, depth + t) for
 * t = 0 .. len-1 (even node ids walk up, odd down), all with the same path count; t <= ts are on a
 * path to a sink, t <= tt reachable backwards from a traceback start (0xFFFF = none); the parents
 * of state 0 are the LAST states of the parent segments (indices into this array, 0xFFFF = none).
 * post.cpp: seg_expand turns this into SubRec + side list for the host half of phase D. */
struct SegRec {
  uint32_t node;
  uint32_t depth_len;  /* depth | len << 16 */
  uint32_t cnt;
  uint32_t ts_tt;      /* ts (bits 0-14) | safe_a << 15 | tt << 16 (bits 16-30) | safe_b << 31; 0x7FFF = none */
  uint32_t par01, par23;
  uint32_t flags;      /* G2S_SUB_SOURCE: state 0 is a left-flank k-mer at its offset (not expanded) */
  uint32_t pad;        /* split: states t <= split carry safe_a, the others safe_b (device analysis only) */
};

struct GapOut {
  uint32_t flags;
  uint32_t n_right;    // visited oriented nodes in the right set
  uint32_t x_right;    // expansions done by the right BFS
  uint32_t n_states;   // states set by the left DP (= S_B)
  uint32_t x_left;     // expansions done by the left DP (= X_B)
  int32_t final_d;     // currentD when the DP loop ended (D+1, or the -best-only break level)
  int32_t c_count;     // phase C count (:1131-1150)
  int32_t n_len;       // pathLengths.size()
  int32_t len[2];
  int32_t reached_j;   // reachedFuz
  uint32_t n_sub;      // states in the backward closure (phase D input)
  uint64_t sub_off;    // offset of this gap's closure in the packed output (SubRec / SubState units)
  uint32_t x_sub;      // expansions done by the backward sweep
  uint32_t n_xl;       // LDS tier: entries in the gap's extra-parent list; segment tier: SegRec records emitted
  uint32_t top_level;  // LDS tier: last DP level that holds a state
  uint32_t n_xp;       // LDS tier: entries of the closure's side list (parents beyond the first)
  // LDS tier statistics: per-level iterations / bulk iterations of phases A, B, D1 and
  // shader cycles (in units of 256) spent in A, B+C, D1
  uint32_t stat[8];
  // segment tier: phase D2 and the stop-depth analysis done on the device (valid with G2S_DEVA_ANALYSED)
  int32_t fixed_draws[2];  // rand() draws a traceback from start j consumes, or -1 when that depends on the draws
  uint32_t start_seg;      // emitted segment of traceback start 0 | start 1 << 16 (0xFFFF none)
  uint32_t start_t;        // its position inside the segment, same packing
  uint32_t sub_vertices, sub_edges;  // SubgraphStats of a closure without a repeated k-mer (nothing contracted)
  int32_t count_s;         // all-paths recount: sum of the counts of the sink states (:1189-1226)
  uint32_t dflags;
  // stop depths behind traceback start j (:1455-1462): lowest | highest << 16 of the depths at which a traceback
  // from it can meet its left-flank k-mer; equal = fixed_draws[j] is known (d3_device.hip prices the others)
  uint32_t stop[2];
};
#define G2S_DEVA_ANALYSED 0x1u   /* D2 (branch rule) done on the device: safe bits in SegRec.ts_tt, split in SegRec.pad */
#define G2S_DEVA_CHOICE 0x2u     /* some entry of the traceback closure has more than one parent */
#define G2S_DEVA_SINK_SAFE 0x4u  /* branch[sink] == 1 (Q5: what k-mers outside the subgraph read) */
#define G2S_DEVA_D2_PENDING 0x10u /* the fill kernel listed the gap for g2s_d2_* (which runs beside phase D3's first kernels: the
                                    hand-off looks whether G2S_DEVA_RUNS has joined it; if not, the closure is the host's) */
#define G2S_DEVA_D2_FAILED 0x20u  /* g2s_d2_* could not analyse the closure (beyond its capacities): the host's after all */
#define G2S_DEVA_TRACED 0x40u    /* the fill kernel's wave traced the gap itself (one path, nothing to draw for): fill text and result
                                    record are written, GapOut.top_level = the fill's length; phase D3 counts its draws, the trace kernel skips it */
#define G2S_DEVA_SPEC 0x80u      /* the fill kernel's wave wrote a GUESS of the gap's traceback (first path length, first parent at every
                                    choice): text and record in the caller's buffers and in device memory; GapOut.top_level = where that
                                    text begins | ends << 16; the trace kernel traces the gap and sends through the link what differs */
#define G2S_DEVA_RUNS 0x8u       /* (with ANALYSED) analysed by g2s_d2_* (d2_device.hip): the verdicts are the gap's runs (D2Out),
                                    the subgraph statistics D2Out.sub */
