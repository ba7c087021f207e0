// gap2seq_amd/csrc/d3_device.h — phase D3 of fill_gap on the device (d3_device.hip): the rand() stream,
// the stream offsets of every gap of a list and the tracebacks (/root/reference/src/Gap2Seq.cpp:178,
// 1437-1522), for lists whose gaps all finished in the segment tier with phase D2 done by the kernel.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "d2_device.h"
#include "fill_device.h"

#define G2S_D3_BLOCK_VARS 16u            /* draw-dependent gaps per block of the offset chain */
#define G2S_RAND_BLOCK 4096u             /* values one wave of g2s_rand_fill generates (64 per lane) */
#define G2S_RAND_WINDOW 160u             /* words of the stream the host hands over (31 of state + what the jumps read) */
#define G2S_D3_TABLE_BUDGET (8u << 20)   /* entries of the draw-count tables a list may need */

/* D3Summary.status */
#define G2S_D3_UNHANDLED 0x1u  /* a gap overflowed the segment tier, or its closure was left to the host's analysis */
#define G2S_D3_BUDGET 0x2u     /* the draw-count tables would not fit */
#define G2S_D3_ANOMALY 0x4u    /* a walk met something the host path has to look at (never expected) */

namespace g2s {

// one gap of the list, host -> device
struct D3Gap {
  uint64_t arena_off;  // of the gap's fill buffer in its list's share of the arena (D3Params.arena_base in front)
  int32_t skip_thr;    // skip_if_prev_right_fuz_gt (-1: never skipped)
  uint16_t lmf;
  uint8_t kind;        // 0 launched, 1 bad flank
  uint8_t pad;
};

// A gap whose closure the device did not analyse (a k-mer at two depths, or more segments than the kernel's
// pairwise check takes: post.cpp analyses those on segments or runs): its draws are counted here like everybody's —
// the number of draws does not depend on the safe/unsafe verdicts — and the trace kernel hands the host what it
// needs to finish the gap: the gap's record, its closure segments and the rand() values of its traceback.
struct D3HostItem {
  uint32_t gap, n_segs;
  uint64_t seg_off;   // into D3Side.segs
  uint64_t rnd_off;   // into D3Side.rnd: the raw words of the gap's draws
  uint32_t draws;     // it will consume
  uint32_t pad;       // D3Side.items: 1 once everything of the item is in host memory (the host polls it, and zeroes it)
};
struct D3Side {  // pinned host memory, written by g2s_d3_trace
  D3HostItem* items = nullptr;
  GapOut* outs = nullptr;  // by item
  SegRec* segs = nullptr;
  uint32_t* rnd = nullptr;
  uint64_t cap_items = 0, cap_segs = 0, cap_rnd = 0;
  unsigned long long* count = nullptr;  // items to expect (bit 63: something did not fit; ~0: not known yet), written behind the offsets' chain
};

// What g2s_d3_tables needs of the v-th draw-dependent gap, and what g2s_d3_trace needs of gap i besides its GapOut
// record: one record each, written by the kernels in front (a wave that gathers them from six arrays spends
// its time in a chain of memory round trips — on a 500-gap list those chains were the kernels).
struct alignas(16) D3Var {
  uint32_t gap, R, toff, tile0;   // the gap; deviations in front of it; where its table starts; tiles in front of its
  uint32_t base, dmin, dspread, ns;  // first draw at deviation 0; fewest draws; most - fewest; closure segments
  uint64_t sub_at;                // its closure, in 16-byte units from the list's first closure record
  uint32_t start_seg, start_t;
  int32_t len0, len1, n_len;
  uint32_t pad;
};
struct alignas(16) D3Trace {
  uint64_t arena_off;  // of the gap's fill buffer in the arena
  uint64_t sub_at;     // its closure (as D3Var.sub_at)
  uint32_t gi;         // ginfo
  uint32_t off, want;  // first draw, draws
  uint16_t lmf, nsegs;
};
static_assert(sizeof(D3Var) == 64 && sizeof(D3Trace) == 32, "record layouts");

// what the host reads after the last kernel (g2s_d3_trace's last wave copies it into pinned memory)
struct D3Summary {
  uint32_t status, unhandled, n_var, anomalies;
  unsigned long long host_items, host_segs, host_rnd;  // cursors of D3Side
  uint32_t handoff_waves, tiles;  // tiles: of 256 table entries, all gaps
  uint64_t table_entries, block_entries;
  uint64_t draws_min, draws_spread, draws_total;
  uint64_t xA, sA, xB, sB, xD, sD, segs, fill_bytes;
  uint32_t seg_gaps, filled;
  uint32_t rand_state[31];  // the 31 words in front of the first value the list did not consume
  uint32_t trace_waves;     // waves of g2s_d3_trace that are through
  uint32_t big_gaps;        // gaps that ran in the large variant of the segment tier (G2S_DEV_BIG)
  uint32_t traced_gaps;     // gaps whose fill kernel's wave wrote text and record itself (G2S_DEVA_TRACED)
  uint32_t spec_gaps;       // gaps whose fill kernel's wave wrote a guess (G2S_DEVA_SPEC); how much of them the trace kernel sent
                            // again: word 1 of each of the 64 fill-byte counters' lines (groups of 64 bases compared | sent << 32)
};
static_assert(sizeof(D3Summary) <= 512, "the lap stamps live at byte 512 of the summary's slot");

// the jump tables of the generator (seed independent): x^(2^20 a), x^(4096 b), x^(64 l) modulo the
// recurrence's polynomial, 31 coefficients each
struct RandTables {
  const uint32_t* hi = nullptr;   // [128][31]
  const uint32_t* mid = nullptr;  // [256][31]
  const uint32_t* lane = nullptr; // [64][31]
};
void rand_tables_host(uint32_t* hi /* 128*31 */, uint32_t* mid /* 256*31 */, uint32_t* lane /* 64*31 */);

// work areas of one list (all device memory, sized by d3_work_bytes and carved by d3_work_carve)
struct D3Work {
  uint32_t* ginfo = nullptr;    // [n] class | filled << 2 | skipped << 3 | right_fuz << 8
  uint32_t* dmin = nullptr;     // [n] draws of the gap (fewest, when they depend on the draws)
  uint32_t* dspread = nullptr;  // [n] most - fewest
  uint32_t* base = nullptr;     // [n] sum of dmin over the gaps in front
  uint32_t* vrank = nullptr;    // [n] draw-dependent gaps in front
  uint32_t* var_gap = nullptr;  // [n] the draw-dependent gaps in list order
  uint32_t* var_R = nullptr;    // [n + 1] sum of dspread over the draw-dependent gaps in front
  uint32_t* var_toff = nullptr; // [n + 1] where the gap's table starts
  uint32_t* var_tile = nullptr; // [n + 1] tiles of 256 table entries in front of the gap's
  uint32_t* blk_toff = nullptr; // [n / BLOCK_VARS + 2]
  uint32_t* blk_in = nullptr;   // [n / BLOCK_VARS + 2] deviation in front of the block
  uint32_t* dvar = nullptr;     // [n + 1] deviation in front of the i-th draw-dependent gap (last: of the whole list)
  int32_t* skip = nullptr;      // [n] skip_if_prev_right_fuz_gt (-1: never skipped)
  uint32_t* tile_var = nullptr; // [n + G2S_D3_TABLE_BUDGET / 256 + 2] the draw-dependent gap a tile belongs to
  D3Var* vdesc = nullptr;       // [n + 1]
  D3Trace* tdesc = nullptr;     // [n]
  uint32_t* host_slot = nullptr;  // [n] host-finished gaps: the gap's item of D3Side (~0: it did not fit)
  D3HostItem* hitems = nullptr;   // [n] by item: where the gap's wave of the trace kernel puts what the host needs
  uint16_t* tab = nullptr;      // [G2S_D3_TABLE_BUDGET] draws - dmin by (gap, deviation)
  uint32_t* btab = nullptr;     // [G2S_D3_TABLE_BUDGET / 4] deviation behind a block by deviation in front of it
  D3Summary* sum = nullptr;
  unsigned long long* fill_bytes = nullptr;  // 64 counters, 16 words apart, right behind the summary's 1024 bytes
  // (not carved: the caller's) 32 words that outlive the list's summary: the generator's state behind the list's last
  // draw, for the next list's stream when that list is queued before this one has ended (g2s_fill_begin); may be null
  uint32_t* link = nullptr;
  // (not carved: the caller's) what g2s_d2_* left for the gaps it analysed (GapOut.dflags & G2S_DEVA_RUNS): per-gap
  // statistics and runs (d2_device.h); hops: where the trace kernel lists the segments a traceback enters when the
  // closure is too large for its LDS — an entry per closure segment, at the closure's own offset.  All null: no gap
  // carries G2S_DEVA_RUNS and every closure the trace kernel meets fits its LDS.
  const D2Out* d2out = nullptr;
  const uint32_t* d2runs = nullptr;
  uint64_t* hops = nullptr;  // (depth at which the hop is entered | (segment | entry state << 16) << 32)
  // (not carved: the caller's) where the fill kernel's waves left their GUESSES of tracebacks that have choices
  // (G2S_DEVA_SPEC: text at the arena's offsets, records as g2s_result): the trace kernel compares and sends through the
  // link what differs.  Null: every traced gap is written in full.
  const char* spec_text = nullptr;
  const uint32_t* spec_res = nullptr;
};
size_t d3_work_bytes(uint32_t n);
void d3_work_carve(void* p, uint32_t n, D3Work* w);

struct D3Params {
  int32_t k, skip_confident, all_paths, unique_paths;
  uint64_t max_states;
  uint64_t arena_base; // where the list's share of the arena begins
  uint32_t n;          // gaps of the list
  uint32_t has_skip;   // some gap carries a skip rule
  uint32_t seg_cap;    // closure segments the trace kernel stages in LDS
  uint32_t map_cap;    // fill-buffer positions it can map there: the longest path of the list + 2
  uint32_t group_size; // gaps per group of the list (a list filled by several sessions: one region of closure records
  uint64_t sub_region; // per group, sub_region 16-byte units apart); one group: group_size >= n
  uint32_t laps;       // G2S_DEBUG: the trace kernel's waves record their laps (atomics on a few words)
  uint32_t self_clean; // the trace kernel zeroes the gaps' records, the summary and the counters behind itself
  // A list sharded over the sessions of a team, one group each (launch_d3_sharded_*): this group's place in the list's
  // one rand() stream.  base0: fewest draws of all groups in front; R0: their summed spreads (the deviations this
  // group's first draw-dependent gap can start with: 0 .. R0); d_in: the deviation it does start with.
  uint32_t base0, R0, d_in;
  // g2s_d2_* runs on a stream of its own beside this list's phase D3: the trace waves of the gaps it analyses wait for
  // their gap's verdict word, and the wave that cleans up behind the list waits until d2_wgs workgroups of it have
  // left (*d2_done counts them; null: nothing of the kind is running)
  const unsigned long long* d2_done;
  uint32_t d2_wgs, pad1;
};

// the stream: values [0, capacity) into rnd_all[31 ..] (sum_dev = nullptr; independent of the list's kernels, so
// it runs beside the fill kernel on a stream of its own), or only as far as the summary at sum_dev says the list
// can draw.  rnd_all[0 .. G2S_RAND_WINDOW) come from the host.
// first_value: values in front of its block of 4 096 are not generated (a group of a sharded list: nothing of the
// list draws them on this device)
hipError_t launch_rand_fill(hipStream_t st, uint32_t* rnd_all, const RandTables& rt, const D3Summary* sum_dev, uint64_t capacity,
                            uint64_t first_value = 0);

// all of phase D3 behind the fill kernel, on its stream, without a host round trip:
//   g2s_d3_classify / g2s_d3_scan   classes, draw counts, the skip rule (:369), prefix sums, table layout
//   (g2s_rand_fill has filled rnd_all by then: launch_rand_fill, on another stream)
//   g2s_d3_tables  draws of every draw-dependent gap for every offset it can start at
//   g2s_d3_blocks / g2s_d3_chain  the chain of deviations through those tables, block-wise
//   g2s_d3_handoff every gap's first draw and draw count; what the host needs to finish the gaps whose closure it
//                  analyses, into pinned memory (*side.count says when: the host polls it)
//   g2s_d3_trace   one wave per gap: the traceback, fill text and result record; the last wave copies the summary
// The same in three steps, for a list whose groups stay on the GPUs that filled them (one session per GPU, every one
// tracing its own gaps and writing its own results through its own link).  Between the steps the host exchanges a
// few words per group:
//   classes  classify + scan with offsets 0: the group's totals (D3Summary.draws_min, .draws_spread, .status)
//            -> host: base0, R0 of every group = prefix sums over the groups in front
//   tables   scan again with (P.base0, P.R0), tables, blocks, then the GROUP FUNCTION: the deviation behind the
//            group for every deviation 0 .. R0 in front of it, into group_fn[] (pinned)
//            -> host: d_in of group q+1 = group_fn_q[d_in of group q], d_in of group 0 = 0
//   trace    chain from P.d_in, hand-off, trace kernel (results and text of this group's gaps)
// rnd_all holds the list's stream from its first value (every session generates what its group can reach).
hipError_t launch_d3_sharded_classes(hipStream_t st, const D3Params& P, const D3Work& W, const GapOut* outs, const D3Gap* dgaps,
                                     bool summary_is_clean);
hipError_t launch_d3_sharded_tables(hipStream_t st, const D3Params& P, const D3Work& W, const GapOut* outs, const SubRec* sub,
                                    uint32_t* rnd_all, uint64_t rnd_capacity, uint32_t* group_fn /* [P.R0 + 1], device-writable */);
hipError_t launch_d3_sharded_trace(hipStream_t st, const D3Params& P, const D3Work& W, const GapOut* outs, const SubRec* sub,
                                   const char* lastch_up, const char* lastch_dn, uint32_t* rnd_all, uint64_t rnd_capacity,
                                   void* results, char* arena, const D3Side& side, void* summary_host,
                                   hipEvent_t ev_d2 = nullptr /* as launch_d3 */);

// the first G2S_RAND_WINDOW words of a stream from the 31 words of state another list's kernels left (D3Work.link)
hipError_t launch_rand_window(hipStream_t st, uint32_t* rnd_all, const uint32_t* link);

hipError_t launch_d3(hipStream_t st, const D3Params& P, const D3Work& W, const GapDev* gaps, const GapOut* outs,
                     const D3Gap* dgaps, const SubRec* sub, const char* lastch_up, const char* lastch_dn,
                     const RandTables& rt, uint32_t* rnd_all /* [31 + capacity]: first G2S_RAND_WINDOW words set */,
                     uint64_t rnd_capacity, void* results /* g2s_result[n], device-writable */,
                     char* arena /* device-writable */, const D3Side& side /* *side.count: ~0 until the hand-over is complete */,
                     void* summary_host /* device-visible pinned memory: D3Summary in 1024 bytes, then the 64 fill-byte counters */,
                     bool summary_is_clean /* the summary and the counters are zero already */,
                     uint32_t* clean_words /* with P.self_clean: eight words the last wave zeroes (the fill kernel's cursors); may be null */,
                     hipEvent_t ev_chain = nullptr /* recorded behind the kernel that knows the list's draws (W.link is written) */,
                     hipEvent_t ev_d2 = nullptr /* behind g2s_d2_* on its own stream: the hand-off waits for it (null: nothing to wait for) */,
                     hipEvent_t ev_stop = nullptr /* the trace kernel's own stop time (hipExtLaunchKernelGGL): a timed list's last kernel */);

}  // namespace g2s
