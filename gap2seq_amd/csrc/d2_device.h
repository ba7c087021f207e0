// gap2seq_amd/csrc/d2_device.h — phase D2 of fill_gap on the device for the closures the fill kernels do not analyse
// themselves (d2_device.hip): strong components, contraction and the safe-vertex rule
// (/root/reference/src/Gap2Seq.cpp:1314-1435) for closures of more than 192 segments and for closures in which a
// k-mer occurs at several depths.  Runs between the fill kernels and phase D3 (d3_device.hip) in resident mode.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "fill_device.h"

namespace g2s {

// What the kernel leaves per analysed gap (GapOut.dflags gains G2S_DEVA_ANALYSED | G2S_DEVA_RUNS, and
// G2S_DEVA_SINK_SAFE when branch[sink] == 1): the subgraph statistics (Gap2Seq.hpp:50-58) and where the gap's runs
// lie.  A run is a stretch of consecutive k-mer indices whose vertices share the branch rule's verdict: two words
// {first index, last index | safe << 31}, sorted by first index, disjoint; a k-mer in no run reads branch[sink] (Q5).
struct D2Out {
  uint32_t run_off, n_runs;
  uint32_t sub[6];  // vertices, edges, nontrivial components, their size, vertices_final, edges_final
};
static_assert(sizeof(D2Out) == 32, "D2Out layout");

// capacities of the two instantiations (closure segments on paths to a sink, closure segments in all, cut points,
// nodes of the run graph, its edges): a gap that does not fit the small one is passed on to the large one, a gap that
// does not fit that stays the host's (post.cpp)
#define G2S_D2_SMALL_NS 256u
#define G2S_D2_SMALL_NREC 512u
#define G2S_D2_SMALL_BP 1024u
#define G2S_D2_SMALL_NV 512u
#define G2S_D2_SMALL_E 1024u
#ifndef G2S_D2_BIG_NT
#define G2S_D2_BIG_NT 1024u  // threads of a workgroup of the large instantiation
#endif
#define G2S_D2_BIG_NS 4096u
#define G2S_D2_BIG_NREC 16384u
#define G2S_D2_BIG_BP 16384u
#define G2S_D2_BIG_NV 6144u
#define G2S_D2_BIG_E 12288u

struct D2Args {
  const GapDev* gaps;
  const GapLite* lite;                  // not null: the descriptors as short records (fill_device.h: GapSrc), `gaps` is not read
  int32_t lite_e;
  const uint32_t* flank_nodes;
  GapOut* outs;
  SubRec* sub;                          // the closures (SegRec), where the fill kernels left them
  uint32_t* list;                       // the gaps to analyse (| tag, when the fill kernels tag their entries) ...
  const unsigned long long* count;      // ... how many (device memory: the fill kernels count them)
  unsigned long long* next;             // work counter of this launch (zero before it)
  uint32_t* list_next;                  // small instantiation: the gaps it passes on, counted in *count_next
  unsigned long long* count_next;
  D2Out* d2out;                         // by gap
  uint32_t* runs;                       // two words per run
  unsigned long long* run_cursor;       // runs handed out (zero before the first launch)
  unsigned long long run_cap;
  uint32_t* scratch;                    // d2_scratch_words() words per workgroup
  int32_t all_paths;
  uint32_t list_cap;                    // most gaps the list can hold (the grid is sized by it)
  uint32_t pass_all;                    // (tests) bit 0: the small instantiation passes every gap on to the large one; bit 1: no chains are contracted
  unsigned long long* prof;             // (tools, may be null) 2 x 32 counters (small, large): ticks per section of the analysis, summed over gaps; counts
  unsigned long long* log;              // (tools, may be null) 16 words per closure taken: see tools/d2_log.py; the first half is the small instantiation's
  uint32_t log_cap;
  uint32_t behind;                      // the gaps' results are read only behind the launch (an event): no release per closure
  unsigned long long* wgs_done;         // workgroups of g2s_d2_* that are through (the trace kernel's last wave waits for all of them)
  // entries carry this tag in bits 24-31 (0: plain gap numbers); poll: the launch runs BESIDE the fill kernel and takes
  // entries as they appear, until the 64 counters at `done` add up to `expected` (every gap of the fill launch is
  // through) and the list is empty
  uint32_t tag, poll;
  const unsigned long long* done;
  unsigned long long expected;
};

size_t d2_scratch_bytes(bool big, uint32_t workgroups);
// both instantiations behind each other on the stream: small (many workgroups a compute unit), then large (one)
hipError_t launch_d2(hipStream_t st, const D2Args& A, uint32_t small_wgs, uint32_t big_wgs, uint32_t* scratch_small,
                     uint32_t* scratch_big, uint32_t* list_big, unsigned long long* count_big, unsigned long long* next_big);
// the small instantiation beside the fill kernel (A.poll, A.done, A.expected, A.tag set): a few workgroups that take
// the closures as their gaps end; launch_d2 follows behind the fill kernels for what is left
hipError_t launch_d2_poll(hipStream_t st, const D2Args& A, uint32_t wgs, uint32_t* scratch_small, uint32_t* list_big,
                          unsigned long long* count_big);

}  // namespace g2s
