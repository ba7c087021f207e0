// gap2seq_amd/csrc/fill_seg.h — launcher of the segment tier (fill_seg.hip): phases A-D1 of
// /root/reference/src/Gap2Seq.cpp:858-1312 as a search over unitig segments.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "fill_device.h"

#define G2S_SEG_CAP 512u  /* segments per gap kept in LDS */
#define G2S_SEG_ASETS 4   /* right-set entries per gap: 64 per set, in registers */
/* the large variant (g2s_fill_segx): gaps that outgrow the capacities above */
#define G2S_SEGX_CAP 16384u  /* segments per gap, in the workgroup's global scratch */
#define G2S_SEGX_EA 6144u    /* right-set entries */
#define G2S_SEGX_AS 16384u   /* slots of the label table of phase A (LDS) */
#define G2S_SEGX_QCAP 16384u /* entries one round of phase A may queue */
#define G2S_SEGX_PE 1024u    /* pending events (LDS slots) */
#define G2S_SEGX_HS 4096u    /* slots of the event table */

#include "flank_lookup.h"

namespace g2s {

// resident mode, deep lists: where g2s_fill_segw puts the closures the host will analyse the moment their gaps end
// (SegArgs.early_*: device-visible pinned memory, the counters in device memory)
struct SegEarly {
  SegRec* segs = nullptr;
  uint32_t* items = nullptr;  // eight words per item: {gap, segments (0: no room), offset, -, ready, -, -, -}
  GapOut* outs = nullptr;
  unsigned long long* ctr = nullptr;
  uint32_t cap_items = 0, cap_segs = 0;
};

// launch_fill_seg: flank look-ups inside the fill kernel (fill_seg.hip, SegArgs.inl_*)
struct SegInline {
  FlankLookup lk;
  const char* text = nullptr;        // device-readable: the list's flank text (GapDev.rs_mask: the gap's offset)
  uint32_t* nodes_dev = nullptr;     // the table flank_nodes points at, writable
  uint32_t* nodes_host = nullptr;    // its pinned copy for the host's half (device-writable)
  uint32_t text_stride = 0;          // not 0: gap i's text at text + i * text_stride (a list without a bad flank)
};

// launch_fill_seg: tracebacks that have no choice to make, by the gap's own wave (fill_seg.hip, SegArgs.tr_*)
struct SegTrace {
  uint32_t* results = nullptr;  // g2s_result[n], device-writable
  char* arena = nullptr;        // device-writable
  unsigned long long arena_base = 0, max_states = 0;
  const char* chu = nullptr;
  const char* chd = nullptr;
  int k = 0;
  char* spec_text = nullptr;     // device memory, the arena's size: guessed tracebacks (G2S_DEVA_SPEC), or null
  uint32_t* spec_res = nullptr;  // device memory, g2s_result[n]
  uint32_t guess_until = 0;      // two waves per gap: how many gaps — in the order their searches end — may guess (0: all)
};

// (tools, G2S_D2_LOG) where in d2_list the fill kernels note when a gap's closure was listed; 0: nowhere
extern uint32_t d2_ticks_offset;

size_t fill_seg_lds_bytes();
uint32_t fill_seg_dbg_words();  // words per gap of the optional diagnostics buffer
// phases A-D1 of every listed gap in one launch; results land in pinned host memory exactly as
// the LDS tier leaves them (GapOut per gap, closures packed by an atomic cursor, completion list)
hipError_t launch_fill_seg(hipStream_t st, uint32_t ngaps, const uint32_t* succ, const uint32_t* urec /* seg_tables.hip */,
                           const GapDev* gaps,
                           const uint32_t* gap_ids, const uint32_t* flank_nodes, SubRec* sub_out /* pinned host */,
                           unsigned long long out_cap /* records */, unsigned long long* out_counter, GapOut* outs,
                           GapOut* outs_host /* pinned host */, uint32_t* done_list /* pinned host */, int skip_confident,
                           uint32_t* dbg /* nullptr, or ngaps * fill_seg_dbg_words() words */,
                           bool two_waves /* g2s_fill_seg2: phase A on a second wave beside the first half of phase B */,
                           // announcements in batches of pub_batch gaps per XCD (fill_seg.hip, `publish`): 8 zeroed
                           // counters, 8 lists of xcd_stride >= ngaps entries set to 0xFFFFFFFF; pub_batch 1 = per gap
                           unsigned long long* xcd_tickets, uint32_t* xcd_list, uint32_t xcd_stride, uint32_t pub_batch,
                           // results stay on the device (sub_out, outs: device memory; outs_host, done_list unused):
                           // phase D3 follows on the stream (d3_device.hip); ovf_list (optional, device memory, ngaps
                           // words): the gaps that outgrew the tier's capacities, counted in out_counter[1], for
                           // launch_fill_segw(..., ngaps_dev = out_counter + 1) behind this launch
                           bool resident = false, uint32_t* ovf_list = nullptr,
                           // (resident) the gaps whose closure the kernel leaves unanalysed, counted in out_counter[4]:
                           // g2s_d2_* (d2_device.hip) behind the fill kernels; null: the host analyses them
                           uint32_t* d2_list = nullptr, uint32_t d2_tag = 0u /* SegArgs.d2_tag */,
                           // (resident) the waves resolve their gap's flank k-mers themselves (SegArgs.inl_*): the flank
                           // text, where the ids go besides flank_nodes' own table; null: flank_nodes holds them already
                           const SegInline* inl = nullptr,
                           // (resident, lists that are not deep) the closures the host will analyse also go to pinned
                           // memory as their gaps end (SegArgs.early_*); null: behind phase D3's hand-over
                           const SegEarly* early = nullptr,
                           // (resident) single-path gaps traced by their own waves (SegArgs.tr_*); null: none
                           const SegTrace* tr = nullptr,
                           // (resident) the gaps' descriptors as GapLite records (fill_device.h) + the list's constants:
                           // `gaps` is not read then
                           const GapLite* lite = nullptr, int lite_e = 0, int lite_all_paths = 0,
                           // the launch's own start and stop times into these HIP events (hipExtLaunchKernelGGL: no
                           // packets of their own in front of and behind the kernel, as two hipEventRecord calls cost
                           // the stream ~10 us a launch); null: not timed
                           hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// The large variant: `workgroups` persistent workgroups (one per compute unit) take the listed gaps
// in order from the counter *next_gap (zero before the launch); scratch: fill_segx_scratch_bytes().
size_t fill_segx_lds_bytes();
size_t fill_segx_scratch_bytes(uint32_t workgroups);
uint32_t fill_segx_dbg_words();
hipError_t launch_fill_segx(hipStream_t st, uint32_t ngaps, uint32_t workgroups, const uint32_t* succ, const uint32_t* urec,
                            const GapDev* gaps, const uint32_t* gap_ids, const uint32_t* flank_nodes, SubRec* sub_out,
                            unsigned long long out_cap, unsigned long long* out_counter, GapOut* outs, GapOut* outs_host,
                            uint32_t* done_list, int skip_confident, uint32_t* dbg, uint32_t* scratch,
                            unsigned long long* next_gap);

// The large variant on a workgroup of eight waves per gap (fill_segw.hip): same arguments; the workgroups take all of a
// compute unit's LDS, scratch: fill_segw_scratch_bytes().  resident: closures and records stay in device memory.
size_t fill_segw_lds_bytes();
size_t fill_segw_scratch_bytes(uint32_t workgroups);
hipError_t launch_fill_segw(hipStream_t st, uint32_t ngaps, uint32_t workgroups, const uint32_t* succ, const uint32_t* urec,
                            const GapDev* gaps, const uint32_t* gap_ids, const uint32_t* flank_nodes, SubRec* sub_out,
                            unsigned long long out_cap, unsigned long long* out_counter, GapOut* outs, GapOut* outs_host,
                            uint32_t* done_list, int skip_confident, uint32_t* dbg, uint32_t* scratch,
                            unsigned long long* next_gap, bool resident = false,
                            // the number of listed gaps is read from device memory when the kernel starts (the list
                            // was written by the launch in front: launch_fill_seg's ovf_list); ngaps is then the most
                            // there can be
                            const unsigned long long* ngaps_dev = nullptr, const SegEarly* early = nullptr,
                            uint32_t* d2_list = nullptr, uint32_t d2_tag = 0u /* as launch_fill_seg */,
                            const GapLite* lite = nullptr, int lite_e = 0, int lite_all_paths = 0 /* as launch_fill_seg */);

}  // namespace g2s
