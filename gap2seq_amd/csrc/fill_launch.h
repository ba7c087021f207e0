// gap2seq_amd/csrc/fill_launch.h — host-callable launchers of the HIP kernels
// in fill_kernels.hip (phases A-C of /root/reference/src/Gap2Seq.cpp:858-1167).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "fill_device.h"

namespace g2s {

hipError_t launch_right_bfs(hipStream_t st, uint32_t ngaps, const uint32_t* succ, const uint32_t* predtab,
                            const GapDev* gaps, const uint32_t* gap_ids, const uint32_t* flank_nodes, uint32_t* rs_all,
                            uint32_t* rlog_all, GapOut* outs);

hipError_t launch_left_dp(hipStream_t st, uint32_t ngaps, const uint32_t* succ, const GapDev* gaps,
                          const uint32_t* gap_ids, const uint32_t* flank_nodes, const uint32_t* rs_all,
                          uint64_t* st_keys_all, uint32_t* st_cnt_all, uint32_t* slog_all, GapOut* outs);

// phase D1 on the device (Gap2Seq.cpp:1169-1312) + the traceback's closure
hipError_t launch_extract(hipStream_t st, uint32_t ngaps, const uint32_t* succ, const uint32_t* predtab,
                          const GapDev* gaps, const uint32_t* gap_ids, const uint32_t* flank_nodes,
                          const uint64_t* st_keys_all, const uint32_t* st_cnt_all, uint32_t* st_mark_all,
                          SubState* sub_scratch, SubState* sub_out, unsigned long long* out_counter, GapOut* outs,
                          int skip_confident);

// ---- LDS tier (fill_lds.hip) ----------------------------------------------------------
size_t fill_lds_bytes(uint32_t rs_cap, uint32_t fcap);
size_t extract_lds_bytes(uint32_t fcap);
uint32_t fill_lds_frontier_cap();
uint32_t fill_lds_max_fuz();
// phases A-D1 of every listed gap in one launch; results land in pinned host memory
hipError_t launch_fill_lds(hipStream_t st, uint32_t ngaps, uint32_t rs_cap_max, uint32_t num_oriented,
                           const uint32_t* succ, const uint64_t* ustart, const GapDev* gaps, const uint32_t* gap_ids,
                           const uint32_t* flank_nodes, uint64_t* log_all, uint32_t* lvl_all, uint32_t* plk_all,
                           uint64_t* xl_all, uint64_t* xo_all, SubRec* sub_scratch, SubRec* sub_out /* pinned host */,
                           unsigned long long out_cap /* records */, unsigned long long* out_counter, GapOut* outs,
                           GapOut* outs_host /* pinned host */, uint32_t* done_list /* pinned host, ngaps entries */,
                           int skip_confident,
                           uint32_t* rs_global /* nullptr: right set in LDS */, uint32_t fcap /* frontier capacity */,
                           uint32_t* rs_pool /* spill pool for LDS right sets, 0xFF filled */, uint32_t pool_chunks,
                           uint32_t chunk_entries /* power of two */,
                           void* log_pool /* chunks for state logs that outgrow their slice, or nullptr */,
                           uint32_t log_chunks, uint32_t log_chunk_states);
size_t fill_lds_log_chunk_bytes(uint32_t states);

}  // namespace g2s
