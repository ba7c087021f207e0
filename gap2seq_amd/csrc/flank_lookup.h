// gap2seq_amd/csrc/flank_lookup.h — see flank_lookup.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace g2s {

// device copy of what Graph::node_of searches (dbg.hpp): the sorted canonical k-mers, their prefix
// index, and the rank -> (node index, strand flip) tables
struct FlankLookup {
  const void* kmers = nullptr;      // uint64_t[n] (k <= 31) or unsigned __int128[n]
  const uint32_t* bucket = nullptr; // [(1 << bucket_bits) + 1]
  const uint32_t* rank2node = nullptr;  // by sorted rank: 2 * node id | strand flip of the canonical k-mer
  int32_t k = 0, bucket_bits = 0, wide = 0, pad = 0;
};

// one gap: where its flank text is — the first k+lmf characters of the left flank, the first k+rmf
// and the last k+rmf characters of the right flank (fill_gap reads nothing else of them), start
// 4-byte aligned — and where its (lmf+1) + 2 (rmf+1) node ids go
struct FlankDesc {
  uint32_t text_off;
  uint32_t flank_off;
  uint16_t lmf, rmf;  // rmf: bit 15 = the right flank is exactly k + rmf characters and stands in the text once
};
#define G2S_FLANK_RIGHT_ONCE 0x8000u
#define G2S_FLANK_TEXT_MAX 2048 /* bytes of flank text per gap the kernel stages in LDS */

hipError_t launch_resolve_flanks(hipStream_t st, const FlankLookup& lk, uint32_t ngaps, const FlankDesc* desc /* device-readable */,
                                 const char* text /* device-readable */, uint32_t* nodes_dev, uint32_t* nodes_host);

}  // namespace g2s
