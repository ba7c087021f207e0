// gap2seq_amd/csrc/seg_tables.h — see seg_tables.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#define G2S_REM_CAP 65535u /* rem[] saturates here: longer unitigs are walked in several segments */

namespace g2s {

// rem[v] for every oriented node (2n entries), from the device copy of the unitig-start bitmap.
hipError_t build_rem_table(const uint64_t* ustart_dev, uint64_t n, uint32_t** rem_out);
// urec[v] (8 words per oriented node) = {succ[last node of v's unitig walk][0..3], rem[v], 0, 0, 0}
hipError_t build_urec_table(const uint32_t* succ_dev, const uint32_t* rem_dev, uint64_t n, uint32_t** urec_out);

}  // namespace g2s
