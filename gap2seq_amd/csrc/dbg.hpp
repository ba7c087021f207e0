// gap2seq_amd/csrc/dbg.hpp — host-side de Bruijn graph: exact solid k-mer set,
// unitig-ordered node numbering and the 4-slot oriented successor table that is
// uploaded to HBM.
//
// Replaces gatb Graph::create / Graph::load as used at
// /root/reference/src/Gap2Seq.cpp:193-219 and the neighbour primitives
// Graph::successors / predecessors / contains / buildNode / toString whose call
// sites are Gap2Seq.cpp:879,884,924,955,957,995,1000,1043,1084,1086,1114,1199,
// 1203,1204,1263,1271,1452,1456,1476.  Membership is exact (no Bloom filter).
#pragma once
#include <mutex>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "kmer.hpp"

namespace g2s {

static const uint32_t kInvalidNode = 0xFFFFFFFFu;
static const uint32_t kUstartPad = 64;  // 64-bit words of all-ones padding on either side of the device bitmap

struct DeviceGraph {
  uint32_t* succ = nullptr;  // [2n*4]
  uint32_t* pred = nullptr;  // [2n*4], only when k is even (palindromic k-mers exist)
  uint64_t* ustart = nullptr;  // unitig-start bitmap, offset by kUstartPad words of all-ones padding
  uint32_t* rem = nullptr;     // [2n] unitig-internal steps left from an oriented node (seg_tables.hip), odd k only
  uint32_t* urec = nullptr;    // [2n][8] successor record of the end of the node's unitig walk + rem (seg_tables.hip)
  uint64_t bytes = 0;
};

struct Graph {
  int k = 0;
  int solid = 0;      // abundance threshold the set was built with (0: unknown, a cache written before it was recorded)
  bool wide = false;  // 128-bit k-mers (k >= 32)
  uint64_t n = 0;     // canonical solid k-mers
  uint64_t n_unitigs = 0;
  // sorted canonical k-mers; exactly one of the two is used
  std::vector<uint64_t> kmers64;
  std::vector<u128> kmers128;
  std::vector<uint32_t> bucket;   // prefix index over the sorted array
  int bucket_bits = 0;
  std::vector<uint32_t> rank2id;  // sorted rank -> node index (unitig order)
  std::vector<uint32_t> id2rank;
  // Orientation bit of an oriented node id is RELATIVE TO ITS UNITIG: bit 0 = the
  // direction in which the unitig was numbered.  flip[rank] = GATB strand
  // (0 = canonical) of that direction, so GATB strand = orientation ^ flip.  Inside a
  // unitig the only successor of an even id v is v+2 and of an odd id v is v-2, which
  // lets the kernels walk unitigs by arithmetic and verify in bulk.  GATB's strand is
  // only a label separating the two DP rows of a k-mer (Node::operator== ignores it),
  // so any consistent labelling gives identical results.
  std::vector<uint8_t> flip;
  // oriented node = 2*index + orientation.  succ[v*4 + nt] = oriented successor
  // obtained by appending nt (A,C,T,G) or kInvalidNode.
  std::vector<uint32_t> succ;
  std::vector<uint32_t> pred;     // explicit predecessor table, even k only
  std::vector<uint8_t> lastnt;    // [2n] code of the last base of the oriented sequence
  // The same as upper-case characters, one array per orientation ([n] each, by k-mer index): the bases along a
  // unitig are then consecutive bytes, and the traceback copies a run of states instead of looking every base
  // up (ensure_lastch builds them once, from lastnt).
  mutable std::vector<char> lastch_up, lastch_dn;
  mutable std::once_flag lastch_once;
  void ensure_lastch() const {
    std::call_once(lastch_once, [this]() {
      static const char kUpChar[4] = {'A', 'C', 'T', 'G'};  // GATB codes (kmer.hpp)
      lastch_up.resize((size_t)n);
      lastch_dn.resize((size_t)n);
      for (size_t i = 0; i < (size_t)n; i++) { lastch_up[i] = kUpChar[lastnt[2 * i] & 3]; lastch_dn[i] = kUpChar[lastnt[2 * i + 1] & 3]; }
    });
  }
  // bit i set: k-mer index i is the first of its unitig in numbering order, i.e. the edge
  // 2(i-1) -> 2i is NOT unitig-internal.  Between two set bits the walk arithmetic
  // (v +/- 2) is exact: every node there has exactly one predecessor and one successor.
  std::vector<uint64_t> ustart;
  std::map<int, DeviceGraph> dev; // per device copies

  // buildNode + contains
  uint32_t node_of(const char* s) const;
  inline uint32_t succ_of(uint32_t v, int nt) const { return succ[(size_t)v * 4 + nt]; }
  // predecessor i in GATB order (prepend T,G,A,C)
  inline uint32_t pred_of(uint32_t v, int nt) const {
    if (!pred.empty()) return pred[(size_t)v * 4 + nt];
    uint32_t w = succ[(size_t)(v ^ 1u) * 4 + nt];
    return w == kInvalidNode ? w : (w ^ 1u);
  }
  std::string node_string(uint32_t v) const;
  inline char last_char(uint32_t v) const { return kNtChar[lastnt[v]]; }
};

// seqs may contain any bytes; k-mers containing N/n are skipped (GATB model).
Graph* graph_build(const std::vector<std::pair<const char*, uint64_t>>& seqs, int k, int solid, int nthreads,
                   std::string* err);
bool graph_save(const Graph& g, const std::string& path, std::string* err);
Graph* graph_load(const std::string& path, std::string* err);

}  // namespace g2s
