// gap2seq_amd/csrc/post.hpp — host part of fill_gap's phase D, driven by the DP
// state log that the HIP kernels produced for one gap:
//   D1 subgraph extraction   /root/reference/src/Gap2Seq.cpp:1169-1312
//   D2 SCC contraction + the "vertex on all paths" rule        :1314-1435
//   D3 random traceback                                         :1437-1522
// D1/D2 are independent per gap (run on a thread pool); D3 consumes the single
// rand() stream and therefore runs in gap order on one thread.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/g2s.h"
#include "dbg.hpp"
#include "fill_device.h"
#include "glibc_rand.hpp"

namespace g2s {

// One gap as the host keeps it.
struct GapJob {
  std::string left, right;
  int g = 0, lmf = 0, rmf = 0;
  int skip_if_prev_right_fuz_gt = -1;
  bool bad_flank = false;
  // oriented node (or kInvalidNode) of: left.substr(d,k) d=0..lmf | right-BFS
  // seeds right.substr(len-k-j,k) j=0..rmf | targets right.substr(j,k) j=0..rmf
  std::vector<uint32_t> flank_nodes;
  const uint32_t* lseeds() const { return flank_nodes.data(); }
  const uint32_t* rseeds() const { return flank_nodes.data() + (lmf + 1); }
  const uint32_t* targets() const { return flank_nodes.data() + (lmf + 1) + (rmf + 1); }
  size_t buf_bytes(int k, int d_err) const { return (size_t)(g + k + d_err + lmf + rmf + 1 + 2); }
};

// Device results of one gap, as seen on the host after the copy back.
struct DpView {
  const GapOut* out = nullptr;
  const uint32_t* lvl = nullptr;  // D+2 offsets relative to `states`
  uint64_t* states = nullptr;     // (node << 32 | count), level by level; sorted per level by prepare()
  int D = 0;
};

struct PostPrep {
  bool phase_d = false;   // count > 0 && pathLengths non-empty
  int count = 0;          // value fill_gap will return (before the memory verdict)
  uint32_t flags = 0;     // G2S_GAP_* bits found on the host (Q7 in D1)
  uint64_t sub[6] = {0, 0, 0, 0, 0, 0};
  uint64_t xD = 0, sD = 0;
  std::unordered_map<uint32_t, int> vertex_of;  // canonical index -> subgraph vertex
  std::vector<int> branch;                      // per real vertex; 0 = sink, 1 = source
};

struct FillParams {
  int k = 31, d_err = 500;
  bool skip_confident = false, all_paths = true, unique_paths = false;
};

// Sort every level by node id (the kernels emit levels in arrival order).
void dp_sort_levels(DpView* v);
// count of state (node, depth) or 0
uint32_t dp_find(const DpView& v, int depth, uint32_t node);

void post_extract(const Graph& g, const FillParams& p, const GapJob& job, const DpView& v, PostPrep* out);
// Writes the reference's `fill` buffer into buf (size job.buf_bytes) and fills res.
void post_traceback(const Graph& g, const FillParams& p, const GapJob& job, const DpView& v, const PostPrep& prep,
                    GlibcRand& rng, char* buf, g2s_result* res);

}  // namespace g2s
