// gap2seq_amd/csrc/dbg_gpu.hip — unitig-ordered node numbering on the GPU.
//
// The graph build (replaces gatb Graph::create, /root/reference/src/Gap2Seq.cpp:193-219)
// ends with numbering the k-mers along maximal non-branching paths (dbg.cpp:unitig_order).
// On the host that walk is one dependent, cache-missing load per k-mer: 0.35 s of the 0.49 s
// build at 3 Mbp, 11.7 s of 16.6 s at 60 Mbp.  Here it is list ranking by pointer jumping:
//   nxt[v]   the successor of oriented node v when the edge is unitig-internal (same
//            predicate as the host's `step`), else INVALID;
//   chains   every unitig is two mirrored chains (v -> w  <=>  w^1 -> v^1); a chain's head is
//            the node without an internal predecessor;
//   ranking  pd[v] = (predecessor, distance) is squared log2(longest unitig) times until it
//            holds (head, distance from head);
//   ids      of the two mirrored chains the one with the smaller head is kept; kept heads take
//            consecutive id ranges in head order (exclusive scan of the chain lengths), node
//            id = base[head] + distance, orientation bit = the node's strand on that chain.
// Circular unitigs have no head; their nodes come back unnumbered and the host walk numbers
// them after the rest.  HBM-bound random access: ~10 ms at 3 Mbp.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "dbg.hpp"

namespace {

constexpr uint32_t INV = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t only_succ(const uint32_t* __restrict__ succ, uint32_t v, uint32_t* deg) {
  const uint4 r = *(const uint4*)(succ + (size_t)v * 4);
  const uint32_t d = (r.x != INV) + (r.y != INV) + (r.z != INV) + (r.w != INV);
  *deg = d;
  return r.x != INV ? r.x : r.y != INV ? r.y : r.z != INV ? r.z : r.w;
}

// the unique continuation v -> w when the edge is unitig-internal (dbg.cpp: step)
__global__ void k_next(const uint32_t* __restrict__ succ, uint32_t n2, uint32_t* __restrict__ nxt) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n2) return;
  uint32_t deg, w = only_succ(succ, v, &deg), out = INV;
  if (deg == 1 && (w >> 1) != (v >> 1)) {
    uint32_t dback, back = only_succ(succ, w ^ 1u, &dback);  // in-degree of w
    if (dback == 1 && back == (v ^ 1u)) out = w;
  }
  nxt[v] = out;
}

__global__ void k_init(const uint32_t* __restrict__ nxt, uint32_t n2, uint64_t* __restrict__ pd) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n2) return;
  const uint32_t m = nxt[v ^ 1u];  // v has an internal predecessor p  <=>  v^1 -> p^1 is internal
  pd[v] = m == INV ? ((uint64_t)v << 32) : (((uint64_t)(m ^ 1u) << 32) | 1ull);
}

__global__ void k_jump(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n2, uint32_t* changed) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n2) return;
  const uint64_t a = in[v];
  const uint32_t p = (uint32_t)(a >> 32);
  const uint64_t b = in[p];
  const uint32_t pp = (uint32_t)(b >> 32);
  if (pp != p) {  // p is not a head yet: jump over it
    out[v] = ((uint64_t)pp << 32) | (uint32_t)((uint32_t)a + (uint32_t)b);
    *changed = 1u;
  } else {
    out[v] = a;
  }
}

// tails report the length of their chain to the head
__global__ void k_tail(const uint32_t* __restrict__ nxt, const uint64_t* __restrict__ pd, uint32_t n2,
                       uint32_t* __restrict__ len, uint32_t* __restrict__ tailof) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n2 || nxt[v] != INV) return;
  const uint64_t a = pd[v];
  const uint32_t h = (uint32_t)(a >> 32);
  if ((uint32_t)(pd[h] >> 32) != h) return;  // on a cycle (cannot happen for a tail, kept for safety)
  len[h] = (uint32_t)a + 1u;
  tailof[h] = v;
}

// cnt[v] = length of the chain headed by v when that chain is the one kept of its mirrored pair
__global__ void k_keep(const uint32_t* __restrict__ len, const uint32_t* __restrict__ tailof, uint32_t n2,
                       uint32_t* __restrict__ cnt, uint32_t* n_kept) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n2) return;
  const uint32_t l = len[v];
  const bool keep = l != 0u && v < (tailof[v] ^ 1u);
  cnt[v] = keep ? l : 0u;
  if (keep) atomicAdd(n_kept, 1u);
}

__global__ void k_assign(const uint64_t* __restrict__ pd, const uint32_t* __restrict__ cnt,
                         const uint32_t* __restrict__ base, uint32_t n2, uint32_t* __restrict__ rank2id,
                         uint8_t* __restrict__ flip) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n2) return;
  const uint64_t a = pd[v];
  const uint32_t h = (uint32_t)(a >> 32);
  if ((uint32_t)(pd[h] >> 32) != h || cnt[h] == 0u) return;  // cycle, or the mirrored chain is the kept one
  rank2id[v >> 1] = base[h] + (uint32_t)a;
  flip[v >> 1] = (uint8_t)(v & 1u);
}

struct Dev {
  void* p = nullptr;
  ~Dev() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};

}  // namespace

namespace g2s {

// rank-space successor table in, numbering out.  Entries of rank2id left at kInvalidNode
// (circular unitigs) are numbered by the caller from *next_id on.  Returns false (with a
// reason) when the device cannot be used; the caller then runs the host walk.
bool unitig_order_gpu(const std::vector<uint32_t>& succ_r, uint64_t n, int device, std::vector<uint32_t>* rank2id,
                      std::vector<uint8_t>* flip, uint64_t* n_unitigs, uint32_t* next_id, std::string* why) {
#define G2S_GPU_TRY(expr)                                                                 \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) { if (why) *why = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } \
  } while (0)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { if (why) *why = "no device"; return false; }
  if (n == 0 || 2 * n >= (1ull << 31)) { if (why) *why = "size"; return false; }
  G2S_GPU_TRY(hipSetDevice(device));
  const uint32_t n2 = (uint32_t)(2 * n);
  Dev d_succ, d_nxt, d_pd0, d_pd1, d_len, d_tail, d_cnt, d_base, d_id, d_flip, d_flag, d_tmp;
  G2S_GPU_TRY(d_succ.alloc(succ_r.size() * 4));
  G2S_GPU_TRY(d_nxt.alloc((size_t)n2 * 4));
  G2S_GPU_TRY(d_pd0.alloc((size_t)n2 * 8));
  G2S_GPU_TRY(d_pd1.alloc((size_t)n2 * 8));
  G2S_GPU_TRY(d_len.alloc((size_t)n2 * 4));
  G2S_GPU_TRY(d_tail.alloc((size_t)n2 * 4));
  G2S_GPU_TRY(d_cnt.alloc((size_t)n2 * 4));
  G2S_GPU_TRY(d_base.alloc((size_t)n2 * 4));
  G2S_GPU_TRY(d_id.alloc((size_t)n * 4));
  G2S_GPU_TRY(d_flip.alloc((size_t)n));
  G2S_GPU_TRY(d_flag.alloc(16));
  G2S_GPU_TRY(hipMemcpy(d_succ.p, succ_r.data(), succ_r.size() * 4, hipMemcpyHostToDevice));
  G2S_GPU_TRY(hipMemset(d_len.p, 0, (size_t)n2 * 4));
  G2S_GPU_TRY(hipMemset(d_tail.p, 0, (size_t)n2 * 4));
  G2S_GPU_TRY(hipMemset(d_id.p, 0xFF, (size_t)n * 4));
  G2S_GPU_TRY(hipMemset(d_flip.p, 0, (size_t)n));
  const dim3 blk(256), grd((n2 + 255) / 256);
  hipLaunchKernelGGL(k_next, grd, blk, 0, 0, (const uint32_t*)d_succ.p, n2, (uint32_t*)d_nxt.p);
  hipLaunchKernelGGL(k_init, grd, blk, 0, 0, (const uint32_t*)d_nxt.p, n2, (uint64_t*)d_pd0.p);
  uint64_t *cur = (uint64_t*)d_pd0.p, *oth = (uint64_t*)d_pd1.p;
  for (int round = 0; round < 34; round++) {  // 2^32 > any chain; cycles never settle and stop here
    G2S_GPU_TRY(hipMemset(d_flag.p, 0, 4));
    hipLaunchKernelGGL(k_jump, grd, blk, 0, 0, (const uint64_t*)cur, oth, n2, (uint32_t*)d_flag.p);
    uint32_t changed = 0;
    G2S_GPU_TRY(hipMemcpy(&changed, d_flag.p, 4, hipMemcpyDeviceToHost));
    std::swap(cur, oth);
    if (!changed) break;
  }
  hipLaunchKernelGGL(k_tail, grd, blk, 0, 0, (const uint32_t*)d_nxt.p, (const uint64_t*)cur, n2, (uint32_t*)d_len.p,
                     (uint32_t*)d_tail.p);
  G2S_GPU_TRY(hipMemset(d_flag.p, 0, 4));
  hipLaunchKernelGGL(k_keep, grd, blk, 0, 0, (const uint32_t*)d_len.p, (const uint32_t*)d_tail.p, n2, (uint32_t*)d_cnt.p,
                     (uint32_t*)d_flag.p);
  size_t tmp_bytes = 0;
  G2S_GPU_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, (const uint32_t*)d_cnt.p, (uint32_t*)d_base.p, 0u, (size_t)n2,
                                      rocprim::plus<uint32_t>()));
  G2S_GPU_TRY(d_tmp.alloc(tmp_bytes));
  G2S_GPU_TRY(rocprim::exclusive_scan(d_tmp.p, tmp_bytes, (const uint32_t*)d_cnt.p, (uint32_t*)d_base.p, 0u, (size_t)n2,
                                      rocprim::plus<uint32_t>()));
  hipLaunchKernelGGL(k_assign, grd, blk, 0, 0, (const uint64_t*)cur, (const uint32_t*)d_cnt.p, (const uint32_t*)d_base.p, n2,
                     (uint32_t*)d_id.p, (uint8_t*)d_flip.p);
  G2S_GPU_TRY(hipGetLastError());
  rank2id->assign((size_t)n, kInvalidNode);
  flip->assign((size_t)n, 0);
  G2S_GPU_TRY(hipMemcpy(rank2id->data(), d_id.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  G2S_GPU_TRY(hipMemcpy(flip->data(), d_flip.p, (size_t)n, hipMemcpyDeviceToHost));
  uint32_t kept = 0, last_base = 0, last_cnt = 0;
  G2S_GPU_TRY(hipMemcpy(&kept, d_flag.p, 4, hipMemcpyDeviceToHost));
  G2S_GPU_TRY(hipMemcpy(&last_base, (const uint32_t*)d_base.p + (n2 - 1), 4, hipMemcpyDeviceToHost));
  G2S_GPU_TRY(hipMemcpy(&last_cnt, (const uint32_t*)d_cnt.p + (n2 - 1), 4, hipMemcpyDeviceToHost));
  *n_unitigs = kept;
  *next_id = last_base + last_cnt;  // ids handed out so far
  return true;
#undef G2S_GPU_TRY
}

}  // namespace g2s
