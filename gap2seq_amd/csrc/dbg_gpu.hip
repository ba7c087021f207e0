// gap2seq_amd/csrc/dbg_gpu.hip — the graph build after the solid k-mer set, on the GPU.
//
// The build (replaces gatb Graph::create, /root/reference/src/Gap2Seq.cpp:193-219) has four
// steps after the sorted set of solid canonical k-mers exists (dbg.cpp: count_solid):
//   1. successor table in sorted-rank space: 8 neighbour k-mers per k-mer, each a binary
//      search inside its prefix bucket (k_succ; 64- and 128-bit k-mers);
//   2. numbering along maximal non-branching paths.  On the host this is one dependent,
//      cache-missing load per k-mer (0.35 s of a 0.49 s build at 3 Mbp, 11.7 s of 16.6 s at
//      60 Mbp); here it is list ranking by pointer jumping:
//        nxt[v]   the successor of oriented node v when the edge is unitig-internal (same
//                 predicate as the host's `step`), else INVALID;
//        chains   every unitig is two mirrored chains (v -> w  <=>  w^1 -> v^1); a chain's
//                 head is the node without an internal predecessor;
//        ranking  pd[v] = (predecessor, distance) is squared log2(longest unitig) times
//                 until it holds (head, distance from head);
//        ids      of the two mirrored chains the one with the smaller head is kept; kept
//                 heads take consecutive id ranges in head order (exclusive scan of the
//                 chain lengths), node id = base[head] + distance, orientation bit = the
//                 node's strand on that chain.
//      Circular unitigs have no head; the host walk numbers them after the rest.
//   3. the tables permuted into id space + the last base of every oriented node (k_remap);
//   4. the unitig-start bitmap (k_ustart).
// The id-space table and the bitmap stay on the device: they are the graph's copy for the
// fill path.  All of it is HBM-bound random access: 0.05 s at 3 Mbp.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <functional>
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

// ---- successor table in sorted-rank space (dbg.cpp: build_tables_rank, odd k) -------------
typedef unsigned __int128 u128;

__device__ __forceinline__ uint64_t d_revcomp32(uint64_t x) {  // all 32 bases of a word (kmer.hpp)
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = __builtin_bswap64(x);
  return x ^ 0xAAAAAAAAAAAAAAAAULL;
}
__device__ __forceinline__ uint64_t d_revcomp(uint64_t x, int k) { return d_revcomp32(x) >> (64 - 2 * k); }
__device__ __forceinline__ u128 d_revcomp(u128 x, int k) {
  const u128 y = ((u128)d_revcomp32((uint64_t)x) << 64) | (u128)d_revcomp32((uint64_t)(x >> 64));
  return y >> (128 - 2 * k);
}

template <class KT>
__device__ __forceinline__ uint32_t d_rank_of(const KT* __restrict__ v, const uint32_t* __restrict__ bucket, int shift,
                                              KT x) {
  const uint32_t b = (uint32_t)(x >> shift);
  uint32_t lo = bucket[b], hi = bucket[b + 1];
  const uint32_t end = hi;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (v[mid] < x) lo = mid + 1; else hi = mid;
  }
  return (lo < end && v[lo] == x) ? lo : INV;
}

template <class KT>
__global__ void k_succ(const KT* __restrict__ v, const uint32_t* __restrict__ bucket, int shift, uint32_t n2, int k,
                       uint32_t* __restrict__ succ) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;  // oriented node in rank space: 2*rank + strand
  if (t >= n2) return;
  const KT mask = (2 * k >= (int)(8 * sizeof(KT))) ? ~(KT)0 : ((((KT)1) << (2 * k)) - 1);
  const KT c = v[t >> 1], rc = d_revcomp(c, k);
  const KT seq = (t & 1u) ? rc : c, rseq = (t & 1u) ? c : rc;
  uint32_t out[4];
#pragma unroll
  for (int nt = 0; nt < 4; nt++) {
    const KT y = ((seq << 2) | (KT)nt) & mask;
    const KT ry = (rseq >> 2) | ((KT)(nt ^ 2) << (2 * (k - 1)));
    const uint32_t r = d_rank_of<KT>(v, bucket, shift, y < ry ? y : ry);
    out[nt] = r == INV ? INV : 2u * r + (y < ry ? 0u : 1u);
  }
  *(uint4*)(succ + (size_t)t * 4) = make_uint4(out[0], out[1], out[2], out[3]);
}

// ---- tables in id space (dbg.cpp: finish_graph) ----------------------------------------------
template <class KT>
__global__ void k_remap(const KT* __restrict__ v, const uint32_t* __restrict__ succ_r, const uint32_t* __restrict__ rank2id,
                        const uint8_t* __restrict__ flip, uint32_t n2, int k, uint32_t* __restrict__ succ_id,
                        uint8_t* __restrict__ lastnt, uint32_t* __restrict__ id2rank) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n2) return;
  const uint32_t r = t >> 1, s = t & 1u;
  const uint32_t row = 2u * rank2id[r] + (s ^ (uint32_t)flip[r]);
  const uint4 in = *(const uint4*)(succ_r + (size_t)t * 4);
  auto remap = [&](uint32_t w) -> uint32_t { return w == INV ? INV : 2u * rank2id[w >> 1] + ((w & 1u) ^ (uint32_t)flip[w >> 1]); };
  *(uint4*)(succ_id + (size_t)row * 4) = make_uint4(remap(in.x), remap(in.y), remap(in.z), remap(in.w));
  const KT c = v[r];
  lastnt[row] = s == 0 ? (uint8_t)(c & 3) : (uint8_t)(((c >> (2 * (k - 1))) & 3) ^ 2);
  if (s == 0) id2rank[rank2id[r]] = r;
}

// unitig-start bitmap from the id-space table (dbg.cpp: build_ustart); one wave per 64-bit word
__global__ void k_ustart(const uint32_t* __restrict__ succ, uint32_t n, uint32_t nbits, uint64_t* __restrict__ words) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool start = true;
  if (i < n && i > 0) {
    const uint32_t v = 2u * (i - 1u), want = 2u * i;
    uint32_t deg, w = only_succ(succ, v, &deg);
    if (deg == 1 && w == want) {
      uint32_t dback, back = only_succ(succ, want ^ 1u, &dback);
      start = !(dback == 1 && back == (v ^ 1u));
    }
  }
  const uint64_t m = __ballot(i < nbits && start);
  if ((threadIdx.x & 63u) == 0 && i < nbits) words[i >> 6] = m;
}

// ---- solid k-mer set (dbg.cpp: count_solid) ----------------------------------------------
// one thread per text position: the canonical k-mer starting there, or all-ones when the
// window holds an N/n (GATB: k-mers with an invalid character are skipped) or runs off the end
template <class KT>
__global__ void k_extract(const uint8_t* __restrict__ text, uint64_t len, int k, KT* __restrict__ keys) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  KT f = 0;
  bool ok = i + (uint64_t)k <= len;
  if (ok) {
    for (int j = 0; j < k; j++) {
      const uint8_t c = text[i + (uint64_t)j];
      ok = ok && ((c >> 3) & 1u) == 0u;
      f = (f << 2) | (KT)((c >> 1) & 3u);
    }
  }
  const KT r = d_revcomp(f, k);
  keys[i] = ok ? (f < r ? f : r) : ~(KT)0;
}
template <class KT>
__global__ void k_split(const KT* __restrict__ keys, uint64_t n, uint64_t* __restrict__ lo, uint64_t* __restrict__ hi) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  lo[i] = (uint64_t)keys[i];
  hi[i] = (uint64_t)(keys[i] >> 64);
}
__global__ void k_join(const uint64_t* __restrict__ lo, const uint64_t* __restrict__ hi, uint64_t n, u128* __restrict__ keys) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = ((u128)hi[i] << 64) | (u128)lo[i];
}
// heads of runs of equal keys in the sorted array (the all-ones filler is no k-mer)
template <class KT>
__global__ void k_heads(const KT* __restrict__ keys, uint64_t n, uint32_t* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const KT x = keys[i];
  flag[i] = (x != ~(KT)0 && (i == 0 || keys[i - 1] != x)) ? 1u : 0u;
}
template <class KT>
__global__ void k_head_index(const KT* __restrict__ keys, const uint32_t* __restrict__ flag,
                             const uint32_t* __restrict__ pos, uint64_t n, uint32_t* __restrict__ hidx,
                             uint32_t* n_valid) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (flag[i]) hidx[pos[i]] = (uint32_t)i;
  // the first filler (or the end) closes the last run
  if (keys[i] != ~(KT)0 && (i + 1 == n || keys[i + 1] == ~(KT)0)) *n_valid = (uint32_t)(i + 1);
}
__global__ void k_solid_flag(const uint32_t* __restrict__ hidx, uint32_t nheads, uint32_t n_valid, uint32_t solid,
                             uint32_t* __restrict__ keep) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nheads) return;
  const uint32_t end = j + 1 < nheads ? hidx[j + 1] : n_valid;
  keep[j] = (end - hidx[j] >= solid) ? 1u : 0u;
}
template <class KT>
__global__ void k_compact(const KT* __restrict__ keys, const uint32_t* __restrict__ hidx, const uint32_t* __restrict__ keep,
                          const uint32_t* __restrict__ kpos, uint32_t nheads, KT* __restrict__ out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nheads || !keep[j]) return;
  out[kpos[j]] = keys[hidx[j]];
}
// prefix index over the sorted set: bucket[b] = first rank whose top bits are >= b
template <class KT>
__global__ void k_bucket(const KT* __restrict__ v, uint32_t n, int shift, uint32_t nb, uint32_t* __restrict__ bucket) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nb) return;
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((uint64_t)(v[mid] >> shift) < (uint64_t)b) lo = mid + 1; else hi = mid;
  }
  bucket[b] = lo;
}

struct Dev {
  void* p = nullptr;
  ~Dev() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  void* release() { void* q = p; p = nullptr; return q; }
};

}  // namespace

namespace g2s {

// Everything of the graph build after the sorted solid k-mer set (and its prefix index):
// successor table, unitig-ordered numbering, tables in id space, last-base table, unitig-start
// bitmap.  The id-space table and the bitmap stay on the device as the graph's copy for the
// fill path (g2s_graph_upload finds them).  `host_walk` numbers what list ranking leaves
// unnumbered (circular unitigs).  Returns false (with a reason) when the device cannot be
// used; nothing of g is touched then and the caller runs the host build.
template <class KT>
static bool finish_gpu_t(Graph& g, const std::vector<KT>& kmers, int device,
                         const std::function<void(const std::vector<uint32_t>&, uint32_t)>& host_walk, std::string* why) {
#define G2S_GPU_TRY(expr)                                                                 \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) { if (why) *why = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } \
  } while (0)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { if (why) *why = "no device"; return false; }
  const uint64_t n = g.n;
  if (n == 0 || 2 * n >= (1ull << 31)) { if (why) *why = "size"; return false; }
  G2S_GPU_TRY(hipSetDevice(device));
  const uint32_t n2 = (uint32_t)(2 * n);
  const int k = g.k;
  const dim3 blk(256), grd((n2 + 255) / 256);
  Dev d_km, d_bucket, d_succ, d_nxt, d_pd0, d_pd1, d_len, d_tail, d_cnt, d_base, d_id, d_flip, d_flag, d_tmp;
  // ---- successor table, rank space
  G2S_GPU_TRY(d_km.alloc(kmers.size() * sizeof(KT)));
  G2S_GPU_TRY(d_bucket.alloc(g.bucket.size() * 4));
  G2S_GPU_TRY(d_succ.alloc((size_t)n2 * 16));
  G2S_GPU_TRY(hipMemcpy(d_km.p, kmers.data(), kmers.size() * sizeof(KT), hipMemcpyHostToDevice));
  G2S_GPU_TRY(hipMemcpy(d_bucket.p, g.bucket.data(), g.bucket.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_succ<KT>, grd, blk, 0, 0, (const KT*)d_km.p, (const uint32_t*)d_bucket.p, 2 * k - g.bucket_bits, n2, k,
                     (uint32_t*)d_succ.p);
  // ---- numbering along unitigs: list ranking
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
  G2S_GPU_TRY(hipMemset(d_len.p, 0, (size_t)n2 * 4));
  G2S_GPU_TRY(hipMemset(d_tail.p, 0, (size_t)n2 * 4));
  G2S_GPU_TRY(hipMemset(d_id.p, 0xFF, (size_t)n * 4));
  G2S_GPU_TRY(hipMemset(d_flip.p, 0, (size_t)n));
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
  uint32_t kept = 0, last_base = 0, last_cnt = 0;
  G2S_GPU_TRY(hipMemcpy(&kept, d_flag.p, 4, hipMemcpyDeviceToHost));
  G2S_GPU_TRY(hipMemcpy(&last_base, (const uint32_t*)d_base.p + (n2 - 1), 4, hipMemcpyDeviceToHost));
  G2S_GPU_TRY(hipMemcpy(&last_cnt, (const uint32_t*)d_cnt.p + (n2 - 1), 4, hipMemcpyDeviceToHost));
  const uint32_t next_id = last_base + last_cnt;  // ids handed out so far
  // free the ranking scratch before the id-space tables are allocated
  for (Dev* d : {&d_nxt, &d_pd0, &d_pd1, &d_len, &d_tail, &d_cnt, &d_base, &d_tmp}) { if (d->p) (void)hipFree(d->release()); }
  // ---- everything that can still fail is done: from here on g is written
  g.rank2id.assign((size_t)n, kInvalidNode);
  g.flip.assign((size_t)n, 0);
  g.n_unitigs = kept;
  if (next_id < n) {  // circular unitigs: the host walk numbers them after the rest
    std::vector<uint32_t> succ_r((size_t)n2 * 4);
    G2S_GPU_TRY(hipMemcpy(succ_r.data(), d_succ.p, (size_t)n2 * 16, hipMemcpyDeviceToHost));
    G2S_GPU_TRY(hipMemcpy(g.rank2id.data(), d_id.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    G2S_GPU_TRY(hipMemcpy(g.flip.data(), d_flip.p, (size_t)n, hipMemcpyDeviceToHost));
    host_walk(succ_r, next_id);
    G2S_GPU_TRY(hipMemcpy(d_id.p, g.rank2id.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    G2S_GPU_TRY(hipMemcpy(d_flip.p, g.flip.data(), (size_t)n, hipMemcpyHostToDevice));
  }
  // ---- tables in id space
  Dev d_sid, d_last, d_i2r, d_us;
  const size_t sbytes = (size_t)n2 * 16;
  const size_t words = (size_t)((n + 63) / 64 + 1);
  G2S_GPU_TRY(d_sid.alloc(sbytes + 1024));  // the LDS tier may read past the end: INVALID padding
  G2S_GPU_TRY(hipMemset(d_sid.p, 0xFF, sbytes + 1024));
  G2S_GPU_TRY(d_last.alloc((size_t)n2));
  G2S_GPU_TRY(d_i2r.alloc((size_t)n * 4));
  G2S_GPU_TRY(d_us.alloc((words + 2 * kUstartPad) * 8));
  G2S_GPU_TRY(hipMemset(d_us.p, 0xFF, (words + 2 * kUstartPad) * 8));
  hipLaunchKernelGGL(k_remap<KT>, grd, blk, 0, 0, (const KT*)d_km.p, (const uint32_t*)d_succ.p, (const uint32_t*)d_id.p,
                     (const uint8_t*)d_flip.p, n2, k, (uint32_t*)d_sid.p, (uint8_t*)d_last.p, (uint32_t*)d_i2r.p);
  const uint32_t nbits = (uint32_t)(words * 64);
  hipLaunchKernelGGL(k_ustart, dim3((nbits + 255) / 256), blk, 0, 0, (const uint32_t*)d_sid.p, (uint32_t)n, nbits,
                     (uint64_t*)d_us.p + kUstartPad);
  G2S_GPU_TRY(hipGetLastError());
  g.succ.resize((size_t)n2 * 4);
  g.pred.clear();
  g.lastnt.resize((size_t)n2);
  g.id2rank.resize((size_t)n);
  g.ustart.resize(words);
  G2S_GPU_TRY(hipMemcpy(g.succ.data(), d_sid.p, sbytes, hipMemcpyDeviceToHost));
  G2S_GPU_TRY(hipMemcpy(g.lastnt.data(), d_last.p, (size_t)n2, hipMemcpyDeviceToHost));
  G2S_GPU_TRY(hipMemcpy(g.id2rank.data(), d_i2r.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  G2S_GPU_TRY(hipMemcpy(g.ustart.data(), (uint64_t*)d_us.p + kUstartPad, words * 8, hipMemcpyDeviceToHost));
  if (next_id >= n) {
    G2S_GPU_TRY(hipMemcpy(g.rank2id.data(), d_id.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    G2S_GPU_TRY(hipMemcpy(g.flip.data(), d_flip.p, (size_t)n, hipMemcpyDeviceToHost));
  }
  // the device copy of the graph for the fill path
  DeviceGraph dg;
  dg.succ = (uint32_t*)d_sid.release();
  dg.ustart = (uint64_t*)d_us.release() + kUstartPad;
  dg.bytes = sbytes + (words + 2 * kUstartPad) * 8;
  g.dev[device] = dg;
  return true;
#undef G2S_GPU_TRY
}

// The sorted set of solid canonical k-mers and its prefix index, from the reads.  One key per
// text position, radix sort (rocPRIM; 128-bit keys as two stable 64-bit passes), run heads,
// runs of at least `solid` copies compacted.  Returns false (g untouched) when the device
// cannot be used or the text does not fit comfortably.
template <class KT>
static bool count_solid_gpu_t(Graph& g, std::vector<KT>& out, const std::vector<std::pair<const char*, uint64_t>>& seqs,
                              int solid, int device, std::string* why) {
#define G2S_GPU_TRY(expr)                                                                 \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) { if (why) *why = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } \
  } while (0)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { if (why) *why = "no device"; return false; }
  G2S_GPU_TRY(hipSetDevice(device));
  const int k = g.k;
  uint64_t T = 0;
  for (auto& sq : seqs) T += sq.second + 1;  // one separator after every sequence
  if (T == 0 || T >= (1ull << 32)) { if (why) *why = "text size"; return false; }
  size_t free_b = 0, total_b = 0;
  G2S_GPU_TRY(hipMemGetInfo(&free_b, &total_b));
  if ((double)T * (double)(4 * sizeof(KT) + 24) > 0.5 * (double)free_b) { if (why) *why = "text too large for the device"; return false; }
  std::vector<uint8_t> text((size_t)T);
  {
    size_t pos = 0;
    for (auto& sq : seqs) { memcpy(text.data() + pos, sq.first, (size_t)sq.second); pos += (size_t)sq.second; text[pos++] = 'N'; }
  }
  Dev d_text, d_keys, d_alt, d_lo, d_hi, d_lo2, d_hi2, d_flag, d_pos, d_hidx, d_keep, d_kpos, d_out, d_tmp, d_misc;
  G2S_GPU_TRY(d_text.alloc((size_t)T));
  G2S_GPU_TRY(d_keys.alloc((size_t)T * sizeof(KT)));
  G2S_GPU_TRY(d_misc.alloc(16));
  G2S_GPU_TRY(hipMemcpy(d_text.p, text.data(), (size_t)T, hipMemcpyHostToDevice));
  const dim3 blk(256), grdT((unsigned)((T + 255) / 256));
  hipLaunchKernelGGL(k_extract<KT>, grdT, blk, 0, 0, (const uint8_t*)d_text.p, T, k, (KT*)d_keys.p);
  (void)hipFree(d_text.release());
  KT* sorted = nullptr;
  if constexpr (sizeof(KT) == 8) {
    G2S_GPU_TRY(d_alt.alloc((size_t)T * 8));
    size_t tb = 0;
    G2S_GPU_TRY(rocprim::radix_sort_keys(nullptr, tb, (uint64_t*)d_keys.p, (uint64_t*)d_alt.p, (size_t)T, 0, 64));
    G2S_GPU_TRY(d_tmp.alloc(tb));
    // (the filler is all-ones: sort on all 64 bits so that it ends up last)
    G2S_GPU_TRY(rocprim::radix_sort_keys(d_tmp.p, tb, (uint64_t*)d_keys.p, (uint64_t*)d_alt.p, (size_t)T, 0, 64));
    sorted = (KT*)d_alt.p;
  } else {
    G2S_GPU_TRY(d_lo.alloc((size_t)T * 8));
    G2S_GPU_TRY(d_hi.alloc((size_t)T * 8));
    G2S_GPU_TRY(d_lo2.alloc((size_t)T * 8));
    G2S_GPU_TRY(d_hi2.alloc((size_t)T * 8));
    hipLaunchKernelGGL(k_split<KT>, grdT, blk, 0, 0, (const KT*)d_keys.p, T, (uint64_t*)d_lo.p, (uint64_t*)d_hi.p);
    size_t tb = 0;
    G2S_GPU_TRY(rocprim::radix_sort_pairs(nullptr, tb, (uint64_t*)d_lo.p, (uint64_t*)d_lo2.p, (uint64_t*)d_hi.p,
                                          (uint64_t*)d_hi2.p, (size_t)T, 0, 64));
    G2S_GPU_TRY(d_tmp.alloc(tb));
    // least significant word first, then a stable pass on the most significant word
    G2S_GPU_TRY(rocprim::radix_sort_pairs(d_tmp.p, tb, (uint64_t*)d_lo.p, (uint64_t*)d_lo2.p, (uint64_t*)d_hi.p,
                                          (uint64_t*)d_hi2.p, (size_t)T, 0, 64));
    G2S_GPU_TRY(rocprim::radix_sort_pairs(d_tmp.p, tb, (uint64_t*)d_hi2.p, (uint64_t*)d_hi.p, (uint64_t*)d_lo2.p,
                                          (uint64_t*)d_lo.p, (size_t)T, 0, 64));
    hipLaunchKernelGGL(k_join, grdT, blk, 0, 0, (const uint64_t*)d_lo.p, (const uint64_t*)d_hi.p, T, (u128*)d_keys.p);
    sorted = (KT*)d_keys.p;
    for (Dev* d : {&d_lo2, &d_hi2}) (void)hipFree(d->release());
  }
  // ---- runs of equal keys -> the k-mers seen at least `solid` times
  G2S_GPU_TRY(d_flag.alloc((size_t)T * 4));
  G2S_GPU_TRY(d_pos.alloc((size_t)T * 4));
  hipLaunchKernelGGL(k_heads<KT>, grdT, blk, 0, 0, (const KT*)sorted, T, (uint32_t*)d_flag.p);
  size_t tb2 = 0;
  G2S_GPU_TRY(rocprim::exclusive_scan(nullptr, tb2, (const uint32_t*)d_flag.p, (uint32_t*)d_pos.p, 0u, (size_t)T,
                                      rocprim::plus<uint32_t>()));
  Dev d_tmp2;
  G2S_GPU_TRY(d_tmp2.alloc(tb2));
  G2S_GPU_TRY(rocprim::exclusive_scan(d_tmp2.p, tb2, (const uint32_t*)d_flag.p, (uint32_t*)d_pos.p, 0u, (size_t)T,
                                      rocprim::plus<uint32_t>()));
  uint32_t lastf = 0, lastp = 0;
  G2S_GPU_TRY(hipMemcpy(&lastf, (const uint32_t*)d_flag.p + (T - 1), 4, hipMemcpyDeviceToHost));
  G2S_GPU_TRY(hipMemcpy(&lastp, (const uint32_t*)d_pos.p + (T - 1), 4, hipMemcpyDeviceToHost));
  const uint32_t nheads = lastp + lastf;
  uint32_t n_solid = 0;
  if (nheads) {
    G2S_GPU_TRY(d_hidx.alloc((size_t)nheads * 4));
    G2S_GPU_TRY(d_keep.alloc((size_t)nheads * 4));
    G2S_GPU_TRY(d_kpos.alloc((size_t)nheads * 4));
    G2S_GPU_TRY(hipMemset(d_misc.p, 0, 16));
    hipLaunchKernelGGL(k_head_index<KT>, grdT, blk, 0, 0, (const KT*)sorted, (const uint32_t*)d_flag.p,
                       (const uint32_t*)d_pos.p, T, (uint32_t*)d_hidx.p, (uint32_t*)d_misc.p);
    uint32_t n_valid = 0;
    G2S_GPU_TRY(hipMemcpy(&n_valid, d_misc.p, 4, hipMemcpyDeviceToHost));
    const dim3 grdH((nheads + 255) / 256);
    hipLaunchKernelGGL(k_solid_flag, grdH, blk, 0, 0, (const uint32_t*)d_hidx.p, nheads, n_valid, (uint32_t)std::max(1, solid),
                       (uint32_t*)d_keep.p);
    size_t tb3 = 0;
    G2S_GPU_TRY(rocprim::exclusive_scan(nullptr, tb3, (const uint32_t*)d_keep.p, (uint32_t*)d_kpos.p, 0u, (size_t)nheads,
                                        rocprim::plus<uint32_t>()));
    Dev d_tmp3;
    G2S_GPU_TRY(d_tmp3.alloc(tb3));
    G2S_GPU_TRY(rocprim::exclusive_scan(d_tmp3.p, tb3, (const uint32_t*)d_keep.p, (uint32_t*)d_kpos.p, 0u, (size_t)nheads,
                                        rocprim::plus<uint32_t>()));
    uint32_t lk = 0, lkp = 0;
    G2S_GPU_TRY(hipMemcpy(&lk, (const uint32_t*)d_keep.p + (nheads - 1), 4, hipMemcpyDeviceToHost));
    G2S_GPU_TRY(hipMemcpy(&lkp, (const uint32_t*)d_kpos.p + (nheads - 1), 4, hipMemcpyDeviceToHost));
    n_solid = lk + lkp;
    if (n_solid) {
      G2S_GPU_TRY(d_out.alloc((size_t)n_solid * sizeof(KT)));
      hipLaunchKernelGGL(k_compact<KT>, grdH, blk, 0, 0, (const KT*)sorted, (const uint32_t*)d_hidx.p,
                         (const uint32_t*)d_keep.p, (const uint32_t*)d_kpos.p, nheads, (KT*)d_out.p);
    }
  }
  G2S_GPU_TRY(hipGetLastError());
  // ---- prefix index and the copies the host keeps (node look-ups, node strings)
  const int bits = std::min(2 * k, 22);
  const uint32_t nb = 1u << bits;
  std::vector<uint32_t> bucket((size_t)nb + 1, 0);
  std::vector<KT> host((size_t)n_solid);
  if (n_solid) {
    Dev d_bucket;
    G2S_GPU_TRY(d_bucket.alloc(((size_t)nb + 1) * 4));
    hipLaunchKernelGGL(k_bucket<KT>, dim3((nb + 1 + 255) / 256), blk, 0, 0, (const KT*)d_out.p, n_solid, 2 * k - bits, nb,
                       (uint32_t*)d_bucket.p);
    G2S_GPU_TRY(hipMemcpy(bucket.data(), d_bucket.p, ((size_t)nb + 1) * 4, hipMemcpyDeviceToHost));
    G2S_GPU_TRY(hipMemcpy(host.data(), d_out.p, (size_t)n_solid * sizeof(KT), hipMemcpyDeviceToHost));
  }
  out.swap(host);
  g.n = n_solid;
  g.bucket.swap(bucket);
  g.bucket_bits = bits;
  return true;
#undef G2S_GPU_TRY
}

bool count_solid_gpu(Graph& g, const std::vector<std::pair<const char*, uint64_t>>& seqs, int solid, int device,
                     std::string* why) {
  return g.wide ? count_solid_gpu_t<u128>(g, g.kmers128, seqs, solid, device, why)
                : count_solid_gpu_t<uint64_t>(g, g.kmers64, seqs, solid, device, why);
}

bool graph_finish_gpu(Graph& g, int device, const std::function<void(const std::vector<uint32_t>&, uint32_t)>& host_walk,
                      std::string* why) {
  if ((g.k % 2) == 0) { if (why) *why = "even k"; return false; }
  return g.wide ? finish_gpu_t<u128>(g, g.kmers128, device, host_walk, why)
                : finish_gpu_t<uint64_t>(g, g.kmers64, device, host_walk, why);
}

}  // namespace g2s
