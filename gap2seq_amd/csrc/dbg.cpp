// gap2seq_amd/csrc/dbg.cpp — see dbg.hpp.
#include "dbg.hpp"

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <functional>
#include <thread>

namespace g2s {

// dbg_gpu.hip
bool count_solid_gpu(Graph& g, const std::vector<std::pair<const char*, uint64_t>>& seqs, int solid, int device,
                     std::string* why);
bool graph_finish_gpu(Graph& g, int device, const std::function<void(const std::vector<uint32_t>&, uint32_t)>& host_walk,
                      std::string* why);

namespace {

template <class KT> std::vector<KT>& kmer_vec(Graph& g);
template <> std::vector<uint64_t>& kmer_vec<uint64_t>(Graph& g) { return g.kmers64; }
template <> std::vector<u128>& kmer_vec<u128>(Graph& g) { return g.kmers128; }
template <class KT> const std::vector<KT>& kmer_vec(const Graph& g);
template <> const std::vector<uint64_t>& kmer_vec<uint64_t>(const Graph& g) { return g.kmers64; }
template <> const std::vector<u128>& kmer_vec<u128>(const Graph& g) { return g.kmers128; }

template <class F>
void parallel_for(uint64_t n, int nthreads, F f) {
  if (nthreads <= 1 || n < 4096) { f((uint64_t)0, n, 0); return; }
  std::vector<std::thread> th;
  uint64_t chunk = (n + nthreads - 1) / nthreads;
  for (int t = 0; t < nthreads; t++) {
    uint64_t b = std::min(n, chunk * t), e = std::min(n, chunk * (t + 1));
    if (b >= e) break;
    th.emplace_back([=]() { f(b, e, t); });
  }
  for (auto& x : th) x.join();
}

// sorted rank of a canonical k-mer, or -1
template <class KT>
int64_t rank_of(const Graph& g, KT x) {
  const std::vector<KT>& v = kmer_vec<KT>(g);
  const int shift = 2 * g.k - g.bucket_bits;
  size_t b = (size_t)(x >> shift);
  size_t lo = g.bucket[b], hi = g.bucket[b + 1];
  while (lo < hi) {
    size_t mid = (lo + hi) >> 1;
    if (v[mid] < x) lo = mid + 1; else hi = mid;
  }
  return (lo < g.bucket[b + 1] && v[lo] == x) ? (int64_t)lo : -1;
}

template <class KT>
void build_bucket_index(Graph& g) {
  const std::vector<KT>& v = kmer_vec<KT>(g);
  g.bucket_bits = std::min(2 * g.k, 22);
  const int shift = 2 * g.k - g.bucket_bits;
  const size_t nb = (size_t)1 << g.bucket_bits;
  g.bucket.assign(nb + 1, 0);
  for (size_t i = 0; i < v.size(); i++) g.bucket[(size_t)(v[i] >> shift) + 1]++;
  for (size_t b = 0; b < nb; b++) g.bucket[b + 1] += g.bucket[b];
}

// 1. collect canonical k-mers, 2. sort, 3. keep those seen >= solid times
template <class KT>
void count_solid(Graph& g, const std::vector<std::pair<const char*, uint64_t>>& seqs, int solid, int nthreads) {
  const int k = g.k;
  // chunk long sequences so that threads share one chromosome
  struct Chunk { const char* p; uint64_t len; };
  std::vector<Chunk> chunks;
  const uint64_t kChunk = 1 << 20;
  for (auto& s : seqs) {
    if (s.second < (uint64_t)k) continue;
    for (uint64_t off = 0; off + k <= s.second; off += kChunk) {
      uint64_t len = std::min<uint64_t>(kChunk + k - 1, s.second - off);
      chunks.push_back({s.first + off, len});
    }
  }
  nthreads = std::max(1, nthreads);
  std::vector<std::vector<KT>> local((size_t)nthreads);
  std::atomic<size_t> next(0);
  auto work = [&](int t) {
    std::vector<KT>& out = local[(size_t)t];
    while (true) {
      size_t ci = next.fetch_add(1);
      if (ci >= chunks.size()) break;
      KmerRoller<KT> r(k);
      const char* p = chunks[ci].p;
      for (uint64_t i = 0; i < chunks[ci].len; i++)
        if (r.push(p[i])) out.push_back(r.canonical());
    }
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
  }
  // range partition by the top byte so that partitions sort independently
  const int pshift = std::max(0, 2 * k - 8);
  const size_t P = (size_t)1 << std::min(8, 2 * k);
  std::vector<uint64_t> cnt(P + 1, 0);
  for (auto& v : local) for (KT x : v) cnt[(size_t)(x >> pshift) + 1]++;
  for (size_t b = 0; b < P; b++) cnt[b + 1] += cnt[b];
  std::vector<KT> all((size_t)cnt[P]);
  {
    std::vector<uint64_t> pos(cnt.begin(), cnt.end() - 1);
    for (auto& v : local) {
      for (KT x : v) all[(size_t)pos[(size_t)(x >> pshift)]++] = x;
      std::vector<KT>().swap(v);
    }
  }
  {
    std::atomic<size_t> nb(0);
    auto sorter = [&]() {
      while (true) {
        size_t b = nb.fetch_add(1);
        if (b >= P) break;
        std::sort(all.begin() + (ptrdiff_t)cnt[b], all.begin() + (ptrdiff_t)cnt[b + 1]);
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(sorter);
    sorter();
    for (auto& x : th) x.join();
  }
  std::vector<KT>& keep = kmer_vec<KT>(g);
  keep.clear();
  for (size_t i = 0; i < all.size();) {
    size_t j = i + 1;
    while (j < all.size() && all[j] == all[i]) j++;
    if ((int64_t)(j - i) >= (int64_t)solid) keep.push_back(all[i]);
    i = j;
  }
  keep.shrink_to_fit();
  g.n = keep.size();
}

// successor / predecessor tables in sorted-rank space
template <class KT>
void build_tables_rank(const Graph& g, int nthreads, std::vector<uint32_t>* succ, std::vector<uint32_t>* pred) {
  const std::vector<KT>& v = kmer_vec<KT>(g);
  const int k = g.k;
  const KT mask = KmerOps<KT>::mask(k);
  const bool even = (k % 2) == 0;
  succ->assign((size_t)g.n * 8, kInvalidNode);
  if (even) pred->assign((size_t)g.n * 8, kInvalidNode);
  parallel_for(g.n, nthreads, [&](uint64_t b, uint64_t e, int) {
    for (uint64_t i = b; i < e; i++) {
      const KT c = v[(size_t)i], rc = KmerOps<KT>::revcomp(c, k);
      for (int strand = 0; strand < 2; strand++) {
        if (strand == 0 && c == rc) continue;  // palindrome: only strand 1 exists (tie rule)
        const KT seq = strand == 0 ? c : rc;
        const KT rseq = strand == 0 ? rc : c;
        const size_t base = ((size_t)i * 2 + strand) * 4;
        for (int nt = 0; nt < 4; nt++) {
          KT y = ((seq << 2) | (KT)nt) & mask;
          KT ry = (rseq >> 2) | ((KT)(nt ^ 2) << (2 * (k - 1)));
          KT cy = y < ry ? y : ry;
          int64_t r = rank_of<KT>(g, cy);
          if (r >= 0) (*succ)[base + nt] = (uint32_t)(2 * r + (y < ry ? 0 : 1));
          if (even) {
            KT z = ((rseq << 2) | (KT)nt) & mask;                 // revcomp of the predecessor
            KT p = (seq >> 2) | ((KT)(nt ^ 2) << (2 * (k - 1)));  // the predecessor
            KT cp = p < z ? p : z;
            int64_t rp = rank_of<KT>(g, cp);
            if (rp >= 0) (*pred)[base + nt] = (uint32_t)(2 * rp + (p < z ? 0 : 1));
          }
        }
      }
    }
  });
}

inline int out_degree(const std::vector<uint32_t>& succ, uint32_t v, uint32_t* only) {
  int d = 0;
  for (int nt = 0; nt < 4; nt++) {
    uint32_t w = succ[(size_t)v * 4 + nt];
    if (w != kInvalidNode) { d++; *only = w; }
  }
  return d;
}

// New node indices in unitig order: k-mers of one maximal non-branching path are
// consecutive, in path order.  rank-space tables in, permutation out.
// With `resume` the k-mers already numbered (by unitig_order_gpu) are kept and the walk
// numbers the rest from first_id on.
void unitig_order(const std::vector<uint32_t>& succ, uint64_t n, bool even_k, std::vector<uint32_t>* rank2id,
                  std::vector<uint8_t>* flip, uint64_t* n_unitigs, bool resume = false, uint32_t first_id = 0) {
  if (!resume) {
    rank2id->assign((size_t)n, kInvalidNode);
    flip->assign((size_t)n, 0);
  }
  uint32_t next_id = resume ? first_id : 0;
  uint64_t unitigs = resume ? *n_unitigs : 0;
  // the unique continuation v -> w when the edge is unitig-internal
  auto step = [&](uint32_t v) -> uint32_t {
    uint32_t w = kInvalidNode, back = kInvalidNode;
    if (out_degree(succ, v, &w) != 1) return kInvalidNode;
    if ((w >> 1) == (v >> 1)) return kInvalidNode;              // self loop / hairpin
    if (out_degree(succ, w ^ 1u, &back) != 1) return kInvalidNode;  // in-degree of w
    if (back != (v ^ 1u)) return kInvalidNode;                  // (palindromes, even k)
    return w;
  };
  for (uint64_t r = 0; r < n; r++) {
    if ((*rank2id)[(size_t)r] != kInvalidNode) continue;
    uint32_t start = (uint32_t)(2 * r);
    if (even_k) {
      // palindromic k-mers have only strand 1; their strand-0 row is empty and isolated
      bool empty0 = true;
      for (int nt = 0; nt < 4; nt++) if (succ[(size_t)start * 4 + nt] != kInvalidNode) empty0 = false;
      uint32_t dummy;
      if (empty0 && out_degree(succ, start | 1u, &dummy) > 0) start |= 1u;
    }
    // walk backwards (= forwards on the reverse strand) to the unitig start
    uint32_t v = start ^ 1u;
    while (true) {
      uint32_t w = step(v);
      if (w == kInvalidNode || (w >> 1) == (start >> 1) || (*rank2id)[w >> 1] != kInvalidNode) break;
      v = w;
    }
    v ^= 1u;  // first node of the unitig, forward orientation
    unitigs++;
    while (true) {
      (*rank2id)[v >> 1] = next_id++;
      (*flip)[v >> 1] = (uint8_t)(v & 1u);  // GATB strand of the numbering direction
      uint32_t w = step(v);
      if (w == kInvalidNode || (*rank2id)[w >> 1] != kInvalidNode) break;
      v = w;
    }
  }
  *n_unitigs = unitigs;
}

// unitig-start bitmap from the id-space successor table (same predicate as unitig_order's step)
void build_ustart(Graph& g) {
  const uint64_t n = g.n;
  g.ustart.assign((size_t)((n + 63) / 64 + 1), 0);
  auto only_out = [&](uint32_t v, uint32_t* w) -> int {
    int c = 0;
    for (int nt = 0; nt < 4; nt++) {
      const uint32_t x = g.succ[(size_t)v * 4 + nt];
      if (x != kInvalidNode) { c++; *w = x; }
    }
    return c;
  };
  for (uint64_t i = 0; i < n; i++) {
    bool internal = false;
    if (i > 0) {
      const uint32_t v = (uint32_t)(2 * (i - 1)), want = (uint32_t)(2 * i);
      uint32_t w = kInvalidNode, back = kInvalidNode;
      internal = only_out(v, &w) == 1 && w == want && only_out(want ^ 1u, &back) == 1 && back == (v ^ 1u);
    }
    if (!internal) g.ustart[(size_t)(i >> 6)] |= 1ull << (i & 63);
  }
  for (uint64_t i = n; i < (uint64_t)g.ustart.size() * 64; i++) g.ustart[(size_t)(i >> 6)] |= 1ull << (i & 63);
}

template <class KT>
void finish_graph(Graph& g, int nthreads) {
  const auto f0 = std::chrono::steady_clock::now();
  if (g.bucket.empty()) build_bucket_index<KT>(g);  // (the GPU k-mer set brings its index along)
  // With a GPU (odd k): successor table by binary search, numbering along unitigs by list
  // ranking, tables in id space and the unitig-start bitmap all on the device (dbg_gpu.hip);
  // the host only numbers circular unitigs.  Otherwise, and for even k, the host build below.
  if (!getenv("G2S_HOST_BUILD")) {
    std::string why;
    const int dev = getenv("G2S_DEVICE") ? atoi(getenv("G2S_DEVICE")) : 0;
    const bool ok = graph_finish_gpu(g, dev, [&](const std::vector<uint32_t>& succ_r, uint32_t first_id) {
      unitig_order(succ_r, g.n, false, &g.rank2id, &g.flip, &g.n_unitigs, true, first_id);
    }, &why);
    if (getenv("G2S_DEBUG"))
      fprintf(stderr, "[g2s]   tables, unitig order, id space: %.3f s on the %s%s%s\n",
              std::chrono::duration<double>(std::chrono::steady_clock::now() - f0).count(), ok ? "GPU" : "host next (",
              ok ? "" : why.c_str(), ok ? "" : ")");
    if (ok) return;
  }
  std::vector<uint32_t> succ_r, pred_r;
  build_tables_rank<KT>(g, nthreads, &succ_r, &pred_r);
  const auto f1 = std::chrono::steady_clock::now();
  unitig_order(succ_r, g.n, (g.k % 2) == 0, &g.rank2id, &g.flip, &g.n_unitigs);
  if (getenv("G2S_DEBUG"))
    fprintf(stderr, "[g2s]   index + successor table %.3f s, unitig order %.3f s (host walk)\n",
            std::chrono::duration<double>(f1 - f0).count(),
            std::chrono::duration<double>(std::chrono::steady_clock::now() - f1).count());
  g.id2rank.assign((size_t)g.n, 0);
  for (uint64_t r = 0; r < g.n; r++) g.id2rank[g.rank2id[(size_t)r]] = (uint32_t)r;
  // permute tables into id space
  const std::vector<KT>& v = kmer_vec<KT>(g);
  g.succ.assign((size_t)g.n * 8, kInvalidNode);
  if (!pred_r.empty()) g.pred.assign((size_t)g.n * 8, kInvalidNode);
  g.lastnt.assign((size_t)g.n * 2, 0);
  const int k = g.k;
  parallel_for(g.n, nthreads, [&](uint64_t b, uint64_t e, int) {
    auto remap = [&](uint32_t w) -> uint32_t {  // rank-space (GATB strand) -> id-space (unitig orientation)
      return 2 * g.rank2id[w >> 1] + ((w & 1u) ^ (uint32_t)g.flip[w >> 1]);
    };
    for (uint64_t r = b; r < e; r++) {
      const uint32_t id = g.rank2id[(size_t)r];
      for (int s = 0; s < 2; s++) {
        const size_t row = ((size_t)id * 2 + (size_t)(s ^ g.flip[(size_t)r])) * 4;
        for (int nt = 0; nt < 4; nt++) {
          uint32_t w = succ_r[((size_t)r * 2 + s) * 4 + nt];
          if (w != kInvalidNode) g.succ[row + nt] = remap(w);
          if (!pred_r.empty()) {
            uint32_t p = pred_r[((size_t)r * 2 + s) * 4 + nt];
            if (p != kInvalidNode) g.pred[row + nt] = remap(p);
          }
        }
      }
      const KT c = v[(size_t)r];
      const uint8_t last_fwd = (uint8_t)(c & 3), last_rev = (uint8_t)(((c >> (2 * (k - 1))) & 3) ^ 2);
      g.lastnt[(size_t)id * 2 + (size_t)(0 ^ g.flip[(size_t)r])] = last_fwd;
      g.lastnt[(size_t)id * 2 + (size_t)(1 ^ g.flip[(size_t)r])] = last_rev;
    }
  });
  build_ustart(g);
}

}  // namespace

uint32_t Graph::node_of(const char* s) const {
  if (n == 0) return kInvalidNode;
  int strand = 0;
  int64_t r;
  if (!wide) {
    uint64_t c;
    encode_kmer<uint64_t>(s, k, &c, &strand);
    r = rank_of<uint64_t>(*this, c);
  } else {
    u128 c;
    encode_kmer<u128>(s, k, &c, &strand);
    r = rank_of<u128>(*this, c);
  }
  if (r < 0) return kInvalidNode;
  return 2 * rank2id[(size_t)r] + ((uint32_t)strand ^ (uint32_t)flip[(size_t)r]);
}

std::string Graph::node_string(uint32_t v) const {
  const uint32_t r = id2rank[v >> 1];
  const int strand = (int)((v & 1u) ^ (uint32_t)flip[r]);
  if (!wide) return decode_kmer<uint64_t>(kmers64[r], strand, k);
  return decode_kmer<u128>(kmers128[r], strand, k);
}

Graph* graph_build(const std::vector<std::pair<const char*, uint64_t>>& seqs, int k, int solid, int nthreads,
                   std::string* err) {
  if (k < 1 || k > 63) { if (err) *err = "k must be in [1,63]"; return nullptr; }
  if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
  Graph* g = new Graph();
  g->k = k;
  g->solid = solid;
  g->wide = k >= 32;
  const auto t0 = std::chrono::steady_clock::now();
  // the solid k-mer set: sort on the GPU when there is one (dbg_gpu.hip), host threads otherwise
  bool set_on_gpu = false;
  if (!getenv("G2S_HOST_BUILD")) {
    std::string why;
    set_on_gpu = count_solid_gpu(*g, seqs, solid, getenv("G2S_DEVICE") ? atoi(getenv("G2S_DEVICE")) : 0, &why);
    if (!set_on_gpu && getenv("G2S_DEBUG")) fprintf(stderr, "[g2s]   k-mer set on the host (%s)\n", why.c_str());
  }
  if (!set_on_gpu) { if (!g->wide) count_solid<uint64_t>(*g, seqs, solid, nthreads); else count_solid<u128>(*g, seqs, solid, nthreads); }
  const auto t1 = std::chrono::steady_clock::now();
  if (g->n >= (1ull << 30)) { if (err) *err = "too many k-mers for 32-bit oriented node ids"; delete g; return nullptr; }
  if (!g->wide) finish_graph<uint64_t>(*g, nthreads); else finish_graph<u128>(*g, nthreads);
  if (getenv("G2S_DEBUG"))
    fprintf(stderr, "[g2s] graph build: %llu k-mers; solid k-mer set %.3f s (%s), tables + unitig order %.3f s (%d threads)\n",
            (unsigned long long)g->n, std::chrono::duration<double>(t1 - t0).count(), set_on_gpu ? "GPU sort" : "host",
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count(), nthreads);
  return g;
}

// ---- own cache format ------------------------------------------------------
static const char kMagic[8] = {'G', '2', 'S', 'D', 'B', 'G', '0', '2'};

bool graph_save(const Graph& g, const std::string& path, std::string* err) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { if (err) *err = "cannot open " + path; return false; }
  // (header word 3: bit 0 = explicit predecessor table, bits 8.. = the abundance threshold of the set)
  uint64_t hdr[4] = {(uint64_t)g.k, g.n, g.n_unitigs, (uint64_t)(g.pred.empty() ? 0 : 1) | ((uint64_t)std::max(0, g.solid) << 8)};
  bool ok = fwrite(kMagic, 1, 8, f) == 8 && fwrite(hdr, 8, 4, f) == 4;
  auto put = [&](const void* p, size_t bytes) { if (ok && bytes) ok = fwrite(p, 1, bytes, f) == bytes; };
  if (!g.wide) put(g.kmers64.data(), g.kmers64.size() * 8); else put(g.kmers128.data(), g.kmers128.size() * 16);
  put(g.rank2id.data(), g.rank2id.size() * 4);
  put(g.flip.data(), g.flip.size());
  put(g.succ.data(), g.succ.size() * 4);
  put(g.pred.data(), g.pred.size() * 4);
  put(g.lastnt.data(), g.lastnt.size());
  fclose(f);
  if (!ok && err) *err = "short write to " + path;
  return ok;
}

Graph* graph_load(const std::string& path, std::string* err) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { if (err) *err = "cannot open " + path; return nullptr; }
  char magic[8];
  uint64_t hdr[4];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, kMagic, 8) != 0 || fread(hdr, 8, 4, f) != 4) {
    fclose(f);
    if (err) *err = "not a g2s graph cache: " + path;
    return nullptr;
  }
  // the file is not trusted: every size and index is checked before it is used
  const uint64_t solid_of_cache = hdr[3] >> 8;
  hdr[3] &= 0xFFull;
  if (hdr[0] < 1 || hdr[0] > 63 || hdr[1] >= (1ull << 30) || hdr[2] > hdr[1] || hdr[3] > 1 || solid_of_cache > (1ull << 30)) {
    fclose(f);
    if (err) *err = "corrupt graph cache header: " + path;
    return nullptr;
  }
  {
    const uint64_t per = (hdr[0] >= 32 ? 16u : 8u) + 4u + 1u + 32u + (hdr[3] ? 32u : 0u) + 2u;
    const long here = ftell(f);
    fseek(f, 0, SEEK_END);
    const long end = ftell(f);
    fseek(f, here, SEEK_SET);
    if (here < 0 || end < 0 || (uint64_t)(end - here) != per * hdr[1]) {
      fclose(f);
      if (err) *err = "graph cache has the wrong size for its header: " + path;
      return nullptr;
    }
  }
  Graph* g = new Graph();
  g->k = (int)hdr[0];
  g->solid = (int)solid_of_cache;
  g->n = hdr[1];
  g->n_unitigs = hdr[2];
  g->wide = g->k >= 32;
  bool ok = true;
  auto get = [&](void* p, size_t bytes) { if (ok && bytes) ok = fread(p, 1, bytes, f) == bytes; };
  if (!g->wide) { g->kmers64.resize((size_t)g->n); get(g->kmers64.data(), (size_t)g->n * 8); }
  else { g->kmers128.resize((size_t)g->n); get(g->kmers128.data(), (size_t)g->n * 16); }
  g->rank2id.resize((size_t)g->n); get(g->rank2id.data(), (size_t)g->n * 4);
  g->flip.resize((size_t)g->n); get(g->flip.data(), (size_t)g->n);
  g->succ.resize((size_t)g->n * 8); get(g->succ.data(), (size_t)g->n * 32);
  if (hdr[3]) { g->pred.resize((size_t)g->n * 8); get(g->pred.data(), (size_t)g->n * 32); }
  g->lastnt.resize((size_t)g->n * 2); get(g->lastnt.data(), (size_t)g->n * 2);
  fclose(f);
  if (!ok) { if (err) *err = "truncated graph cache: " + path; delete g; return nullptr; }
  auto corrupt = [&](const char* what) -> Graph* {
    if (err) *err = std::string("corrupt graph cache (") + what + "): " + path;
    delete g;
    return nullptr;
  };
  // sorted k-mer set, rank2id a permutation, neighbour ids in range, codes 0..3
  for (uint64_t r = 1; r < g->n; r++) {
    const bool ascending = g->wide ? g->kmers128[(size_t)r - 1] < g->kmers128[(size_t)r]
                                   : g->kmers64[(size_t)r - 1] < g->kmers64[(size_t)r];
    if (!ascending) return corrupt("k-mers not strictly ascending");
  }
  g->id2rank.assign((size_t)g->n, kInvalidNode);
  for (uint64_t r = 0; r < g->n; r++) {
    const uint32_t id = g->rank2id[(size_t)r];
    if (id >= g->n || g->id2rank[id] != kInvalidNode) return corrupt("rank2id is not a permutation");
    g->id2rank[id] = (uint32_t)r;
  }
  for (uint32_t w : g->succ) if (w != kInvalidNode && w >= 2 * g->n) return corrupt("successor out of range");
  for (uint32_t w : g->pred) if (w != kInvalidNode && w >= 2 * g->n) return corrupt("predecessor out of range");
  for (uint8_t c : g->lastnt) if (c > 3) return corrupt("base code out of range");
  for (uint8_t c : g->flip) if (c > 1) return corrupt("strand flag out of range");
  if (!g->wide) build_bucket_index<uint64_t>(*g); else build_bucket_index<u128>(*g);
  build_ustart(*g);
  return g;
}

}  // namespace g2s
