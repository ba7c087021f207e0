// ============================================================================
//  oracle/g2s_oracle.cpp — see g2s_oracle.hpp.  TEST INFRASTRUCTURE ONLY,
//  PARITY UNPINNED BY THE REFERENCE (no reference tests/fixtures exist).
//
//  Shape of the algorithm is deliberately the reference's: hash maps keyed by
//  canonical k-mer, sparse append-only DP rows with a last/second-last fast
//  path, neighbour enumeration by k-mer arithmetic + membership probes, level
//  by level border sets.  That makes it a fair timed CPU baseline ("port").
// ============================================================================
#include "g2s_oracle.hpp"

#include <algorithm>
#include <chrono>
#include <ctime>
#include <cctype>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <unordered_map>

namespace orc {

// ---------------------------------------------------------------------------
// glibc rand(): stdlib/random_r.c TYPE_3.  srand(seed) fills r[0..30] with the
// Park-Miller LCG (16807, Schrage), then discards 310 outputs.
// ---------------------------------------------------------------------------
void GlibcRand::seed(unsigned int s) {
  if (s == 0) s = 1;
  int32_t word = (int32_t)s;
  r_[0] = word;
  for (int i = 1; i < 31; i++) {
    long hi = word / 127773, lo = word % 127773;
    long w = 16807 * lo - 2836 * hi;
    if (w < 0) w += 2147483647;
    word = (int32_t)w;
    r_[i] = word;
  }
  f_ = 3;
  b_ = 0;
  for (int i = 0; i < 310; i++) (void)next();
}

int GlibcRand::next() {
  uint32_t v = (uint32_t)r_[f_] + (uint32_t)r_[b_];
  r_[f_] = (int32_t)v;
  int out = (int)((v >> 1) & 0x7fffffff);
  if (++f_ >= 31) f_ = 0;
  if (++b_ >= 31) b_ = 0;
  return out;
}

// ---------------------------------------------------------------------------
// GATB k-mer codec: code = (c>>1)&3 => A0 C1 T2 G3,
// first base most significant, complement = code^2, canonical = min(fwd, rc),
// invalid char flag = (c>>3)&1 (true for N/n) used only by k-mer counting.
// ---------------------------------------------------------------------------
static inline int nt_code(char c) { return (c >> 1) & 3; }
static inline bool nt_invalid(char c) { return (c >> 3) & 1; }
static const char NT_CHAR[4] = {'A', 'C', 'T', 'G'};

static inline uint64_t rc64_full(uint64_t x) {
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = __builtin_bswap64(x);
  return x ^ 0xAAAAAAAAAAAAAAAAULL;
}
static inline uint64_t kmer_revcomp(uint64_t x, int k) { return rc64_full(x) >> (64 - 2 * k); }
static inline u128 kmer_revcomp(u128 x, int k) {
  u128 y = ((u128)rc64_full((uint64_t)x) << 64) | (u128)rc64_full((uint64_t)(x >> 64));
  return y >> (128 - 2 * k);
}
static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
struct KHash {
  size_t operator()(uint64_t x) const { return (size_t)mix64(x); }
  size_t operator()(u128 x) const { return (size_t)(mix64((uint64_t)x) ^ mix64((uint64_t)(x >> 64) + 0x9e3779b97f4a7c15ULL)); }
};

// Exact membership set (open addressing).  Replaces GATB's Bloom + cFP (D1).
template <class KT>
class KmerSet {
 public:
  void build(const std::vector<KT>& keys) {
    size_t cap = 16;
    while (cap < keys.size() * 2 + 2) cap <<= 1;
    tab_.assign(cap, ~KT(0));
    mask_ = cap - 1;
    n_ = keys.size();
    for (KT x : keys) {
      size_t h = KHash()(x) & mask_;
      while (tab_[h] != ~KT(0)) h = (h + 1) & mask_;
      tab_[h] = x;
    }
  }
  bool contains(KT x) const {
    size_t h = KHash()(x) & mask_;
    while (true) {
      KT y = tab_[h];
      if (y == x) return true;
      if (y == ~KT(0)) return false;
      h = (h + 1) & mask_;
    }
  }
  size_t size() const { return n_; }
 private:
  std::vector<KT> tab_;
  size_t mask_ = 0, n_ = 0;
};

template <class KT>
struct NodeT {
  KT kmer;          // canonical value
  uint8_t strand;   // 0 FORWARD (sequence == canonical), 1 REVCOMP
};

class GraphBase {
 public:
  virtual ~GraphBase() {}
  int k = 0;
  virtual uint64_t num_kmers() const = 0;
};

// GATB Graph primitives used by the hot path (Appendix B.2-B.4).
template <class KT>
class OGraph : public GraphBase {
 public:
  typedef NodeT<KT> Node;
  KT mask;
  KmerSet<KT> set;

  void init(int kk) {
    k = kk;
    mask = (2 * kk == (int)sizeof(KT) * 8) ? ~KT(0) : ((KT(1) << (2 * kk)) - 1);
  }
  uint64_t num_kmers() const override { return set.size(); }

  // buildNode(const char*): reads the first k chars.
  Node buildNode(const char* s) const {
    KT f = 0;
    for (int i = 0; i < k; i++) f = (f << 2) | (KT)nt_code(s[i]);
    KT r = kmer_revcomp(f, k);
    Node n;
    if (f < r) { n.kmer = f; n.strand = 0; } else { n.kmer = r; n.strand = 1; }
    return n;
  }
  bool contains(const Node& n) const { return set.contains(n.kmer); }
  std::string toString(const Node& n) const {
    KT seq = n.strand == 0 ? n.kmer : kmer_revcomp(n.kmer, k);
    std::string s(k, 'A');
    for (int i = k - 1; i >= 0; i--) { s[i] = NT_CHAR[(int)(seq & 3)]; seq >>= 2; }
    return s;
  }
  // Outgoing neighbours in GATB order: append A, C, T, G.
  int successors(const Node& n, Node out[4]) const {
    KT rcv = kmer_revcomp(n.kmer, k);
    KT seq = n.strand == 0 ? n.kmer : rcv;
    KT rseq = n.strand == 0 ? rcv : n.kmer;
    int c = 0;
    for (int nt = 0; nt < 4; nt++) {
      KT y = ((seq << 2) | (KT)nt) & mask;
      KT ry = (rseq >> 2) | ((KT)(nt ^ 2) << (2 * (k - 1)));
      Node m;
      if (y < ry) { m.kmer = y; m.strand = 0; } else { m.kmer = ry; m.strand = 1; }
      if (set.contains(m.kmer)) out[c++] = m;
    }
    return c;
  }
  // Incoming neighbours in GATB order: revcomp, append A,C,T,G  => the
  // predecessor sequences are T+X', G+X', A+X', C+X' (X' = X without last base).
  int predecessors(const Node& n, Node out[4]) const {
    KT rcv = kmer_revcomp(n.kmer, k);
    KT seq = n.strand == 0 ? n.kmer : rcv;
    KT rseq = n.strand == 0 ? rcv : n.kmer;
    int c = 0;
    for (int nt = 0; nt < 4; nt++) {
      KT z = ((rseq << 2) | (KT)nt) & mask;                         // revcomp of the predecessor
      KT p = (seq >> 2) | ((KT)(nt ^ 2) << (2 * (k - 1)));          // the predecessor itself
      Node m;
      if (p < z) { m.kmer = p; m.strand = 0; } else { m.kmer = z; m.strand = 1; }
      if (set.contains(m.kmer)) out[c++] = m;
    }
    return c;
  }
};

// ---------------------------------------------------------------------------
// FASTA/FASTQ reading, solid k-mer counting (Appendix B.5-B.6)
// ---------------------------------------------------------------------------
bool read_file(const std::string& path, std::string* out) {
  std::ifstream f(path.c_str(), std::ios::in | std::ios::binary);
  if (!f) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  *out = ss.str();
  return true;
}

void parse_fastx(const std::string& text, std::vector<std::pair<std::string, std::string>>* out) {
  size_t pos = 0, n = text.size();
  auto getline = [&](std::string* line) -> bool {
    if (pos >= n) return false;
    size_t e = text.find('\n', pos);
    if (e == std::string::npos) e = n;
    size_t len = e - pos;
    if (len > 0 && text[pos + len - 1] == '\r') len--;
    line->assign(text, pos, len);
    pos = e + 1;
    return true;
  };
  std::string line;
  bool have = getline(&line);
  while (have) {
    if (line.empty()) { have = getline(&line); continue; }
    if (line[0] == '>') {
      std::string comment = line.substr(1), seq;
      while ((have = getline(&line))) {
        if (!line.empty() && line[0] == '>') break;
        seq += line;
      }
      out->push_back(std::make_pair(comment, seq));
    } else if (line[0] == '@') {
      std::string comment = line.substr(1), seq, plus, qual;
      getline(&seq);
      getline(&plus);
      getline(&qual);
      out->push_back(std::make_pair(comment, seq));
      have = getline(&line);
    } else {
      have = getline(&line);
    }
  }
}

template <class KT>
static GraphBase* build_graph(const std::vector<std::string>& seqs, int k, int solid) {
  OGraph<KT>* G = new OGraph<KT>();
  G->init(k);
  std::vector<KT> all;
  for (const std::string& s : seqs) {
    KT f = 0, r = 0;
    int valid = 0;
    for (size_t i = 0; i < s.size(); i++) {
      char c = s[i];
      if (nt_invalid(c)) { valid = 0; f = 0; r = 0; continue; }
      int code = nt_code(c);
      f = ((f << 2) | (KT)code) & G->mask;
      r = (r >> 2) | ((KT)(code ^ 2) << (2 * (k - 1)));
      if (++valid >= k) all.push_back(f < r ? f : r);
    }
  }
  std::sort(all.begin(), all.end());
  std::vector<KT> keep;
  for (size_t i = 0; i < all.size();) {
    size_t j = i;
    while (j < all.size() && all[j] == all[i]) j++;
    if ((long long)(j - i) >= (long long)solid) keep.push_back(all[i]);
    i = j;
  }
  G->set.build(keep);
  return G;
}

GraphBase* graph_from_seqs(const std::vector<std::string>& seqs, int k, int solid) {
  if (k < 1 || k > 63) return NULL;
  if (k <= 31) return build_graph<uint64_t>(seqs, k, solid);
  return build_graph<u128>(seqs, k, solid);
}

GraphBase* graph_from_files(const std::vector<std::string>& files, int k, int solid) {
  std::vector<std::string> seqs;
  for (const std::string& f : files) {
    std::string text;
    if (!read_file(f, &text)) return NULL;
    std::vector<std::pair<std::string, std::string>> recs;
    parse_fastx(text, &recs);
    for (auto& r : recs) seqs.push_back(std::move(r.second));
  }
  return graph_from_seqs(seqs, k, solid);
}
void graph_free(GraphBase* g) { delete g; }
uint64_t graph_num_kmers(const GraphBase* g) { return g->num_kmers(); }
int graph_k(const GraphBase* g) { return g->k; }

// ---------------------------------------------------------------------------
// DP rows (Gap2Seq.cpp:488-834).  Append-only, sorted by depth, two strands.
// A byte counter stands in for count_allocator (Gap2Seq.cpp:441-485): the real
// accounting is libstdc++-internal (divergence D3), this only keeps the polls.
// ---------------------------------------------------------------------------
struct MemModel {
  long long bytes = 0;
  void vec_grow(size_t old_cap, size_t new_cap, size_t elem) { bytes += (long long)(new_cap - old_cap) * (long long)elem; }
};

// map_element: (depth, count) pairs.
struct Row {
  std::vector<std::pair<int, int>> s[2];
  int get(int strand, int i) const {
    const std::vector<std::pair<int, int>>& v = s[strand];
    size_t n = v.size();
    if (n == 0) return 0;
    if (v[n - 1].first == i) return v[n - 1].second;
    if (n > 1 && v[n - 2].first == i) return v[n - 2].second;
    size_t lo = 0, hi = n;
    while (lo < hi) {
      size_t mid = (lo + hi) / 2;
      if (v[mid].first < i) lo = mid + 1; else hi = mid;
    }
    return (lo < n && v[lo].first == i) ? v[lo].second : 0;
  }
  // only "update last" or "append larger depth" (Gap2Seq.cpp:584-608)
  void set(int strand, int i, int val, MemModel& mm, uint64_t* new_states) {
    std::vector<std::pair<int, int>>& v = s[strand];
    if (!v.empty() && v.back().first == i) { v.back().second = val; return; }
    size_t c0 = v.capacity();
    v.push_back(std::make_pair(i, val));
    if (v.capacity() != c0) mm.vec_grow(c0, v.capacity(), sizeof(std::pair<int, int>));
    (*new_states)++;
  }
};

// map_element2: depths only.
struct Row2 {
  std::vector<int> s[2];
  void set(int strand, int i, MemModel& mm, uint64_t* new_states) {
    std::vector<int>& v = s[strand];
    if (!v.empty() && v.back() == i) return;
    size_t c0 = v.capacity();
    v.push_back(i);
    if (v.capacity() != c0) mm.vec_grow(c0, v.capacity(), sizeof(int));
    (*new_states)++;
  }
};

static const long long MAP_NODE_BYTES = 64;     // model: hash node + element header
static const long long BORDER_NODE_BYTES = 40;  // model: unordered_set node

// Border set keyed by canonical k-mer only (GATB Node::operator== ignores the
// strand).  First insertion wins; a second strand of the same k-mer in one
// level is the irreproducible corner Q7 -> flagged.
template <class KT>
struct Border {
  std::vector<NodeT<KT>> items;
  std::unordered_map<KT, uint8_t, KHash> seen;
  bool insert(const NodeT<KT>& n, MemModel& mm, int* q7) {
    auto it = seen.find(n.kmer);
    if (it != seen.end()) {
      if (it->second != n.strand) *q7 = 1;
      return false;
    }
    seen.emplace(n.kmer, n.strand);
    items.push_back(n);
    mm.bytes += BORDER_NODE_BYTES;
    return true;
  }
  void clear(MemModel& mm) {
    mm.bytes -= BORDER_NODE_BYTES * (long long)items.size();
    items.clear();
    seen.clear();
  }
  void swap(Border& o) { items.swap(o.items); seen.swap(o.seen); }
};

static inline int sat_add(int a, int b) {
  long long s = (long long)a + (long long)b;
  return s > MAX_PATHS ? MAX_PATHS : (int)s;
}

// Stand-in for boost::adjacency_list<vecS, vecS, bidirectionalS> as used at
// Gap2Seq.cpp:1177-1434: parallel edges allowed, edge(u,v) is a linear scan.
struct MiniDigraph {
  std::vector<std::vector<int>> out, in;
  size_t nedges = 0;
  int add_vertex() { out.emplace_back(); in.emplace_back(); return (int)out.size() - 1; }
  size_t num_vertices() const { return out.size(); }
  bool has_edge(int u, int v) const {
    for (int t : out[u]) if (t == v) return true;
    return false;
  }
  void add_edge(int u, int v) { out[u].push_back(v); in[v].push_back(u); nedges++; }
  static void erase_one(std::vector<int>& v, int x) {
    for (size_t i = 0; i < v.size(); i++) if (v[i] == x) { v.erase(v.begin() + i); return; }
  }
  void clear_vertex(int v) {
    for (int t : out[v]) { erase_one(in[t], v); nedges--; }  // t == v: drops the self loop from in[v]
    out[v].clear();
    for (int s : in[v]) { erase_one(out[s], v); nedges--; }  // only s != v is left here
    in[v].clear();
  }
};

// Tarjan SCC, iterative.  Returns number of components, comp[v] in [0,nc).
static size_t tarjan_scc(const MiniDigraph& g, std::vector<size_t>* comp) {
  const int n = (int)g.num_vertices();
  std::vector<int> index(n, -1), low(n, 0), stack, itpos(n, 0), call;
  std::vector<char> onstack(n, 0);
  comp->assign(n, 0);
  int idx = 0;
  size_t nc = 0;
  for (int root = 0; root < n; root++) {
    if (index[root] != -1) continue;
    call.push_back(root);
    index[root] = low[root] = idx++;
    stack.push_back(root);
    onstack[root] = 1;
    while (!call.empty()) {
      int v = call.back();
      if (itpos[v] < (int)g.out[v].size()) {
        int w = g.out[v][itpos[v]++];
        if (index[w] == -1) {
          index[w] = low[w] = idx++;
          stack.push_back(w);
          onstack[w] = 1;
          call.push_back(w);
        } else if (onstack[w]) {
          low[v] = std::min(low[v], index[w]);
        }
      } else {
        call.pop_back();
        if (!call.empty()) { int u = call.back(); low[u] = std::min(low[u], low[v]); }
        if (low[v] == index[v]) {
          while (true) {
            int w = stack.back();
            stack.pop_back();
            onstack[w] = 0;
            (*comp)[w] = nc;
            if (w == v) break;
          }
          nc++;
        }
      }
    }
  }
  return nc;
}

// ---------------------------------------------------------------------------
// fill_gap  (Gap2Seq.cpp:858-1556)
// ---------------------------------------------------------------------------
template <class KT>
static int fill_gap_t(const OGraph<KT>& G, GlibcRand& rng, const std::string& kmer_left,
                      const std::string& kmer_right, int gap_len, int k, int gap_err, int lmf, int rmf,
                      int* left_fuz, int* right_fuz, long long max_mem, char* fill, bool skip_confident,
                      bool all_paths, SubgraphStats* substats, FillInfo* info, std::string* extra_log) {
  typedef NodeT<KT> Node;
  FillInfo dummy;
  if (!info) info = &dummy;
  // D2: the reference would throw std::out_of_range from substr.
  if (lmf < 0 || rmf < 0 || (int)kmer_left.size() < k + lmf || (int)kmer_right.size() < k + rmf) return 0;

  // :862-863  ceilf/floorf of (g+e)/2.f
  const int right_half = rmf + (gap_len + gap_err + 1) / 2;
  const int left_half = lmf + (gap_len + gap_err) / 2;

  MemModel mm;  // :865-869 memuse[id] = 0
  Border<KT> border, nextBorder;
  std::unordered_map<KT, Row2*, KHash> reachRight;
  Node nb[4];

  // ---- Phase A: right BFS (:871-982) -------------------------------------
  int currentD = 0;
  {
    Node node = G.buildNode(kmer_right.c_str() + (kmer_right.size() - k));
    if (G.contains(node)) {
      border.insert(node, mm, &info->q7);
      Row2*& me = reachRight[node.kmer];
      if (!me) { me = new Row2(); mm.bytes += MAP_NODE_BYTES; }
      me->set(node.strand, currentD, mm, &info->ctr.sA);
    }
  }
  long long mymemuse = mm.bytes;
  while (currentD < right_half && mymemuse < max_mem) {
    currentD++;
    info->max_border_a = std::max(info->max_border_a, (int)border.items.size());
    for (size_t bi = 0; bi < border.items.size(); bi++) {
      mymemuse = mm.bytes;
      if (mymemuse >= max_mem) break;
      const Node n = border.items[bi];
      info->ctr.xA++;
      int cnt = G.predecessors(n, nb);
      for (int i = 0; i < cnt; i++) {
        Row2*& me = reachRight[nb[i].kmer];
        if (!me) { me = new Row2(); mm.bytes += MAP_NODE_BYTES; }
        me->set(nb[i].strand, currentD, mm, &info->ctr.sA);
        nextBorder.insert(nb[i], mm, &info->q7);
      }
    }
    border.clear(mm);
    border.swap(nextBorder);
    if (currentD <= rmf) {  // :953 next right-flank seed
      Node node = G.buildNode(kmer_right.c_str() + (kmer_right.size() - k - currentD));
      if (G.contains(node)) {
        border.insert(node, mm, &info->q7);
        Row2*& me = reachRight[node.kmer];
        if (!me) { me = new Row2(); mm.bytes += MAP_NODE_BYTES; }
        me->set(node.strand, currentD, mm, &info->ctr.sA);
      }
    }
    mymemuse = mm.bytes;
  }

  // ---- Phase B: left DP (:984-1167) + Phase C target check (:1107-1159) ---
  border.clear(mm);
  nextBorder.clear(mm);
  int count = 0;
  std::unordered_map<KT, Row*, KHash> reachLeft;
  {
    Node node = G.buildNode(kmer_left.c_str());
    currentD = 0;
    if (G.contains(node)) {
      border.insert(node, mm, &info->q7);
      Row*& me = reachLeft[node.kmer];
      if (!me) { me = new Row(); mm.bytes += MAP_NODE_BYTES; }
      me->set(node.strand, currentD, 1, mm, &info->ctr.sB);
    }
  }
  currentD++;
  mymemuse = mm.bytes;

  Node reachedTarget;
  reachedTarget.kmer = 0;
  reachedTarget.strand = 0;
  int reachedFuz = 0;
  std::vector<int> pathLengths;
  const int prune_from = gap_len / 2 + gap_err / 2 + lmf;  // :1050 (Q1: two int divisions)

  while (currentD <= right_half + left_half && mymemuse < max_mem) {
    info->max_border_b = std::max(info->max_border_b, (int)border.items.size());
    for (size_t bi = 0; bi < border.items.size(); bi++) {
      mymemuse = mm.bytes;
      if (mymemuse >= max_mem) break;
      const Node n = border.items[bi];
      info->ctr.xB++;
      int cnt = G.successors(n, nb);
      Row* node_me = reachLeft[n.kmer];
      const int num_paths = node_me->get(n.strand, currentD - 1);
      for (int i = 0; i < cnt; i++) {
        if (currentD < prune_from || reachRight.find(nb[i].kmer) != reachRight.end()) {  // Q2
          Row*& me = reachLeft[nb[i].kmer];
          if (!me) { me = new Row(); mm.bytes += MAP_NODE_BYTES; }
          me->set(nb[i].strand, currentD, sat_add(me->get(nb[i].strand, currentD), num_paths), mm, &info->ctr.sB);
          nextBorder.insert(nb[i], mm, &info->q7);
        }
      }
    }
    mymemuse = mm.bytes;
    if (mymemuse >= max_mem) break;
    border.clear(mm);
    border.swap(nextBorder);

    if (currentD <= lmf) {  // :1082 next left-flank seed; row value ASSIGNED 1 (Q6)
      Node node = G.buildNode(kmer_left.c_str() + currentD);
      if (G.contains(node)) {
        border.insert(node, mm, &info->q7);
        Row*& me = reachLeft[node.kmer];
        if (!me) { me = new Row(); mm.bytes += MAP_NODE_BYTES; }
        me->set(node.strand, currentD, 1, mm, &info->ctr.sB);
      }
    }

    if (pathLengths.empty() && currentD >= gap_len + lmf + rmf) {  // :1108
      const int err = currentD - gap_len - (lmf + rmf);
      for (int j = 0; j <= rmf && count == 0; j++) {
        reachedTarget = G.buildNode(kmer_right.c_str() + j);
        auto it = reachLeft.find(reachedTarget.kmer);
        if (it == reachLeft.end()) continue;
        const Row* right = it->second;
        const int len1 = gap_len + lmf + j + err;
        const int len2 = gap_len + lmf + j - err;
        reachedFuz = j;
        int v1 = right->get(reachedTarget.strand, len1);
        if (v1 >= 1) { count = sat_add(count, v1); pathLengths.push_back(len1); }
        if (len2 != len1 && len2 >= 0) {
          int v2 = right->get(reachedTarget.strand, len2);
          if (v2 >= 1) { count = sat_add(count, v2); pathLengths.push_back(len2); }
        }
      }
      if (!all_paths && !pathLengths.empty()) break;
    }
    currentD++;
    mymemuse = mm.bytes;
  }
  info->phaseC_count = count;
  info->final_d = currentD;
  if (info->dump_states) {
    std::ostringstream ds;
    for (auto& kv : reachLeft)
      for (int st = 0; st < 2; st++) {
        Node nn; nn.kmer = kv.first; nn.strand = (uint8_t)st;
        for (auto& pr : kv.second->s[st]) ds << G.toString(nn) << " " << pr.first << " " << pr.second << "\n";
      }
    *info->dump_states = ds.str();
  }
  info->n_lengths = (int)pathLengths.size();
  for (size_t i = 0; i < pathLengths.size() && i < 2; i++) info->lengths[i] = pathLengths[i];
  info->reached_fuz = reachedFuz;

  auto free_all = [&]() {
    for (auto& kv : reachLeft) delete kv.second;
    for (auto& kv : reachRight) delete kv.second;
  };

  // ---- Phase D (:1169-1522) ----------------------------------------------
  if (count > 0 && !pathLengths.empty() && fill != NULL) {
    *right_fuz = reachedFuz;
    int currentD2;
    Node current;
    MiniDigraph sub;
    std::unordered_map<KT, int, KHash> node2v;
    std::vector<int> branch;

    if (!skip_confident) {
      Border<KT> back, nextBack;
      MemModel mm2;  // these sets use std allocators' bytes too; not polled
      const int sink = sub.add_vertex();    // vertex 0
      const int source = sub.add_vertex();  // vertex 1
      currentD2 = lmf + gap_len + gap_err + rmf;
      if (all_paths) count = 0;

      auto vertex_of = [&](KT km) -> int {
        auto it = node2v.find(km);
        if (it != node2v.end()) return it->second;
        int v = sub.add_vertex();
        node2v.emplace(km, v);
        return v;
      };

      while (currentD2 >= 0) {
        if (all_paths) {
          if (currentD2 >= lmf + gap_len - gap_err) {
            for (int j = 0; j < rmf; j++) {  // strictly < rmf  (Q3/Q4)
              Node rnode = G.buildNode(kmer_right.c_str() + j);
              if (j < rmf - 1) {
                if (G.contains(rnode)) continue;  // :1201-1206
              }
              auto it = reachLeft.find(rnode.kmer);
              if (it == reachLeft.end()) continue;
              int v = it->second->get(rnode.strand, currentD2);
              if (v >= 1) {
                count = sat_add(count, v);
                if (back.insert(rnode, mm2, &info->q7)) info->ctr.sD++;
                int bv = vertex_of(rnode.kmer);
                if (!sub.has_edge(bv, sink)) sub.add_edge(bv, sink);
              }
            }
          }
        } else {
          for (size_t j = 0; j < pathLengths.size(); j++) {
            if (pathLengths[j] == currentD2) {
              if (back.insert(reachedTarget, mm2, &info->q7)) info->ctr.sD++;
              int bv = vertex_of(reachedTarget.kmer);
              if (!sub.has_edge(bv, sink)) sub.add_edge(bv, sink);
            }
          }
        }

        Node lnode;
        lnode.kmer = 0; lnode.strand = 0;
        bool have_l = false;
        if (currentD2 <= lmf) { lnode = G.buildNode(kmer_left.c_str() + currentD2); have_l = true; }

        for (size_t bi = 0; bi < back.items.size(); bi++) {
          current = back.items[bi];
          info->ctr.xD++;
          // :1270  Node != compares the k-mer value only.  An uninitialised
          // lnode (currentD2 > lmf) is never consulted thanks to the first test.
          if (currentD2 > lmf || !(have_l && current.kmer == lnode.kmer)) {
            int cnt = G.predecessors(current, nb);
            for (int i = 0; i < cnt; i++) {
              auto it = reachLeft.find(nb[i].kmer);
              if (it == reachLeft.end()) continue;
              if (it->second->get(nb[i].strand, currentD2 - 1) > 0) {
                if (nextBack.insert(nb[i], mm2, &info->q7)) info->ctr.sD++;
                int pv = vertex_of(nb[i].kmer);
                int cv = vertex_of(current.kmer);
                if (!sub.has_edge(pv, cv)) sub.add_edge(pv, cv);
              }
            }
          } else {
            int cv = vertex_of(current.kmer);
            if (!sub.has_edge(source, cv)) sub.add_edge(source, cv);
          }
        }
        back.clear(mm2);
        back.swap(nextBack);
        currentD2--;
      }

      // D2: SCC contraction (:1314-1383)
      std::vector<size_t> comp;
      const size_t num_components = tarjan_scc(sub, &comp);
      const size_t num_real_vertices = sub.num_vertices();
      size_t num_real_edges = sub.nedges;
      size_t num_nontrivial = 0, size_nontrivial = 0;
      std::vector<int> csize(num_components, 0);
      if (num_components != num_real_vertices) {
        for (size_t i = 0; i < num_real_vertices; i++) csize[comp[i]]++;
        std::vector<int> cnode(num_components, -1);
        for (size_t i = 0; i < num_components; i++)
          if (csize[i] > 1) { cnode[i] = sub.add_vertex(); num_nontrivial++; }
        for (size_t i = 0; i < num_real_vertices; i++) {
          if (csize[comp[i]] <= 1) continue;
          size_nontrivial++;
          // iterate over snapshots: add_edge only touches cnode / trivial lists,
          // but a trivial vertex's list may be the one iterated through `in`.
          const std::vector<int> ins = sub.in[i];
          for (int s : ins) {
            if (comp[i] == comp[s]) continue;
            if (csize[comp[s]] > 1) {
              if ((size_t)s < i) sub.add_edge(cnode[comp[s]], cnode[comp[i]]);
            } else {
              sub.add_edge(s, cnode[comp[i]]);
            }
          }
          const std::vector<int> outs = sub.out[i];
          for (int t : outs) {
            if (comp[i] == comp[t]) continue;
            if (csize[comp[t]] > 1) {
              if ((size_t)t < i) sub.add_edge(cnode[comp[i]], cnode[comp[t]]);
            } else {
              sub.add_edge(cnode[comp[i]], t);
            }
          }
        }
        for (size_t i = 0; i < num_real_vertices; i++)
          if (csize[comp[i]] > 1) sub.clear_vertex((int)i);
      } else {
        for (size_t i = 0; i < num_components; i++) csize[i] = 1;
      }
      // self loops on trivial vertices (:1385-1402)
      for (size_t i = 0; i < num_real_vertices; i++) {
        if (csize[comp[i]] <= 1) {
          size_t loops = 0;
          for (int t : sub.out[i]) if (t == (int)i) loops++;
          for (size_t l = 0; l < loops; l++) {
            MiniDigraph::erase_one(sub.out[i], (int)i);
            MiniDigraph::erase_one(sub.in[i], (int)i);
            sub.nedges--;
          }
          num_real_edges -= loops;
        }
      }
      substats->vertices = num_real_vertices;
      substats->edges = num_real_edges;
      substats->nontrivial_components = num_nontrivial;
      substats->size_nontrivial_components = size_nontrivial;
      substats->vertices_final = sub.num_vertices() - size_nontrivial;
      substats->edges_final = sub.nedges;

      // branch rule (:1411-1434) over a topological order (Kahn; the result
      // is order independent)
      const size_t nv = sub.num_vertices();
      branch.assign(nv, 0);
      std::vector<int> indeg(nv), order;
      for (size_t v = 0; v < nv; v++) indeg[v] = (int)sub.in[v].size();
      for (size_t v = 0; v < nv; v++) if (indeg[v] == 0) order.push_back((int)v);
      for (size_t qi = 0; qi < order.size(); qi++)
        for (int t : sub.out[order[qi]]) if (--indeg[t] == 0) order.push_back(t);
      int branchcount = 1;
      for (int v : order) {
        const int din = (int)sub.in[v].size(), dout = (int)sub.out[v].size();
        if (din >= 1 || dout >= 1) {
          if (din > 1) branchcount -= din - 1;
          branch[v] = branchcount;
          if (dout > 1) branchcount += dout - 1;
        }
      }
    }

    // D3: traceback (:1437-1517)
    currentD2 = pathLengths[rng.next() % pathLengths.size()];
    info->draws++;
    int lastSolid = currentD2;
    current = reachedTarget;
    std::vector<Node> backv;
    fill[currentD2] = '\0';
    while (currentD2 >= 0) {
      std::string str = G.toString(current);
      if (currentD2 <= lmf) {
        Node lnode = G.buildNode(kmer_left.c_str() + currentD2);
        if (lnode.kmer == current.kmer) { *left_fuz = lmf - currentD2; break; }
      }
      if (currentD2 > 0) {
        bool solid = skip_confident;
        if (!skip_confident) {
          auto it = node2v.find(current.kmer);
          int bv = (it == node2v.end()) ? 0 : it->second;  // Q5: default-inserted 0 = sink
          solid = (branch[bv] == 1);
        }
        if (solid) lastSolid = currentD2;
        char c = str[str.size() - 1];
        fill[currentD2 - 1] = (currentD2 > lastSolid - k) ? (char)toupper(c) : (char)tolower(c);
        int cnt = G.predecessors(current, nb);
        for (int i = 0; i < cnt; i++) {
          auto it = reachLeft.find(nb[i].kmer);
          if (it == reachLeft.end()) continue;
          if (it->second->get(nb[i].strand, currentD2 - 1) > 0) backv.push_back(nb[i]);
        }
        if (backv.empty()) {
          if (extra_log) {
            std::ostringstream os;
            os << "Unable to backtrace! " << currentD2 << " " << currentD << " " << G.toString(reachedTarget) << "\n";
            *extra_log += os.str();
          }
          info->backtrace_failed = 1;
          free_all();
          return 0;
        }
        current = backv[rng.next() % backv.size()];
        info->draws++;
      }
      currentD2--;
      backv.clear();
    }
  }

  if (mymemuse > max_mem) { count = -1; info->mem_exceeded = 1; }  // :1525
  free_all();
  return count;
}

int fill_gap(const GraphBase* g, GlibcRand& rng, const std::string& kmer_left, const std::string& kmer_right,
             int gap_len, int k, int gap_err, int left_max_fuz, int right_max_fuz, int* left_fuz, int* right_fuz,
             long long max_mem, char* fill, bool skip_confident, bool all_paths, SubgraphStats* substats,
             FillInfo* info) {
  SubgraphStats local;
  if (!substats) substats = &local;
  std::string extra;
  int r;
  if (g->k <= 31)
    r = fill_gap_t<uint64_t>(*static_cast<const OGraph<uint64_t>*>(g), rng, kmer_left, kmer_right, gap_len, k, gap_err,
                             left_max_fuz, right_max_fuz, left_fuz, right_fuz, max_mem, fill, skip_confident,
                             all_paths, substats, info, &extra);
  else
    r = fill_gap_t<u128>(*static_cast<const OGraph<u128>*>(g), rng, kmer_left, kmer_right, gap_len, k, gap_err,
                         left_max_fuz, right_max_fuz, left_fuz, right_fuz, max_mem, fill, skip_confident, all_paths,
                         substats, info, &extra);
  if (info) info->sub = *substats;
  if (!extra.empty()) fputs(extra.c_str(), stdout);
  return r;
}

// fill_gap + capture of the "Unable to backtrace!" line for the run log.
static int fill_gap_logged(const GraphBase* g, GlibcRand& rng, const std::string& kl, const std::string& kr, int gap_len,
                           int k, int gap_err, int lmf, int rmf, int* lf, int* rf, long long max_mem, char* fill,
                           bool skip_confident, bool all_paths, SubgraphStats* ss, FillInfo* info, std::string* log) {
  if (g->k <= 31)
    return fill_gap_t<uint64_t>(*static_cast<const OGraph<uint64_t>*>(g), rng, kl, kr, gap_len, k, gap_err, lmf, rmf, lf,
                                rf, max_mem, fill, skip_confident, all_paths, ss, info, log);
  return fill_gap_t<u128>(*static_cast<const OGraph<u128>*>(g), rng, kl, kr, gap_len, k, gap_err, lmf, rmf, lf, rf,
                          max_mem, fill, skip_confident, all_paths, ss, info, log);
}

// ---------------------------------------------------------------------------
// print_statistics (Gap2Seq.cpp:100-156)
// ---------------------------------------------------------------------------
static void print_statistics(std::ostringstream& os, int filledStart, int gapStart, int gapEnd, int paths,
                             const char* buf, int k, int lmf, int rmf, int left_fuz, int right_fuz,
                             bool skip_confident, bool unique_paths, const SubgraphStats& st, int gap,
                             const std::string& comment) {
  if (paths > 0 && (!unique_paths || paths == 1)) {
    int filledLen = (int)strlen(&buf[lmf - left_fuz]) - k;
    if (!skip_confident) {
      int lower = 0, upper = 0;
      for (int j = 0; j < filledLen; j++) {
        if (isupper((unsigned char)buf[j + lmf - left_fuz])) upper++; else lower++;
      }
      os << "Scaffold: " << comment << " GapStart: " << gapStart << " GapEnd: " << gapEnd << " GapLength: " << gap
         << " PathsFound: " << paths << " FilledStart: " << filledStart << " FilledEnd: " << filledStart + filledLen
         << " FilledGapLength: " << filledLen << " LeftMaxFuz: " << lmf << " LeftFuz: " << left_fuz
         << " RightMaxFuz: " << rmf << " RightFuz: " << right_fuz << " ConfidentBases: " << upper
         << " TotalBases: " << (upper + lower) << "\n";
      os << "SubgraphStats: Vertices: " << st.vertices << " Edges: " << st.edges
         << " NontrivialStrongComponents: " << st.nontrivial_components
         << " SizeNontrivialStrongComponents: " << st.size_nontrivial_components
         << " VerticesFinal: " << st.vertices_final << " EdgesFinal: " << st.edges_final << "\n";
    } else {
      os << "Scaffold: " << comment << " GapStart: " << gapStart << " GapEnd: " << gapEnd << " GapLength: " << gap
         << " PathsFound: " << paths << " FilledStart: " << filledStart << " FilledEnd: " << filledStart + filledLen
         << " FilledGapLength: " << filledLen << " LeftFuz: " << left_fuz << " RightFuz: " << right_fuz << "\n";
    }
  } else {
    os << "Scaffold: " << comment << " GapStart: " << gapStart << " GapEnd: " << gapEnd << " GapLength: " << gap
       << " PathsFound: 0 FilledStart: 0 FilledEnd: 0 FilledGapLength: 0"
       << " LeftMaxFuz: " << lmf << " LeftFuz: " << left_fuz << " RightMaxFuz: " << rmf << " RightFuz: " << right_fuz;
    if (paths == -1) os << " Memory limit exceeded";
    os << "\n";
  }
}

static void echo_params(std::ostringstream& os, const Params& p, const std::string& reads, const std::string& filled,
                        long long max_mem) {
  // Gap2Seq.cpp:180-191
  os << "k-mer size: " << p.k << "\n";
  os << "Solidity threshold: " << p.solid << "\n";
  os << "Reads file: " << reads << "\n";
  os << "Filled scaffolds file: " << filled << "\n";
  os << "Distance error: " << p.d_err << "\n";
  os << "Max Fuz: " << p.max_fuz << "\n";
  os << "Max memory: " << max_mem << "\n";
  os << "Skip confident: " << (int)p.skip_confident << "\n";
  os << "Unique: " << (int)p.unique_paths << "\n";
  os << "All paths: " << (int)p.all_paths << "\n";
  os << "Random seed: " << p.randseed << "\n";
}

static void add_counters(Counters* a, const Counters& b) {
  a->xA += b.xA; a->sA += b.sA; a->xB += b.xB; a->sB += b.sB; a->xD += b.xD; a->sD += b.sD;
}

// ---------------------------------------------------------------------------
// execute(), scaffold mode (Gap2Seq.cpp:285-438) — always -nb-cores 1 ordering
// ---------------------------------------------------------------------------
int execute_scaffolds(const GraphBase* g, const Params& p, const std::string& reads_label,
                      const std::string& filled_label, const std::string& scaffolds_text, std::string* fasta,
                      std::string* log, ExecSummary* total) {
  std::ostringstream os, fa;
  long long max_mem = (long long)(p.max_mem_gb * 1024 * 1024 * 1024);  // :170
  GlibcRand rng;
  rng.seed(p.randseed > 0 ? (unsigned)p.randseed : (unsigned)time(NULL));  // :178
  echo_params(os, p, reads_label, filled_label, max_mem);
  const int k = p.k, max_fuz = p.max_fuz, d_err = p.d_err;
  max_mem = max_mem / std::max(1, p.nb_cores);  // :302
  os << "Max mem: " << max_mem << "\n";

  std::vector<std::pair<std::string, std::string>> recs;
  parse_fastx(scaffolds_text, &recs);
  int gapcount = 0, filledgapcount = 0;
  ExecSummary local;
  if (!total) total = &local;

  for (auto& rec : recs) {
    const std::string& comment = rec.first;
    const std::string& seq = rec.second;
    std::string filledSeq = "";
    int prevGapEnd = 0;
    size_t i = 0;
    while (i < seq.size()) {
      if (seq[i] == 'N' || seq[i] == 'n') {
        gapcount++;
        int lmf = std::min(((int)i + k) - prevGapEnd, max_fuz);  // :349 (Q9)
        int kmer_start = (int)i - k - lmf;
        int gap = 0;
        while (i < seq.size() && (seq[i] == 'N' || seq[i] == 'n')) { i++; gap++; }
        int rmf = std::min((int)seq.size() - ((int)i + k), max_fuz);  // :360
        bool ok = i + k + rmf <= seq.size();
        if (rmf < 0) ok = false;  // D2 (Q10): reference aborts with std::out_of_range
        for (int j = 0; j < k + rmf && ok; j++)
          if (seq[i + j] == 'N' || seq[i + j] == 'n') ok = false;
        if (kmer_start >= prevGapEnd && ok) {
          int left_fuz = 0, right_fuz = 0;
          std::vector<char> bufv(gap + k + d_err + lmf + rmf + 1 + 2, 0);
          char* buf = bufv.data();
          SubgraphStats st;
          FillInfo info;
          auto t0 = std::chrono::steady_clock::now();
          std::string extra;  // "Unable to backtrace!" is printed from inside fill_gap (:1494)
          int s = fill_gap_logged(g, rng, seq.substr(kmer_start, k + lmf), seq.substr(i, k + rmf), gap, k, d_err, lmf,
                                  rmf, &left_fuz, &right_fuz, max_mem, buf, p.skip_confident, p.all_paths, &st, &info,
                                  &extra);
          total->fill_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          os << extra;
          add_counters(&total->ctr, info.ctr);
          if (info.q7) total->q7_gaps++;
          int filledStart = (int)filledSeq.length() + kmer_start + k + lmf - left_fuz - prevGapEnd;
          int gapStart = kmer_start + k + lmf;
          print_statistics(os, filledStart, gapStart, (int)i, s, buf, k, lmf, rmf, left_fuz, right_fuz,
                           p.skip_confident, p.unique_paths, st, gap, comment);
          if (s > 0 && (!p.unique_paths || s == 1)) {
            filledgapcount++;
            // :396 assignment, not append (Q8)
            filledSeq = seq.substr(prevGapEnd, kmer_start + k + lmf - left_fuz - prevGapEnd) +
                        std::string(&buf[lmf - left_fuz]);
            filledSeq = filledSeq.substr(0, filledSeq.length() - k);
            i += right_fuz;
          } else {
            filledSeq = filledSeq + seq.substr(prevGapEnd, kmer_start + k + lmf + gap - prevGapEnd);
          }
        } else {
          filledSeq = filledSeq + seq.substr(prevGapEnd, kmer_start + k + lmf + gap - prevGapEnd);
        }
        prevGapEnd = (int)i;
      } else {
        i++;
      }
    }
    filledSeq = filledSeq + seq.substr(prevGapEnd, seq.length() - prevGapEnd);
    fa << ">" << comment << "\n" << filledSeq << "\n";  // BankFasta::insert, one line (B.6)
  }
  os << "Filled " << filledgapcount << " gaps out of " << gapcount << "\n";
  total->gaps += gapcount;
  total->filled += filledgapcount;
  if (fasta) *fasta = fa.str();
  if (log) *log = os.str();
  return 0;
}

// execute(), single-gap mode (Gap2Seq.cpp:227-283)
int execute_single(const GraphBase* g, const Params& p, const std::string& reads_label,
                   const std::string& filled_label, const std::string& left_flank, const std::string& right_flank,
                   int length, std::string* fasta, std::string* log) {
  std::ostringstream os, fa;
  long long max_mem = (long long)(p.max_mem_gb * 1024 * 1024 * 1024);
  GlibcRand rng;
  rng.seed(p.randseed > 0 ? (unsigned)p.randseed : (unsigned)time(NULL));
  echo_params(os, p, reads_label, filled_label, max_mem);
  const int k = p.k;
  if ((int)left_flank.length() < k || (int)right_flank.length() < k) {
    fprintf(stderr, "Flanks need to be at least k length\n");
    if (fasta) *fasta = "";
    if (log) *log = os.str();
    return 0;
  }
  int lmf = std::min((int)left_flank.length() - k, p.max_fuz);
  int rmf = std::min((int)right_flank.length() - k, p.max_fuz);
  std::vector<char> bufv(length + k + p.d_err + lmf + rmf + 1 + 2, 0);
  char* buf = bufv.data();
  int left_fuz = 0, right_fuz = 0;
  SubgraphStats st;
  FillInfo info;
  std::string extra;
  int n = fill_gap_logged(g, rng, left_flank, right_flank, length, k, p.d_err, lmf, rmf, &left_fuz, &right_fuz, max_mem,
                          buf, p.skip_confident, p.all_paths, &st, &info, &extra);
  os << extra;
  int filledStart = (int)left_flank.length() - lmf - left_fuz;
  print_statistics(os, filledStart, (int)left_flank.length(), (int)left_flank.length() + length, n, buf, k, lmf, rmf,
                   left_fuz, right_fuz, p.skip_confident, p.unique_paths, st, length, "");
  std::string filledSeq;
  if (n > 0 && (!p.unique_paths || n == 1)) {
    filledSeq = left_flank.substr(0, left_flank.length() - left_fuz) + std::string(&buf[lmf - left_fuz]);
  } else {
    filledSeq = left_flank + std::string((size_t)length, 'N') + right_flank;
  }
  fa << ">" << "" << "\n" << filledSeq << "\n";
  if (fasta) *fasta = fa.str();
  if (log) *log = os.str();
  return 0;
}

}  // namespace orc
