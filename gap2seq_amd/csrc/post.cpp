// gap2seq_amd/csrc/post.cpp — see post.hpp.
#include "post.hpp"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <unordered_set>

namespace g2s {

static inline int sat_add(int a, int b) {
  long long s = (long long)a + (long long)b;
  return s > (long long)G2S_MAX_PATHS ? G2S_MAX_PATHS : (int)s;
}

void dp_sort_levels(DpView* v) {
  for (int d = 0; d <= v->D; d++) {
    const uint32_t b = v->lvl[d], e = v->lvl[d + 1];
    if (e - b > 1) std::sort(v->states + b, v->states + e);
  }
}

uint32_t dp_find(const DpView& v, int depth, uint32_t node) {
  if (depth < 0 || depth > v.D) return 0;
  uint32_t lo = v.lvl[depth], hi = v.lvl[depth + 1];
  if (hi - lo <= 8) {
    for (uint32_t i = lo; i < hi; i++)
      if ((uint32_t)(v.states[i] >> 32) == node) return (uint32_t)v.states[i];
    return 0;
  }
  const uint32_t end = hi;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if ((uint32_t)(v.states[mid] >> 32) < node) lo = mid + 1; else hi = mid;
  }
  return (lo < end && (uint32_t)(v.states[lo] >> 32) == node) ? (uint32_t)v.states[lo] : 0;
}

namespace {

// A border of the backward sweep: the reference keys it by canonical k-mer
// (Node::operator== ignores the strand), first insertion wins (Q7 otherwise).
struct BackBorder {
  std::vector<uint32_t> items;
  std::unordered_map<uint32_t, uint32_t> seen;  // only used once the level is wide
  bool insert(uint32_t node, uint32_t* flags) {
    const uint32_t idx = node >> 1;
    if (items.size() < 16 && seen.empty()) {
      for (uint32_t x : items)
        if ((x >> 1) == idx) { if (x != node) *flags |= G2S_GAP_Q7; return false; }
      items.push_back(node);
      return true;
    }
    if (seen.empty()) for (uint32_t x : items) seen.emplace(x >> 1, x);
    auto it = seen.find(idx);
    if (it != seen.end()) { if (it->second != node) *flags |= G2S_GAP_Q7; return false; }
    seen.emplace(idx, node);
    items.push_back(node);
    return true;
  }
  void clear() { items.clear(); seen.clear(); }
};

struct EdgeList {
  std::vector<std::pair<int, int>> e;
  std::unordered_set<uint64_t> have;
  void add_once(int u, int v) {  // boost::edge(u,v).second test + add_edge
    const uint64_t key = ((uint64_t)(uint32_t)u << 32) | (uint32_t)v;
    if (have.insert(key).second) e.emplace_back(u, v);
  }
};

// Tarjan over a CSR adjacency; comp ids are arbitrary.
int strong_components(int nv, const std::vector<int>& off, const std::vector<int>& adj, std::vector<int>* comp) {
  std::vector<int> index((size_t)nv, -1), low((size_t)nv, 0), it((size_t)nv, 0), stack, call;
  std::vector<char> on((size_t)nv, 0);
  comp->assign((size_t)nv, -1);
  int counter = 0, nc = 0;
  for (int r = 0; r < nv; r++) {
    if (index[(size_t)r] >= 0) continue;
    call.push_back(r);
    index[(size_t)r] = low[(size_t)r] = counter++;
    stack.push_back(r);
    on[(size_t)r] = 1;
    while (!call.empty()) {
      const int v = call.back();
      if (it[(size_t)v] < off[(size_t)v + 1] - off[(size_t)v]) {
        const int w = adj[(size_t)(off[(size_t)v] + it[(size_t)v]++)];
        if (index[(size_t)w] < 0) {
          index[(size_t)w] = low[(size_t)w] = counter++;
          stack.push_back(w);
          on[(size_t)w] = 1;
          call.push_back(w);
        } else if (on[(size_t)w]) {
          low[(size_t)v] = std::min(low[(size_t)v], index[(size_t)w]);
        }
      } else {
        call.pop_back();
        if (!call.empty()) low[(size_t)call.back()] = std::min(low[(size_t)call.back()], low[(size_t)v]);
        if (low[(size_t)v] == index[(size_t)v]) {
          int w;
          do { w = stack.back(); stack.pop_back(); on[(size_t)w] = 0; (*comp)[(size_t)w] = nc; } while (w != v);
          nc++;
        }
      }
    }
  }
  return nc;
}

}  // namespace

void post_extract(const Graph& g, const FillParams& p, const GapJob& job, const DpView& v, PostPrep* out) {
  const GapOut& go = *v.out;
  out->count = go.c_count;
  out->phase_d = go.c_count > 0 && go.n_len > 0;  // :1169 (fill is never NULL here)
  if (!out->phase_d || p.skip_confident) return;

  const int lmf = job.lmf, rmf = job.rmf, gl = job.g, e = p.d_err;
  const uint32_t* targets = job.targets();
  const uint32_t* lseeds = job.lseeds();
  const uint32_t reached = targets[go.reached_j];
  const int kSink = 0, kSource = 1;
  int nverts = 2;
  auto vertex = [&](uint32_t node) -> int {
    auto it = out->vertex_of.find(node >> 1);
    if (it != out->vertex_of.end()) return it->second;
    out->vertex_of.emplace(node >> 1, nverts);
    return nverts++;
  };
  EdgeList edges;
  BackBorder back, next;
  int count = p.all_paths ? 0 : go.c_count;  // :1189-1191

  for (int d2 = lmf + gl + e + rmf; d2 >= 0; d2--) {
    if (p.all_paths) {
      if (d2 >= lmf + gl - e) {
        for (int j = 0; j < rmf; j++) {  // strictly < rmf: only j = rmf-1 can be a sink (Q3/Q4)
          const uint32_t rnode = targets[j];
          if (j < rmf - 1 && rnode != kInvalidNode) continue;  // graph.contains(rnode) (:1201-1206)
          if (rnode == kInvalidNode) continue;                 // cannot be in reachableSetLeft
          const uint32_t c = dp_find(v, d2, rnode);
          if (c >= 1) {
            count = sat_add(count, (int)c);
            if (back.insert(rnode, &out->flags)) out->sD++;
            edges.add_once(vertex(rnode), kSink);
          }
        }
      }
    } else {
      for (int j = 0; j < go.n_len; j++) {
        if (go.len[j] == d2) {
          if (back.insert(reached, &out->flags)) out->sD++;
          edges.add_once(vertex(reached), kSink);
        }
      }
    }
    const uint32_t lidx = (d2 <= lmf) ? (lseeds[d2] >> 1) : 0xFFFFFFFFu;  // buildNode(kmer_left.substr(d2,k))
    for (size_t bi = 0; bi < back.items.size(); bi++) {
      const uint32_t cur = back.items[bi];
      out->xD++;
      if (d2 > lmf || (cur >> 1) != lidx) {  // :1270, k-mer comparison only
        for (int nt = 0; nt < 4; nt++) {
          const uint32_t pr = g.pred_of(cur, nt);
          if (pr == kInvalidNode) continue;
          if (dp_find(v, d2 - 1, pr) > 0) {
            if (next.insert(pr, &out->flags)) out->sD++;
            const int pv = vertex(pr), cv = vertex(cur);
            edges.add_once(pv, cv);
          }
        }
      } else {
        edges.add_once(kSource, vertex(cur));
      }
    }
    back.clear();
    std::swap(back.items, next.items);
    std::swap(back.seen, next.seen);
  }
  if (p.all_paths) out->count = count;

  // ---- D2: condensation statistics and the branch rule ----------------------
  const int V = nverts;
  const size_t E = edges.e.size();
  std::vector<int> off((size_t)V + 1, 0), adj(E);
  for (auto& ed : edges.e) off[(size_t)ed.first + 1]++;
  for (int i = 0; i < V; i++) off[(size_t)i + 1] += off[(size_t)i];
  {
    std::vector<int> pos(off.begin(), off.end() - 1);
    for (auto& ed : edges.e) adj[(size_t)pos[(size_t)ed.first]++] = ed.second;
  }
  std::vector<int> comp;
  const int nc = strong_components(V, off, adj, &comp);
  std::vector<int> csize((size_t)nc, 0), cvert((size_t)nc, -1);
  for (int i = 0; i < V; i++) csize[(size_t)comp[(size_t)i]]++;
  int nontrivial = 0, size_nontrivial = 0;
  for (int c = 0; c < nc; c++)
    if (csize[(size_t)c] > 1) { cvert[(size_t)c] = V + nontrivial++; size_nontrivial += csize[(size_t)c]; }
  const int VF = V + nontrivial;  // final vertex ids: trivial real vertices keep theirs, members map to cvert
  auto fin = [&](int x) { return csize[(size_t)comp[(size_t)x]] > 1 ? cvert[(size_t)comp[(size_t)x]] : x; };
  std::vector<int> din((size_t)VF, 0), dout((size_t)VF, 0);
  std::vector<std::pair<int, int>> fe;
  fe.reserve(E);
  size_t loops_trivial = 0;
  for (auto& ed : edges.e) {
    if (comp[(size_t)ed.first] == comp[(size_t)ed.second]) {
      if (csize[(size_t)comp[(size_t)ed.first]] == 1) loops_trivial++;  // self loop on a trivial vertex (:1385-1402)
      continue;  // intra-component edges disappear with clear_vertex (:1374-1378)
    }
    const int a = fin(ed.first), b = fin(ed.second);  // one contracted edge per real edge, parallel edges kept
    fe.emplace_back(a, b);
    dout[(size_t)a]++;
    din[(size_t)b]++;
  }
  out->sub[0] = (uint64_t)V;
  out->sub[1] = (uint64_t)(E - loops_trivial);
  out->sub[2] = (uint64_t)nontrivial;
  out->sub[3] = (uint64_t)size_nontrivial;
  out->sub[4] = (uint64_t)(VF - size_nontrivial);
  out->sub[5] = (uint64_t)fe.size();

  // topological order of the condensed multigraph (Kahn), then :1420-1434
  std::vector<int> foff((size_t)VF + 1, 0), fadj(fe.size());
  for (auto& ed : fe) foff[(size_t)ed.first + 1]++;
  for (int i = 0; i < VF; i++) foff[(size_t)i + 1] += foff[(size_t)i];
  {
    std::vector<int> pos(foff.begin(), foff.end() - 1);
    for (auto& ed : fe) fadj[(size_t)pos[(size_t)ed.first]++] = ed.second;
  }
  std::vector<int> indeg(din), order;
  order.reserve((size_t)VF);
  for (int i = 0; i < VF; i++) if (indeg[(size_t)i] == 0) order.push_back(i);
  for (size_t qi = 0; qi < order.size(); qi++) {
    const int u = order[qi];
    for (int x = foff[(size_t)u]; x < foff[(size_t)u + 1]; x++)
      if (--indeg[(size_t)fadj[(size_t)x]] == 0) order.push_back(fadj[(size_t)x]);
  }
  std::vector<int> fbranch((size_t)VF, 0);
  int bc = 1;
  for (int u : order) {
    if (din[(size_t)u] >= 1 || dout[(size_t)u] >= 1) {
      if (din[(size_t)u] > 1) bc -= din[(size_t)u] - 1;
      fbranch[(size_t)u] = bc;
      if (dout[(size_t)u] > 1) bc += dout[(size_t)u] - 1;
    }
  }
  out->branch.assign((size_t)V, 0);
  for (int i = 0; i < V; i++)
    if (csize[(size_t)comp[(size_t)i]] == 1) out->branch[(size_t)i] = fbranch[(size_t)i];  // SCC members stay 0
}

void post_traceback(const Graph& g, const FillParams& p, const GapJob& job, const DpView& v, const PostPrep& prep,
                    GlibcRand& rng, char* buf, g2s_result* res) {
  const GapOut& go = *v.out;
  const int lmf = job.lmf, k = p.k;
  const uint32_t* lseeds = job.lseeds();
  res->right_fuz = go.reached_j;  // :1171
  res->flags |= G2S_GAP_PHASE_D;
  int d2 = go.len[(size_t)(rng.next() % go.n_len)];  // :1440
  res->draws++;
  int last_solid = d2;
  uint32_t cur = job.targets()[go.reached_j];
  buf[d2] = '\0';
  uint32_t backv[4];
  while (d2 >= 0) {
    if (d2 <= lmf && (lseeds[d2] >> 1) == (cur >> 1)) {  // :1455-1462, k-mer equality
      res->left_fuz = lmf - d2;
      break;
    }
    if (d2 > 0) {
      bool solid = p.skip_confident;
      if (!solid) {
        auto it = prep.vertex_of.find(cur >> 1);
        const int bv = it == prep.vertex_of.end() ? 0 : it->second;  // Q5: unknown k-mers read branch[sink]
        solid = prep.branch[(size_t)bv] == 1;
      }
      if (solid) last_solid = d2;
      const char c = g.last_char(cur);
      buf[d2 - 1] = (d2 > last_solid - k) ? (char)toupper((unsigned char)c) : (char)tolower((unsigned char)c);
      int nb = 0;
      for (int nt = 0; nt < 4; nt++) {  // GATB predecessor order
        const uint32_t pr = g.pred_of(cur, nt);
        if (pr != kInvalidNode && dp_find(v, d2 - 1, pr) > 0) backv[nb++] = pr;
      }
      if (nb == 0) {  // :1493-1510
        snprintf(res->backtrace_msg, sizeof res->backtrace_msg, "Unable to backtrace! %d %d %s", d2,
                 go.final_d, g.node_string(job.targets()[go.reached_j]).c_str());
        res->flags |= G2S_GAP_BACKTRACE_FAIL;
        res->count = 0;
        return;
      }
      cur = backv[rng.next() % nb];  // :1513
      res->draws++;
    }
    d2--;
  }
  res->count = prep.count;
}

}  // namespace g2s
