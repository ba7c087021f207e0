// gap2seq_amd/csrc/post.cpp — see post.hpp.
#include "post.hpp"
#include <mutex>

#include <algorithm>
#include <chrono>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace g2s {

namespace {

inline int sat_add(int a, int b) {
  long long s = (long long)a + (long long)b;
  return s > (long long)G2S_MAX_PATHS ? G2S_MAX_PATHS : (int)s;
}

// canonical k-mer index -> dense vertex id, open addressing (no allocation per key)
struct VertexMap {
  std::vector<uint32_t> key, val;
  uint32_t mask = 0;
  void init(size_t n) {
    size_t cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    key.assign(cap, 0xFFFFFFFFu);
    val.assign(cap, 0);
    mask = (uint32_t)cap - 1;
  }
  static uint32_t h(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
  int find(uint32_t k) const {
    uint32_t s = h(k) & mask;
    while (key[s] != 0xFFFFFFFFu) { if (key[s] == k) return (int)val[s]; s = (s + 1) & mask; }
    return -1;
  }
  int get_or_add(uint32_t k, int* next) {
    uint32_t s = h(k) & mask;
    while (key[s] != 0xFFFFFFFFu) { if (key[s] == k) return (int)val[s]; s = (s + 1) & mask; }
    key[s] = k;
    val[s] = (uint32_t)(*next);
    return (*next)++;
  }
};

}  // namespace
thread_local double g2s_post_laps[12];
namespace {
// Tarjan over a CSR adjacency; component ids are arbitrary.
int strong_components(int nv, const std::vector<int>& off, const std::vector<int>& adj, std::vector<int>* comp) {
  static thread_local std::vector<int> index, low, it, stack, call;
  static thread_local std::vector<char> on;
  index.assign((size_t)nv, -1);
  low.assign((size_t)nv, 0);
  it.assign((size_t)nv, 0);
  on.assign((size_t)nv, 0);
  stack.clear();
  call.clear();
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

void sub_convert(const SubState* in, uint32_t n, std::vector<SubRec>* recs, std::vector<uint64_t>* xp) {
  for (uint32_t i = 0; i < n; i++) {
    const SubState& s = in[i];
    SubRec r;
    r.node = s.node;
    r.cnt = s.cnt;
    r.meta = (s.depth & G2S_SUB_META_DEPTH_MASK) | (s.flags << G2S_SUB_META_FLAG_SHIFT);
    r.pred = -1;
    for (int nt = 0; nt < 4; nt++) {
      if (s.pred[nt] < 0) continue;
      if (r.pred < 0) r.pred = s.pred[nt];
      else { r.pred |= G2S_SUB_MORE; xp->push_back(((uint64_t)i << 32) | (uint32_t)s.pred[nt]); }
    }
    recs->push_back(r);
  }
}

void seg_expand(const FillParams& p, const GapJob& job, const GapOut& go, const SegRec* segs, uint32_t n_segs,
                SubRec* out, uint64_t* xp) {
  const int lmf = job.lmf, rmf = job.rmf;
  const uint32_t* targets = job.targets();
  const bool want_s = !p.skip_confident;
  const uint32_t sinknode = (want_s && p.all_paths && rmf >= 1) ? targets[rmf - 1] : kInvalidNode;  // Q3/Q4 (:1195-1244)
  const int lo_sink = std::max(0, lmf + job.g - p.d_err);                                          // :1196
  const uint32_t reached = targets[go.reached_j];
  const uint32_t t_flags = G2S_SUB_IN_T | G2S_SUB_START_T | ((want_s && !p.all_paths) ? (G2S_SUB_IN_S | G2S_SUB_SINK) : 0u);
  static thread_local std::vector<uint32_t> base;
  base.resize(n_segs);
  uint32_t total = 0;
  for (uint32_t i = 0; i < n_segs; i++) { base[i] = total; total += segs[i].depth_len >> 16; }
  uint32_t nx = 0;
  for (uint32_t i = 0; i < n_segs; i++) {
    const SegRec& s = segs[i];
    const int d0 = (int)(s.depth_len & 0xFFFFu), len = (int)(s.depth_len >> 16);
    const int ts = seg_ts(s), tt = seg_tt(s);
    const bool up = (s.node & 1u) == 0;
    SubRec* o = out + base[i];
    for (int t = len - 1; t >= 0; t--, o++) {
      const uint32_t node = up ? s.node + 2u * (uint32_t)t : s.node - 2u * (uint32_t)t;
      const int depth = d0 + t;
      uint32_t f = (t <= ts ? G2S_SUB_IN_S : 0u) | (t <= tt ? G2S_SUB_IN_T : 0u);
      if (node == sinknode && depth >= lo_sink) f |= G2S_SUB_IN_S | G2S_SUB_SINK;
      if (node == reached && (depth == go.len[0] || (go.n_len > 1 && depth == go.len[1]))) f |= t_flags;  // :1245-1259
      int32_t pred = t > 0 ? (int32_t)(base[i] + (uint32_t)(len - t)) : -1;
      if (t == 0) {
        if (s.flags & G2S_SUB_SOURCE) f |= G2S_SUB_SOURCE;  // :1270
        else {
          const uint32_t ps[4] = {s.par01 & 0xFFFFu, s.par01 >> 16, s.par23 & 0xFFFFu, s.par23 >> 16};
          for (int q = 0; q < 4; q++) {
            if (ps[q] == 0xFFFFu) continue;
            const uint32_t pb = base[ps[q]];  // the parent's last state is its first record
            if (pred < 0) pred = (int32_t)pb;
            else { pred |= G2S_SUB_MORE; xp[nx++] = ((uint64_t)(base[i] + (uint32_t)(len - 1)) << 32) | pb; }
          }
        }
      }
      o->node = node;
      o->cnt = s.cnt;
      o->meta = (uint32_t)depth | (f << G2S_SUB_META_FLAG_SHIFT);
      o->pred = pred;
    }
  }
}

void sub_analyze(const FillParams& p, const GapJob& job, const SubView& v, SubPrep* out) {
  const GapOut& go = *v.out;
  out->count = go.c_count;
  out->phase_d = go.c_count > 0 && go.n_len > 0;  // :1169 (fill is never NULL here)
  if (!out->phase_d) return;
  const uint32_t n = v.n;
  const SubRec* st = v.st;
  (void)job;
  auto FL = [&](uint32_t i) -> uint32_t { return sub_flags(st[i]); };

  // traceback starts and, per start, the depth at which every traceback stops
  // (states are stored depth-descending, so a reverse sweep sees predecessors first)
  // per-thread scratch: these vectors are reused across gaps (no allocation in steady state)
  static thread_local std::vector<int> lo, hi, vid, off, adj, comp, csize, cvert, din, dout, foff, fadj, indeg, order, fbranch;
  static thread_local std::vector<uint64_t> el;
  static thread_local std::vector<std::pair<int, int>> fe;
  static thread_local VertexMap vm;
  // One reverse sweep (states are stored depth-descending, so it sees predecessors first)
  // collects everything that is per state: stop depths of the traceback closure, vertex ids
  // of the S closure, out-degrees, edge and sink/source counts.
  static thread_local std::vector<int> outdeg;
  const bool want_s = !p.skip_confident;  // no D1/D2 with -all-upper (:1181)
  lo.assign((size_t)n, -2);  // -2: not on a traceback closure
  hi.assign((size_t)n, -2);
  int nverts = 2;
  uint32_t n_s = 0, n_t_only = 0;
  int count_s = 0, src_out = 0, sink_in = 0;
  uint64_t edges = 0;
  if (want_s) {
    vm.init(n);
    vid.assign((size_t)n, -1);
    outdeg.assign((size_t)n, 0);
  }
  for (int64_t i = (int64_t)n - 1; i >= 0; i--) {
    const SubRec& s = st[i];
    const uint32_t sf = sub_flags(s);
    int32_t pr[4];
    const bool src = (sf & G2S_SUB_SOURCE) != 0;
    const int np = src ? 0 : sub_preds(v, (uint32_t)i, pr);
    if (sf & G2S_SUB_IN_T) {
      if (src) { lo[(size_t)i] = hi[(size_t)i] = (int)sub_depth(s); }  // :1455-1462
      else if (np == 0) { lo[(size_t)i] = -1; hi[(size_t)i] = 1 << 30; }  // walk ends here without a stop: not fixed
      else {
        int l = 1 << 30, h = -1;
        for (int x = 0; x < np; x++) {
          const int32_t q = pr[x];
          if (lo[(size_t)q] < 0) { l = -1; h = 1 << 30; } else { l = std::min(l, lo[(size_t)q]); h = std::max(h, hi[(size_t)q]); }
        }
        lo[(size_t)i] = l; hi[(size_t)i] = h;
      }
      if (sf & G2S_SUB_START_T) {
        for (int j = 0; j < go.n_len && j < 2; j++)
          if ((int)sub_depth(s) == go.len[j]) {  // the lowest index wins, as in a forward scan
            out->start_idx[j] = (int)i;
            out->stop_depth[j] = (lo[(size_t)i] >= 0 && lo[(size_t)i] == hi[(size_t)i]) ? lo[(size_t)i] : -1;
          }
      }
      if (!(sf & G2S_SUB_IN_S)) n_t_only++;
    }
    if (want_s && (sf & G2S_SUB_IN_S)) {
      vid[(size_t)i] = vm.get_or_add(s.node >> 1, &nverts);
      n_s++;
      if (sf & G2S_SUB_SINK) { sink_in++; edges++; count_s = sat_add(count_s, (int)s.cnt); }
      if (src) { src_out++; edges++; }
      else for (int x = 0; x < np; x++) { outdeg[(size_t)pr[x]]++; edges++; }
    }
  }
  if (!want_s) return;

  if ((uint32_t)nverts == n_s + 2) {
    // ---- fast path: every k-mer occurs at exactly one depth of the S closure.  An edge goes
    // from depth d-1 to depth d, so the subgraph is a DAG (a cycle would need a k-mer at two
    // depths), vertices are the states themselves, nothing is contracted, no self loops, and
    // ascending depth (reverse emission order) is a topological order.
    if (p.all_paths) out->count = count_s;
    out->sub[0] = (uint64_t)nverts;
    out->sub[1] = edges;
    out->sub[2] = 0;
    out->sub[3] = 0;
    out->sub[4] = (uint64_t)nverts;
    out->sub[5] = edges;
    out->safe.assign((size_t)n, 0);
    int bc = 1;
    // source (vertex 1) comes first in topological order: in-degree 0
    if (src_out >= 1) { if (src_out > 1) bc += src_out - 1; }
    for (int64_t i = (int64_t)n - 1; i >= 0; i--) {
      const uint32_t sf = FL((uint32_t)i);
      if (!(sf & G2S_SUB_IN_S)) continue;
      int din = 1;
      if (!(sf & G2S_SUB_SOURCE)) {
        const int32_t pw = st[i].pred;
        din = pw < 0 ? 0 : 1;
        if (pw >= 0 && (pw & G2S_SUB_MORE)) { int32_t pr[4]; din = sub_preds(v, (uint32_t)i, pr); }
      }
      const int dout = outdeg[(size_t)i] + ((sf & G2S_SUB_SINK) ? 1 : 0);
      if (din >= 1 || dout >= 1) {
        if (din > 1) bc -= din - 1;
        out->safe[(size_t)i] = bc == 1;
        if (dout > 1) bc += dout - 1;
      }
    }
    // sink (vertex 0) is last
    bool sink_safe = false;
    if (sink_in >= 1) { if (sink_in > 1) bc -= sink_in - 1; sink_safe = bc == 1; }
    if (n_t_only) {
      for (uint32_t i = 0; i < n; i++) {  // traceback states outside the subgraph (Q5)
        if (!(FL(i) & G2S_SUB_IN_T) || (FL(i) & G2S_SUB_IN_S)) continue;
        const int vtx = vm.find(st[i].node >> 1);
        if (vtx < 0) { out->safe[i] = sink_safe; continue; }
        // its k-mer is in the subgraph at another depth: that vertex's value
        for (uint32_t q = 0; q < n; q++)
          if ((FL(q) & G2S_SUB_IN_S) && vid[q] == vtx) { out->safe[i] = out->safe[q]; break; }
      }
    }
    return;
  }

  el.clear();
  el.reserve((size_t)n + 8);
  int count = 0;
  auto edge = [&](int a, int b) { el.push_back(((uint64_t)(uint32_t)a << 32) | (uint32_t)b); };
  for (uint32_t i = 0; i < n; i++) {
    const SubRec& s = st[i];
    const uint32_t sf = FL(i);
    if (!(sf & G2S_SUB_IN_S)) continue;
    if (sf & G2S_SUB_SINK) { edge(vid[i], 0); count = sat_add(count, (int)s.cnt); }  // :1216-1226 / :1248-1256
    if (sf & G2S_SUB_SOURCE) { edge(1, vid[i]); continue; }                          // :1303-1305
    int32_t pr[4];
    const int np = sub_preds(v, i, pr);
    for (int x = 0; x < np; x++) edge(vid[(size_t)pr[x]], vid[i]);                     // :1283-1297
  }
  if (p.all_paths) out->count = count;  // recount (:1189-1191); -best-only keeps the phase C count
  std::sort(el.begin(), el.end());
  el.erase(std::unique(el.begin(), el.end()), el.end());  // boost::edge(u,v).second de-duplication

  const int V = nverts;
  const size_t E = el.size();
  off.assign((size_t)V + 1, 0);
  adj.resize(E);
  for (uint64_t x : el) off[(size_t)(x >> 32) + 1]++;
  for (int i = 0; i < V; i++) off[(size_t)i + 1] += off[(size_t)i];
  for (size_t x = 0; x < E; x++) adj[x] = (int)(uint32_t)el[x];  // el is sorted by source: already CSR order
  const int nc = strong_components(V, off, adj, &comp);
  csize.assign((size_t)nc, 0);
  cvert.assign((size_t)nc, -1);
  for (int i = 0; i < V; i++) csize[(size_t)comp[(size_t)i]]++;
  int nontrivial = 0, size_nontrivial = 0;
  for (int c = 0; c < nc; c++)
    if (csize[(size_t)c] > 1) { cvert[(size_t)c] = V + nontrivial++; size_nontrivial += csize[(size_t)c]; }
  const int VF = V + nontrivial;  // trivial vertices keep their id, SCC members map to the contracted vertex
  auto fin = [&](int x) { return csize[(size_t)comp[(size_t)x]] > 1 ? cvert[(size_t)comp[(size_t)x]] : x; };
  din.assign((size_t)VF, 0);
  dout.assign((size_t)VF, 0);
  fe.clear();
  fe.reserve(E);
  size_t loops_trivial = 0;
  for (uint64_t x : el) {
    const int a0 = (int)(x >> 32), b0 = (int)(uint32_t)x;
    if (comp[(size_t)a0] == comp[(size_t)b0]) {
      if (csize[(size_t)comp[(size_t)a0]] == 1) loops_trivial++;  // self loop on a trivial vertex (:1385-1402)
      continue;  // intra-component edges vanish with clear_vertex (:1374-1378)
    }
    const int a = fin(a0), b = fin(b0);  // one contracted edge per real edge, parallel edges kept (:1342-1372)
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

  // topological order of the condensed multigraph (Kahn), then the branch rule (:1420-1434)
  foff.assign((size_t)VF + 1, 0);
  fadj.resize(fe.size());
  for (auto& ed : fe) foff[(size_t)ed.first + 1]++;
  for (int i = 0; i < VF; i++) foff[(size_t)i + 1] += foff[(size_t)i];
  {
    std::vector<int> pos(foff.begin(), foff.end() - 1);
    for (auto& ed : fe) fadj[(size_t)pos[(size_t)ed.first]++] = ed.second;
  }
  indeg = din;
  order.clear();
  order.reserve((size_t)VF);
  for (int i = 0; i < VF; i++) if (indeg[(size_t)i] == 0) order.push_back(i);
  for (size_t qi = 0; qi < order.size(); qi++) {
    const int u = order[qi];
    for (int x = foff[(size_t)u]; x < foff[(size_t)u + 1]; x++)
      if (--indeg[(size_t)fadj[(size_t)x]] == 0) order.push_back(fadj[(size_t)x]);
  }
  fbranch.assign((size_t)VF, 0);
  int bc = 1;
  for (int u : order) {
    if (din[(size_t)u] >= 1 || dout[(size_t)u] >= 1) {
      if (din[(size_t)u] > 1) bc -= din[(size_t)u] - 1;
      fbranch[(size_t)u] = bc;
      if (dout[(size_t)u] > 1) bc += dout[(size_t)u] - 1;
    }
  }
  // per-state safe bit for the traceback (:1466): k-mers outside the subgraph read
  // branch[sink] (node2boost default-inserts vertex 0, Q5); SCC members stay 0
  out->safe.assign((size_t)n, 0);
  const bool sink_safe = (csize[(size_t)comp[0]] == 1) && fbranch[0] == 1;
  for (uint32_t i = 0; i < n; i++) {
    if (!(FL(i) & G2S_SUB_IN_T)) continue;
    int vtx = vid[i];
    if (vtx < 0) vtx = vm.find(st[i].node >> 1);
    if (vtx < 0) { out->safe[i] = sink_safe; continue; }
    out->safe[i] = (csize[(size_t)comp[(size_t)vtx]] == 1) && fbranch[(size_t)vtx] == 1;
  }
}

void sub_traceback(const Graph& g, const FillParams& p, const GapJob& job, const SubView& v, const SubPrep& prep,
                   const uint32_t* rands, char* buf, g2s_result* res) {
  const GapOut& go = *v.out;
  const int lmf = job.lmf, k = p.k;
  res->right_fuz = go.reached_j;  // :1171
  res->flags |= G2S_GAP_PHASE_D;
  int draws = 0;
  // rands == nullptr: the caller knows that every draw is taken modulo 1
  auto draw = [&]() -> uint32_t { const uint32_t r = rands ? rands[draws] : 0u; draws++; return r; };
  const int pick = (int)((draw() >> 1) % (uint32_t)go.n_len);  // :1440
  int d2 = go.len[pick];
  int last_solid = d2;
  int i = prep.start_idx[pick];
  buf[d2] = '\0';
  res->count = prep.count;
  while (d2 >= 0 && i >= 0) {
    const SubRec& s = v.st[i];
    if (sub_flags(s) & G2S_SUB_SOURCE) {  // :1455-1462 (depth <= lmf and k-mer equal to the flank k-mer)
      res->left_fuz = lmf - d2;
      break;
    }
    if (d2 > 0) {
      if (p.skip_confident || prep.safe[(size_t)i]) last_solid = d2;  // :1466-1468
      const char c = g.last_char(s.node);
      buf[d2 - 1] = (d2 > last_solid - k) ? (char)toupper((unsigned char)c) : (char)tolower((unsigned char)c);
      int32_t back[4];
      const int nb = sub_preds(v, (uint32_t)i, back);
      if (nb > 1) {  // GATB predecessor order: predecessors(v)[slot] is the parent p whose p^1 ends with base `slot`
        int32_t by_slot[4] = {-1, -1, -1, -1};
        for (int x = 0; x < nb; x++) by_slot[g.lastnt[v.st[(size_t)back[x]].node ^ 1u]] = back[x];
        int w = 0;
        for (int nt = 0; nt < 4; nt++) if (by_slot[nt] >= 0) back[w++] = by_slot[nt];
      }
      if (nb == 0) {  // :1493-1510
        res->backtrace_depth = d2;  // :1494 (the line itself: g2s_backtrace_text)
        res->backtrace_final_d = go.final_d;
        res->flags |= G2S_GAP_BACKTRACE_FAIL;
        res->count = 0;
        buf[lmf] = '\0';  // no fill: the caller's view of the buffer starts here (left_fuz stays 0)
        break;
      }
      const uint32_t rv = draw() >> 1;  // :1513 (drawn even when there is one choice)
      i = nb == 1 ? back[0] : back[rv % (uint32_t)nb];
    }
    d2--;
  }
  res->draws = draws;
}

// The number of rand() values sub_traceback will consume from `rands` on, without writing
// the fill: the in-order offset pass needs only this of a gap whose draw count depends on
// the draws themselves; the traceback proper then runs on the pool like every other.
int sub_count_draws(const Graph& g, const SubView& v, const SubPrep& prep, const uint32_t* rands) {
  const GapOut& go = *v.out;
  int draws = 0;
  const int pick = (int)((rands[draws++] >> 1) % (uint32_t)go.n_len);
  int d2 = go.len[pick];
  int i = prep.start_idx[pick];
  while (d2 >= 0 && i >= 0) {
    if (sub_flags(v.st[i]) & G2S_SUB_SOURCE) break;
    if (d2 > 0) {
      int32_t back[4];
      const int nb = sub_preds(v, (uint32_t)i, back);
      if (nb == 0) break;
      if (nb > 1) {
        int32_t by_slot[4] = {-1, -1, -1, -1};
        for (int x = 0; x < nb; x++) by_slot[g.lastnt[v.st[(size_t)back[x]].node ^ 1u]] = back[x];
        int w = 0;
        for (int nt = 0; nt < 4; nt++) if (by_slot[nt] >= 0) back[w++] = by_slot[nt];
      }
      const uint32_t rv = rands[draws++] >> 1;
      i = nb == 1 ? back[0] : back[rv % (uint32_t)nb];  // (a division per base is most of this walk's time)
    }
    d2--;
  }
  return draws;
}

// ---------------------------------------------------------------------------
// TEST HOOK support (host restatement of what g2s_extract computes)
// ---------------------------------------------------------------------------
uint32_t HostTable::find(int depth, uint32_t node) const {
  if (depth < 0 || depth > D) return 0;
  uint32_t lo = lvl[(size_t)depth], hi = lvl[(size_t)depth + 1];
  const uint32_t end = hi;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if ((uint32_t)(states[mid] >> 32) < node) lo = mid + 1; else hi = mid;
  }
  return (lo < end && (uint32_t)(states[lo] >> 32) == node) ? (uint32_t)states[lo] : 0;
}

void host_closure(const Graph& g, const FillParams& p, const GapJob& job, const HostTable& t, const GapOut& go,
                  std::vector<SubState>* out, uint32_t* q7) {
  out->clear();
  if (!(go.c_count > 0 && go.n_len > 0)) return;
  const int lmf = job.lmf, rmf = job.rmf;
  const uint32_t* targets = job.targets();
  const uint32_t* lseeds = job.lseeds();
  const bool want_s = !p.skip_confident;
  const uint32_t sinknode = (want_s && p.all_paths && rmf >= 1) ? targets[rmf - 1] : kInvalidNode;
  const int lo_sink = std::max(0, lmf + job.g - p.d_err);
  const uint32_t reached = targets[go.reached_j];
  const uint32_t t_flags = G2S_SUB_IN_T | G2S_SUB_START_T | ((want_s && !p.all_paths) ? (G2S_SUB_IN_S | G2S_SUB_SINK) : 0u);
  std::vector<std::pair<uint64_t, int>> index;  // (node<<32|depth) -> state index, kept sorted per level lazily
  auto find_idx = [&](uint32_t node, int depth, size_t from) -> int {
    for (size_t i = from; i < out->size(); i++)
      if ((*out)[i].node == node && (int)(*out)[i].depth == depth) return (int)i;
    return -1;
  };
  auto discover = [&](uint32_t node, int depth, size_t level_from) -> int {
    int idx = find_idx(node, depth, level_from);
    if (idx >= 0) return idx;
    SubState s;
    s.node = node; s.depth = (uint32_t)depth; s.cnt = t.find(depth, node); s.flags = 0;
    s.pred[0] = s.pred[1] = s.pred[2] = s.pred[3] = -1;
    out->push_back(s);
    return (int)out->size() - 1;
  };
  size_t bstart = 0;
  for (int d2 = t.D; d2 >= 0; d2--) {
    if (sinknode != kInvalidNode && d2 >= lo_sink && t.find(d2, sinknode) > 0)
      (*out)[(size_t)discover(sinknode, d2, bstart)].flags |= G2S_SUB_IN_S | G2S_SUB_SINK;
    for (int j = 0; j < go.n_len; j++)
      if (go.len[j] == d2 && t.find(d2, reached) > 0) (*out)[(size_t)discover(reached, d2, bstart)].flags |= t_flags;
    const size_t bend = out->size();
    const uint32_t lidx = d2 <= lmf ? (lseeds[d2] >> 1) : 0xFFFFFFFFu;
    for (size_t i = bstart; i < bend; i++) {
      const uint32_t cur = (*out)[i].node;
      const uint32_t f = (*out)[i].flags & (G2S_SUB_IN_S | G2S_SUB_IN_T);
      if (d2 <= lmf && (cur >> 1) == lidx) { (*out)[i].flags |= G2S_SUB_SOURCE; continue; }
      if (d2 == 0) continue;
      for (int nt = 0; nt < 4; nt++) {
        const uint32_t pr = g.pred_of(cur, nt);
        if (pr == kInvalidNode || t.find(d2 - 1, pr) == 0) continue;
        const int idx = discover(pr, d2 - 1, bend);
        (*out)[i].pred[nt] = idx;
        (*out)[(size_t)idx].flags |= f;
        if (find_idx(pr ^ 1u, d2 - 1, bend) >= 0) *q7 |= G2S_GAP_Q7;
      }
    }
    // a later sibling may have discovered the other strand after the check above
    for (size_t a = bend; a < out->size(); a++)
      for (size_t b2 = a + 1; b2 < out->size(); b2++)
        if (((*out)[a].node ^ (*out)[b2].node) == 1u) *q7 |= G2S_GAP_Q7;
    bstart = bend;
  }
}

}  // namespace g2s

// ---------------------------------------------------------------------------
// Phase D2/D3 on closure segments (segment tier).
// ---------------------------------------------------------------------------
namespace g2s {

namespace {

inline int seg_find(uint32_t v0, int len, uint32_t node) {  // t with state t == node, else -1
  if (node == kInvalidNode || ((node ^ v0) & 1u)) return -1;
  const int dt = (int)(node - v0) >> 1;
  const int t = (v0 & 1u) ? -dt : dt;
  return (t >= 0 && t < len) ? t : -1;
}
inline uint32_t seg_state(const SegRec& s, int t) { return (s.node & 1u) ? s.node - 2u * (uint32_t)t : s.node + 2u * (uint32_t)t; }
inline int seg_parents(const SegRec& s, uint32_t out[4]) {
  if (s.flags & G2S_SUB_SOURCE) return 0;
  int n = 0;
  const uint32_t ps[4] = {s.par01 & 0xFFFFu, s.par01 >> 16, s.par23 & 0xFFFFu, s.par23 >> 16};
  for (int q = 0; q < 4; q++) if (ps[q] != 0xFFFFu) out[n++] = ps[q];
  return n;
}

// the run (SubPrep::runs, run_mode) that holds k-mer index x, or nullptr
inline const SegRun* run_of(const SubPrep& prep, uint32_t x) {
  size_t lo = 0, hi = prep.runs.size();
  while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (prep.runs[mid].lo <= x) lo = mid + 1; else hi = mid; }
  return (lo > 0 && x <= prep.runs[lo - 1].hi) ? &prep.runs[lo - 1] : nullptr;
}

// safe bit of state t of segment i (:1466; k-mers outside the subgraph read branch[sink], Q5)
inline bool seg_safe(const SubView& v, const SubPrep& prep, uint32_t i, int t) {
  const SegRec& s = v.segs[i];
  if (prep.run_mode) {  // the verdict belongs to the k-mer, whatever the depth (k-mers outside the subgraph: Q5)
    const SegRun* r = run_of(prep, seg_state(s, t) >> 1);
    return r ? r->safe != 0 : prep.sink_safe;
  }
  const int ts = seg_ts(s);
  // the branch rule's verdict for state tq of segment q: from the host analysis (prep.seg) or, when
  // phase D2 ran on the device, from the bits it left in the record
  auto verdict = [&](uint32_t q, int tq) -> bool {
    if (prep.seg) return tq <= prep.seg[q].split ? prep.seg[q].safe_a : prep.seg[q].safe_b;
    const SegRec& o = v.segs[q];
    return tq <= (int)o.pad ? (o.ts_tt & 0x8000u) != 0 : (o.ts_tt & 0x80000000u) != 0;
  };
  if (t <= ts) return verdict(i, t);
  // a traceback state outside the subgraph: its k-mer may be in the subgraph at another depth
  const uint32_t x = seg_state(s, t) >> 1;
  if (prep.s_iv) {
    const std::pair<uint32_t, uint32_t>* iv = prep.s_iv;
    size_t lo = 0, hi = prep.n_iv;
    while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (iv[mid].first <= x) lo = mid + 1; else hi = mid; }
    if (lo > 0) {
      const uint32_t q = iv[lo - 1].second;
      const SegRec& o = v.segs[q];
      const uint32_t oidx = o.node >> 1;
      const int tq = (o.node & 1u) ? (int)oidx - (int)x : (int)x - (int)oidx;
      if (tq >= 0 && tq <= seg_ts(o)) return verdict(q, tq);
    }
  } else {  // (device analysis keeps no index of the intervals: such states are rare, scan)
    for (uint32_t q = 0; q < v.n_segs; q++) {
      const SegRec& o = v.segs[q];
      const uint32_t oidx = o.node >> 1;
      const int tq = (o.node & 1u) ? (int)oidx - (int)x : (int)x - (int)oidx;
      if (tq >= 0 && tq <= seg_ts(o)) return verdict(q, tq);
    }
  }
  return prep.sink_safe;
}

}  // namespace

namespace {

// Stable LSD radix sort of unsigned keys on the bits of `bits` (one contiguous field); only the digits in which the
// keys differ take a pass.  (The analysis below sorts some twenty thousand k-mer indices and edges for a closure of four thousand
// segments: with std::sort that was 60 % of its time.)
template <class T>
void radix_sort(T* a, size_t n, T bits) {
  static thread_local std::vector<T> tmp;
  if (n < 2) return;
  if (n <= 24) {
    for (size_t i = 1; i < n; i++) {
      const T x = a[i];
      size_t j = i;
      while (j > 0 && (a[j - 1] & bits) > (x & bits)) { a[j] = a[j - 1]; j--; }
      a[j] = x;
    }
    return;
  }
  T diff = 0;
  for (size_t i = 1; i < n; i++) diff |= a[i] ^ a[0];
  diff &= bits;
  if (tmp.size() < n) tmp.resize(n);
  T* src = a;
  T* dst = tmp.data();
  constexpr unsigned kDigit = 11u;  // (20-odd bits of k-mer index: two passes)
  unsigned b = 0;
  while (b < sizeof(T) * 8u && !((bits >> b) & (T)1)) b++;
  for (; b < sizeof(T) * 8u; b += kDigit) {
    const T dm = (bits >> b) & (T)((1u << kDigit) - 1u);
    if (!((diff >> b) & dm)) continue;
    uint32_t cnt[1u << kDigit];
    memset(cnt, 0, sizeof cnt);
    for (size_t i = 0; i < n; i++) cnt[(size_t)((src[i] >> b) & dm)]++;
    uint32_t sum = 0;
    for (unsigned q = 0; q < (1u << kDigit); q++) { const uint32_t c = cnt[q]; cnt[q] = sum; sum += c; }
    for (size_t i = 0; i < n; i++) dst[cnt[(size_t)((src[i] >> b) & dm)]++] = src[i];
    std::swap(src, dst);
  }
  if (src != a) memcpy(a, src, n * sizeof(T));
}
constexpr uint64_t kHi32 = 0xFFFFFFFF00000000ull, kLo32 = 0x00000000FFFFFFFFull;

// intervals as first << 32 | last: sorted and disjoint afterwards; overlapping ones merge, adjacent ones when asked
void merge_intervals(std::vector<uint64_t>* a, bool adjacent) {
  radix_sort(a->data(), a->size(), kHi32);
  size_t w = 0;
  for (size_t i = 0; i < a->size(); i++) {
    const uint64_t x = (*a)[i];
    const uint32_t lo = (uint32_t)(x >> 32), hi = (uint32_t)x;
    if (w > 0 && (uint64_t)lo <= (uint64_t)(uint32_t)(*a)[w - 1] + (adjacent ? 1u : 0u)) {
      if (hi > (uint32_t)(*a)[w - 1]) (*a)[w - 1] = ((*a)[w - 1] & kHi32) | hi;
    } else (*a)[w++] = x;
  }
  a->resize(w);
}
// membership in such a list for queries that never decrease
struct IntervalSweep {
  const std::vector<uint64_t>& a;
  size_t p = 0;
  explicit IntervalSweep(const std::vector<uint64_t>& v) : a(v) {}
  bool has(uint32_t x) {
    while (p < a.size() && (uint32_t)a[p] < x) p++;
    return p < a.size() && (uint32_t)(a[p] >> 32) <= x;
  }
};

// :1314-1435 on a closure in which k-mers occur at several depths.  The reference's vertices are k-mers
// (node2boost), its edges the deduplicated (boost::edge(u, v).second) state transitions.  Inside a unitig the
// transitions are index +-1 steps, so the k-mer graph is made of CHAINS: cut the closure's index intervals at
// every segment end and at every k-mer that carries another edge (a parent's last k-mer, an entry, a sink
// position); what lies between two cuts is a run of k-mers whose only edges are the chain's own (one
// direction, or both when an upward and a downward segment cover it).  Strong components (Tarjan), the
// contraction and the branch rule run on runs; a run that is not in a component of several vertices stands
// for a row of trivial vertices with one edge in and one out each, along which the running count of the
// branch rule cannot change.  sinkpos: position of the sink state inside each segment or -1.
// (Everything in front of the strong components is radix sorts and sweeps over sorted lists: the closures of
// a -dist-error 2000 list have thousands of segments each, and a list waits for the slowest of them.)
void seg_analyze_runs(const FillParams& p, const SubView& v, SubPrep* out, const std::vector<int>& sinkpos) {
  const uint32_t n = v.n_segs;
  const SegRec* sg = v.segs;
  static thread_local std::vector<uint64_t> viv, uiv, div, sp;  // intervals first << 32 | last; edges from << 32 | to
  static thread_local std::vector<uint32_t> bp, srcs, sinks, loopk;
  static thread_local std::vector<int> bp_run;
  static thread_local std::vector<uint8_t> is_par;
  static thread_local std::vector<std::pair<int, int>> ce;
  static thread_local std::vector<int> off, adj, comp, cnodes, cweight, din, dout, foff, fadj, indeg, order, fbranch, internal, loops, pos;
  static thread_local std::vector<char> cyc, nontriv;
  viv.clear(); uiv.clear(); div.clear(); sp.clear(); bp.clear(); srcs.clear(); sinks.clear(); loopk.clear(); ce.clear(); loops.clear();
  is_par.assign(n, 0);
  int count = 0;
  int lapi = 0;
  auto lap = [&]() { if (lapi < 12) g2s_post_laps[lapi++] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  lap();
  for (uint32_t i = 0; i < n; i++) {
    const SegRec& s = sg[i];
    const int ts = seg_ts(s);
    if (ts < 0) continue;
    const uint32_t idx = s.node >> 1;
    const bool up = (s.node & 1u) == 0;
    const uint32_t lo = up ? idx : idx - (uint32_t)ts, hi = up ? idx + (uint32_t)ts : idx;
    viv.push_back(((uint64_t)lo << 32) | hi);
    bp.push_back(lo);
    bp.push_back(hi);
    if (ts > 0) (up ? uiv : div).push_back(((uint64_t)lo << 32) | (hi - 1u));  // chain edges by the lower index of their two k-mers
    if (s.flags & G2S_SUB_SOURCE) srcs.push_back(idx);                                            // :1303-1305
    else {
      uint32_t ps[4];
      const int np = seg_parents(s, ps);
      for (int x = 0; x < np; x++) {                                                              // :1283-1297
        const SegRec& q = sg[ps[x]];
        const uint32_t last = seg_state(q, (int)(q.depth_len >> 16) - 1) >> 1;
        sp.push_back(((uint64_t)last << 32) | idx);
        if (!is_par[ps[x]]) { is_par[ps[x]] = 1; bp.push_back(last); }
      }
    }
    if (sinkpos[i] >= 0 && sinkpos[i] <= ts) {                                                    // :1216-1226 / :1248-1256
      const uint32_t x = seg_state(s, sinkpos[i]) >> 1;
      sinks.push_back(x);
      bp.push_back(x);
      count = sat_add(count, (int)s.cnt);
    }
  }
  lap();
  if (p.all_paths) out->count = count;  // recount (:1189-1191); -best-only keeps the phase C count
  merge_intervals(&uiv, true);
  merge_intervals(&div, true);
  merge_intervals(&viv, false);  // vertices: overlapping intervals merge (adjacent ones may: whether an edge joins them is looked up)
  uint64_t V = 2;
  for (const uint64_t a : viv) V += (uint64_t)((uint32_t)a - (uint32_t)(a >> 32)) + 1u;
  radix_sort(bp.data(), bp.size(), 0xFFFFFFFFu);
  bp.erase(std::unique(bp.begin(), bp.end()), bp.end());
  radix_sort(srcs.data(), srcs.size(), 0xFFFFFFFFu);
  srcs.erase(std::unique(srcs.begin(), srcs.end()), srcs.end());  // boost::edge(u, v).second de-duplication ...
  radix_sort(sinks.data(), sinks.size(), 0xFFFFFFFFu);
  sinks.erase(std::unique(sinks.begin(), sinks.end()), sinks.end());
  lap();
  // ---- runs: every cut k-mer alone, and what lies between two cuts
  std::vector<SegRun>& runs = out->runs;
  runs.clear();
  runs.reserve(2 * bp.size() + 2);
  bp_run.assign(bp.size(), -1);
  {
    size_t b = 0;
    for (const uint64_t a : viv) {
      const uint32_t a_lo = (uint32_t)(a >> 32), a_hi = (uint32_t)a;
      uint32_t next = a_lo;  // first k-mer not yet in a run
      while (b < bp.size() && bp[b] < a_lo) b++;
      for (; b < bp.size() && bp[b] <= a_hi; b++) {
        if (bp[b] > next) runs.push_back(SegRun{next, bp[b] - 1u, 0});
        bp_run[b] = (int)runs.size();
        runs.push_back(SegRun{bp[b], bp[b], 0});
        next = bp[b] + 1u;
      }
      if (next <= a_hi) runs.push_back(SegRun{next, a_hi, 0});  // (cannot happen: interval ends are cuts)
    }
  }
  const int R = (int)runs.size(), NV = R + 2;  // node 0 = sink, 1 = source, 2 + r = run r
  auto node_search = [&](uint32_t x) -> int {
    size_t lo = 0, hi = runs.size();
    while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (runs[mid].lo <= x) lo = mid + 1; else hi = mid; }
    return 2 + (int)(lo - 1);
  };
  // the node of k-mer x, for queries that never decrease (every end of an edge is a cut: a run of its own)
  struct CutSweep {
    const std::vector<uint32_t>& bp; const std::vector<int>& bp_run; size_t p = 0;
    int at(uint32_t x) { while (p < bp.size() && bp[p] < x) p++; return (p < bp.size() && bp[p] == x) ? bp_run[p] : -1; }
  };
  lap();
  // ---- edges between nodes, one per distinct edge of the reference's graph; edges inside a run are counted
  internal.assign((size_t)NV, 0);
  cyc.assign((size_t)NV, 0);
  uint64_t e_all = 0;
  {
    IntervalSweep su(uiv), sd(div);
    for (int r = 0; r < R; r++) {
      const uint32_t L = runs[(size_t)r].hi - runs[(size_t)r].lo + 1u;
      if (L > 1u) {
        const bool u = su.has(runs[(size_t)r].lo), d = sd.has(runs[(size_t)r].lo);
        internal[(size_t)(2 + r)] = (int)((u ? L - 1u : 0u) + (d ? L - 1u : 0u));
        cyc[(size_t)(2 + r)] = u && d;  // covered in both directions: its k-mers are one strong component
        e_all += (uint64_t)internal[(size_t)(2 + r)];
      }
      if (r + 1 < R && runs[(size_t)r].hi + 1u == runs[(size_t)r + 1].lo) {  // the chain edge(s) into the next run
        const uint32_t x = runs[(size_t)r].hi;
        if (su.has(x)) ce.emplace_back(2 + r, 2 + r + 1);
        if (sd.has(x)) ce.emplace_back(2 + r + 1, 2 + r);
      }
    }
  }
  {
    // the other edges, in the order of their heads first: those that double a chain's own edge go (the
    // de-duplication again), self loops (a homopolymer k-mer: never between components) are set aside, the
    // head becomes its node; then in the order of their tails (stable: equal edges are neighbours), the tail likewise
    radix_sort(sp.data(), sp.size(), kLo32);
    IntervalSweep su(uiv), sd(div);
    CutSweep cb{bp, bp_run};
    size_t w = 0;
    for (const uint64_t e : sp) {
      const uint32_t a = (uint32_t)(e >> 32), b = (uint32_t)e;
      if (b == a + 1u && su.has(a)) continue;
      if (a == b + 1u && sd.has(b)) continue;
      if (a == b) { loopk.push_back(a); continue; }
      int rb = cb.at(b);
      if (rb < 0) rb = node_search(b) - 2;
      sp[w++] = ((uint64_t)a << 32) | (uint32_t)rb;
    }
    sp.resize(w);
    radix_sort(sp.data(), sp.size(), kHi32);
    CutSweep ca{bp, bp_run};
    uint64_t prev = ~0ull;
    for (const uint64_t e : sp) {
      if (e == prev) continue;
      prev = e;
      const uint32_t a = (uint32_t)(e >> 32);
      int ra = ca.at(a);
      if (ra < 0) ra = node_search(a) - 2;
      ce.emplace_back(2 + ra, 2 + (int)(uint32_t)e);
    }
    radix_sort(loopk.data(), loopk.size(), 0xFFFFFFFFu);
    loopk.erase(std::unique(loopk.begin(), loopk.end()), loopk.end());
    CutSweep cl{bp, bp_run};
    for (const uint32_t x : loopk) { const int r = cl.at(x); loops.push_back(r < 0 ? node_search(x) : 2 + r); }
    CutSweep cs{bp, bp_run};
    for (const uint32_t x : srcs) { const int r = cs.at(x); ce.emplace_back(1, r < 0 ? node_search(x) : 2 + r); }
    CutSweep ck{bp, bp_run};
    for (const uint32_t x : sinks) { const int r = ck.at(x); ce.emplace_back(r < 0 ? node_search(x) : 2 + r, 0); }
  }
  e_all += (uint64_t)ce.size() + (uint64_t)loops.size();
  lap();
  // ---- strong components of the node graph
  off.assign((size_t)NV + 1, 0);
  adj.resize(ce.size());
  for (auto& ed : ce) off[(size_t)ed.first + 1]++;
  for (int i = 0; i < NV; i++) off[(size_t)i + 1] += off[(size_t)i];
  {
    pos.assign(off.begin(), off.end() - 1);
    for (auto& ed : ce) adj[(size_t)pos[(size_t)ed.first]++] = ed.second;
  }
  lap();
  const int nc = strong_components(NV, off, adj, &comp);
  lap();
  cnodes.assign((size_t)nc, 0);
  cweight.assign((size_t)nc, 0);
  nontriv.assign((size_t)nc, 0);
  for (int x = 0; x < NV; x++) {
    const int c = comp[(size_t)x];
    cnodes[(size_t)c]++;
    cweight[(size_t)c] += x < 2 ? 1 : (int)(runs[(size_t)(x - 2)].hi - runs[(size_t)(x - 2)].lo + 1u);
    if (cyc[(size_t)x]) nontriv[(size_t)c] = 1;
  }
  int nontrivial = 0;
  uint64_t size_nontrivial = 0;
  for (int c = 0; c < nc; c++) {
    if (cnodes[(size_t)c] > 1) nontriv[(size_t)c] = 1;  // (a cycle through a chain takes all of the chain)
    if (nontriv[(size_t)c]) { nontrivial++; size_nontrivial += (uint64_t)cweight[(size_t)c]; }
  }
  uint64_t loops_trivial = 0;
  for (int x : loops) if (!nontriv[(size_t)comp[(size_t)x]]) loops_trivial++;  // :1385-1402
  // ---- the contracted multigraph: one edge per edge between components (:1342-1378)
  din.assign((size_t)nc, 0);
  dout.assign((size_t)nc, 0);
  uint64_t fe = 0;
  for (int x = 2; x < NV; x++) if (!nontriv[(size_t)comp[(size_t)x]]) fe += (uint64_t)internal[(size_t)x];
  foff.assign((size_t)nc + 1, 0);
  for (auto& ed : ce) {
    const int a = comp[(size_t)ed.first], b = comp[(size_t)ed.second];
    if (a == b) continue;
    fe++;
    dout[(size_t)a]++;
    din[(size_t)b]++;
    foff[(size_t)a + 1]++;
  }
  out->sub[0] = V;
  out->sub[1] = e_all - loops_trivial;
  out->sub[2] = (uint64_t)nontrivial;
  out->sub[3] = size_nontrivial;
  out->sub[4] = V + (uint64_t)nontrivial - size_nontrivial;
  out->sub[5] = fe;
  // ---- topological order of the components (Kahn), then the branch rule (:1420-1434).  A run outside the
  // components of several vertices is a row of vertices with one edge in and one out each: the count is the
  // same in front of every one of them.
  for (int c = 0; c < nc; c++) foff[(size_t)c + 1] += foff[(size_t)c];
  fadj.resize(foff[(size_t)nc]);
  {
    pos.assign(foff.begin(), foff.end() - 1);
    for (auto& ed : ce) {
      const int a = comp[(size_t)ed.first], b = comp[(size_t)ed.second];
      if (a != b) fadj[(size_t)pos[(size_t)a]++] = b;
    }
  }
  indeg = din;
  order.clear();
  for (int c = 0; c < nc; c++) if (indeg[(size_t)c] == 0) order.push_back(c);
  for (size_t qi = 0; qi < order.size(); qi++) {
    const int u = order[qi];
    for (int x = foff[(size_t)u]; x < foff[(size_t)u + 1]; x++)
      if (--indeg[(size_t)fadj[(size_t)x]] == 0) order.push_back(fadj[(size_t)x]);
  }
  fbranch.assign((size_t)nc, 0);
  int bc = 1;
  for (int u : order) {
    if (din[(size_t)u] >= 1 || dout[(size_t)u] >= 1) {
      if (din[(size_t)u] > 1) bc -= din[(size_t)u] - 1;
      fbranch[(size_t)u] = bc;
      if (dout[(size_t)u] > 1) bc += dout[(size_t)u] - 1;
    }
  }
  for (int r = 0; r < R; r++) {
    const int c = comp[(size_t)(2 + r)];
    runs[(size_t)r].safe = !nontriv[(size_t)c] && fbranch[(size_t)c] == 1;
  }
  out->sink_safe = !nontriv[(size_t)comp[0]] && fbranch[(size_t)comp[0]] == 1;
  out->run_mode = true;
  lap();
  if (const char* path = getenv("G2S_D2_STATS")) {  // (tools: the shape of the graphs phase D2 works on, a line per closure)
    // longest path of the condensation in components, and counting only components that branch (in or out degree
    // other than one): what a level-synchronous sweep on the device would take
    std::vector<int> lvl((size_t)nc, 0), blvl((size_t)nc, 0);
    int deep = 0, bdeep = 0, nbranch = 0;
    for (int u : order) {
      const bool br = din[(size_t)u] != 1 || dout[(size_t)u] != 1;
      nbranch += br;
      const int mine = lvl[(size_t)u] + 1, bmine = blvl[(size_t)u] + (br ? 1 : 0);
      deep = std::max(deep, mine); bdeep = std::max(bdeep, bmine);
      for (int x = foff[(size_t)u]; x < foff[(size_t)u + 1]; x++) {
        const int w = fadj[(size_t)x];
        lvl[(size_t)w] = std::max(lvl[(size_t)w], mine);
        blvl[(size_t)w] = std::max(blvl[(size_t)w], bmine);
      }
    }
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (FILE* f = fopen(path, "a")) {
      fprintf(f, "segs %u runs %d edges %zu loops %zu comps %d nontrivial %d size_nontrivial %llu levels %d branch_comps %d branch_levels %d\n",
              n, R, ce.size(), loops.size(), nc, nontrivial, (unsigned long long)size_nontrivial, deep, nbranch, bdeep);
      fclose(f);
    }
  }
}

}  // namespace

bool seg_analyze(const FillParams& p, const GapJob& job, const SubView& v, SubPrep* out, void* scratch) {
  const GapOut& go = *v.out;
  out->count = go.c_count;
  out->phase_d = go.c_count > 0 && go.n_len > 0;  // :1169
  out->seg_mode = true;
  if (!out->phase_d) return true;
  const uint32_t n = v.n_segs;
  const SegRec* sg = v.segs;
  const int lmf = job.lmf, rmf = job.rmf;
  const uint32_t* targets = job.targets();
  const bool want_s = !p.skip_confident;
  const uint32_t sinknode = (want_s && p.all_paths && rmf >= 1) ? targets[rmf - 1] : kInvalidNode;  // Q3/Q4
  const int lo_sink = std::max(0, lmf + job.g - p.d_err);
  const uint32_t reached = targets[go.reached_j];
  const bool t_is_s = want_s && !p.all_paths;  // -best-only: the traceback starts are the sinks
  if (!scratch) { out->own_seg.resize(3 * (size_t)n + 1); scratch = out->own_seg.data(); }
  static_assert(sizeof(SegInfo) == 16, "SegInfo layout");
  SegInfo* si = out->seg = (SegInfo*)scratch;
  out->s_iv = (std::pair<uint32_t, uint32_t>*)(si + n);
  out->n_iv = 0;
  for (uint32_t i = 0; i < n; i++) si[i] = SegInfo{-2, -2, 0, 0, 0, 0};
  static thread_local std::vector<int> sinkpos, outs;
  sinkpos.assign(n, -1);
  // ---- own positions: sinks and traceback starts
  for (uint32_t i = 0; i < n; i++) {
    const SegRec& s = sg[i];
    const int d0 = (int)(s.depth_len & 0xFFFFu), len = (int)(s.depth_len >> 16);
    uint32_t ps[4];
    si[i].npar = (uint8_t)seg_parents(s, ps);
    const int pk = seg_find(s.node, len, sinknode);
    if (pk >= 0 && d0 + pk >= lo_sink) sinkpos[i] = pk;
    const int pt = seg_find(s.node, len, reached);
    if (pt >= 0) {
      for (int j = 0; j < go.n_len && j < 2; j++)
        if (d0 + pt == go.len[j]) {
          if (out->start_seg[j] < 0) { out->start_seg[j] = (int)i; out->start_t[j] = pt; }
          if (t_is_s) sinkpos[i] = pt;
        }
    }
  }
  // ---- stop depths of the traceback closure, parents first (= descending index)
  for (int64_t i = (int64_t)n - 1; i >= 0; i--) {
    const SegRec& s = sg[i];
    const int tt = seg_tt(s);
    if (tt < 0) continue;
    const int d0 = (int)(s.depth_len & 0xFFFFu);
    if (s.flags & G2S_SUB_SOURCE) { si[i].lo = si[i].hi = d0; continue; }  // :1455-1462
    uint32_t ps[4];
    const int np = seg_parents(s, ps);
    if (np == 0) { si[i].lo = -1; si[i].hi = 1 << 30; continue; }
    if (np > 1) out->has_choice = true;
    int l = 1 << 30, h = -1;
    for (int x = 0; x < np; x++) {
      const SegInfo& q = si[ps[x]];
      if (q.lo < 0) { l = -1; h = 1 << 30; } else { l = std::min(l, q.lo); h = std::max(h, q.hi); }
    }
    si[i].lo = l; si[i].hi = h;
  }
  for (int j = 0; j < go.n_len && j < 2; j++) {
    const int i = out->start_seg[j];
    out->stop_depth[j] = (i >= 0 && si[(size_t)i].lo >= 0 && si[(size_t)i].lo == si[(size_t)i].hi) ? si[(size_t)i].lo : -1;
  }
  if (!want_s) return true;  // no D1/D2 with -all-upper (:1181)

  // ---- the S closure: every k-mer at one depth only?  (index intervals must not overlap)
  std::pair<uint32_t, uint32_t>* iv = out->s_iv;
  uint32_t niv = 0;
  uint64_t n_s = 0, edges = 0;
  int count_s = 0, src_out = 0, sink_in = 0;
  outs.assign(n, 0);
  static thread_local std::vector<std::pair<uint32_t, uint32_t>> hi_of;
  hi_of.clear();
  for (uint32_t i = 0; i < n; i++) {
    const SegRec& s = sg[i];
    const int ts = seg_ts(s);
    if (ts < 0) continue;
    const uint32_t idx = s.node >> 1;
    const uint32_t lo = (s.node & 1u) ? idx - (uint32_t)ts : idx;
    iv[niv++] = std::make_pair(lo, i);
    n_s += (uint64_t)ts + 1;
    edges += (uint64_t)ts;  // interior edges
    if (s.flags & G2S_SUB_SOURCE) { src_out++; edges++; }
    else {
      uint32_t ps[4];
      const int np = seg_parents(s, ps);
      edges += (uint64_t)np;
      for (int x = 0; x < np; x++) outs[ps[x]]++;
    }
    if (sinkpos[i] >= 0 && sinkpos[i] <= ts) { sink_in++; edges++; count_s = sat_add(count_s, (int)s.cnt); }
  }
  out->n_iv = niv;
  static_assert(sizeof(iv[0]) == 8, "interval records are (first index, segment) words");
  radix_sort((uint64_t*)iv, (size_t)niv, kLo32);  // by first index (the low word); equal ones stay in segment order
  for (size_t x = 1; x < niv; x++) {
    const SegRec& a = sg[iv[x - 1].second];
    const uint32_t a_hi = iv[x - 1].first + (uint32_t)seg_ts(a);
    if (iv[x].first <= a_hi) {  // a k-mer at two depths: the vertices are k-mers, not states
      static const bool state_d2 = getenv("G2S_STATE_D2") != nullptr;  // (tests: the per-state analysis instead)
      if (state_d2) { out->seg_mode = false; return false; }
      seg_analyze_runs(p, v, out, sinkpos);
      return true;
    }
  }
  if (p.all_paths) out->count = count_s;
  out->sub[0] = n_s + 2; out->sub[1] = edges; out->sub[2] = 0; out->sub[3] = 0; out->sub[4] = n_s + 2; out->sub[5] = edges;
  // ---- branch rule (:1411-1434) over a topological order: parents first, a segment's states in order.
  // Interior states have one edge in and one out, so the running count only moves at a segment's
  // entry (in-degree), at a sink inside it, and at its last S state (out-degree).
  int bc = 1;
  if (src_out > 1) bc += src_out - 1;  // the source pseudo-vertex comes first
  for (int64_t i = (int64_t)n - 1; i >= 0; i--) {
    const SegRec& s = sg[i];
    const int ts = seg_ts(s);
    if (ts < 0) continue;
    const int len = (int)(s.depth_len >> 16);
    const int din = (s.flags & G2S_SUB_SOURCE) ? 1 : (int)si[i].npar;
    if (din > 1) bc -= din - 1;
    si[i].safe_a = bc == 1;
    si[i].split = (int16_t)ts;
    si[i].safe_b = si[i].safe_a;
    const int sp = (sinkpos[i] >= 0 && sinkpos[i] <= ts) ? sinkpos[i] : -1;
    if (sp >= 0 && sp < ts) { bc += 1; si[i].split = (int16_t)sp; si[i].safe_b = bc == 1; }  // out-degree 2: next state + sink
    const int dout = (ts == len - 1 ? outs[(size_t)i] : 0) + (sp == ts ? 1 : 0);
    if (dout > 1) bc += dout - 1;
  }
  out->sink_safe = false;
  if (sink_in >= 1) { if (sink_in > 1) bc -= sink_in - 1; out->sink_safe = bc == 1; }
  return true;
}

void seg_traceback(const Graph& g, const FillParams& p, const GapJob& job, const SubView& v, const SubPrep& prep,
                   const uint32_t* rands, char* buf, g2s_result* res, bool packed12) {
  static const char kUp[4] = {'A', 'C', 'T', 'G'}, kLow[4] = {'a', 'c', 't', 'g'};  // GATB codes (kmer.hpp)
  const GapOut& go = *v.out;
  const int lmf = job.lmf, k = p.k;
  res->right_fuz = go.reached_j;  // :1171
  res->flags |= G2S_GAP_PHASE_D;
  int draws = 0;
  // (raw words: the value is word >> 1; packed: the value's remainder by 12 stands for it)
  auto draw = [&]() -> uint32_t {
    uint32_t r = 0u;
    if (rands) r = packed12 ? ((rands[(size_t)draws >> 3] >> (4 * (draws & 7))) & 15u) << 1 : rands[draws];
    draws++;
    return r;
  };
  const int pick = (int)((draw() >> 1) % (uint32_t)go.n_len);  // :1440
  int d2 = go.len[pick];
  int last_solid = d2;
  int i = prep.start_seg[pick], t = prep.start_t[pick];
  buf[d2] = '\0';
  res->count = prep.count;
  const uint8_t* lastnt = g.lastnt.data();
  g.ensure_lastch();
  const char* chu = g.lastch_up.data();
  const char* chd = g.lastch_dn.data();
  while (d2 >= 0 && i >= 0) {
    const SegRec& s = v.segs[i];
    const int d0 = (int)(s.depth_len & 0xFFFFu);
    // ---- the states t, t-1, ..., 1 of this segment: one choice each (still drawn, :1513), written
    // in runs that share their safe bit (:1466-1468); state 0 is left for the parent choice below
    if (t > 0) {
      const int ts = seg_ts(s);
      const int split = p.skip_confident ? t : (prep.seg ? (int)prep.seg[(size_t)i].split : (int)s.pad);
      const bool sfa = prep.seg ? prep.seg[(size_t)i].safe_a != 0 : (s.ts_tt & 0x8000u) != 0;
      const bool sfb = prep.seg ? prep.seg[(size_t)i].safe_b != 0 : (s.ts_tt & 0x80000000u) != 0;
      int pos = t;
      while (pos > 0) {
        int lo_run;  // the run is [lo_run, pos]
        bool sf;
        if (p.skip_confident) { lo_run = 1; sf = true; }
        else if (prep.run_mode) {  // the states whose k-mers lie in the run of this one share its verdict
          const uint32_t idx0 = s.node >> 1;
          const uint32_t x = (s.node & 1u) ? idx0 - (uint32_t)pos : idx0 + (uint32_t)pos;
          const SegRun* r = run_of(prep, x);
          if (!r) { lo_run = pos; sf = prep.sink_safe; }
          else {
            sf = r->safe != 0;
            lo_run = (s.node & 1u) ? std::max(1, (int)idx0 - (int)r->hi) : std::max(1, (int)r->lo - (int)idx0);
          }
        }
        else if (pos > ts) { lo_run = pos; sf = seg_safe(v, prep, (uint32_t)i, pos); }  // outside the subgraph (Q5): state by state
        else if (pos > split) { lo_run = std::max(1, split + 1); sf = sfb; }
        else { lo_run = 1; sf = sfa; }
        // the bases of states lo_run .. pos are consecutive bytes of the unitig's last-base string (ascending
        // for an upward segment, descending for a downward one): copied, then lower-cased where the rule says so
        const bool up = (s.node & 1u) == 0;
        const uint32_t idx0 = s.node >> 1;
        const int cnt = pos - lo_run + 1;
        char* dst = buf + (d0 + lo_run - 1);
        if (up) memcpy(dst, chu + ((size_t)idx0 + (size_t)lo_run), (size_t)cnt);
        else {
          const char* src = chd + ((size_t)idx0 - (size_t)pos);  // k-mer idx0 - q for q = pos down to lo_run
          for (int x = 0; x < cnt; x++) dst[cnt - 1 - x] = src[x];
        }
        if (sf) last_solid = d0 + lo_run;
        else {  // (:1466-1468) lower case unless within k of the last safe base: d0 + q > last_solid - k
          const int qmax = std::min(pos, last_solid - k - d0);
          for (int q = lo_run; q <= qmax; q++) dst[q - lo_run] |= 0x20;
        }
        pos = lo_run - 1;
      }
      draws += t;
      d2 -= t;
      t = 0;
    }
    if (s.flags & G2S_SUB_SOURCE) {  // :1455-1462
      res->left_fuz = lmf - d2;
      break;
    }
    if (d2 > 0) {
      if (p.skip_confident || seg_safe(v, prep, (uint32_t)i, 0)) last_solid = d2;
      const uint8_t c = lastnt[s.node];
      buf[d2 - 1] = (d2 > last_solid - k) ? kUp[c] : kLow[c];
      uint32_t back[4];
      const int nb = seg_parents(s, back);
      if (nb > 1 && !(s.flags & G2S_SEG_ORDERED)) {  // GATB predecessor order: predecessors(v)[slot] is the parent p whose p^1 ends with base `slot`
        int64_t by_slot[4] = {-1, -1, -1, -1};
        for (int x = 0; x < nb; x++) {
          const SegRec& q = v.segs[back[x]];
          by_slot[lastnt[seg_state(q, (int)(q.depth_len >> 16) - 1) ^ 1u]] = (int64_t)back[x];
        }
        int w = 0;
        for (int nt = 0; nt < 4; nt++) if (by_slot[nt] >= 0) back[w++] = (uint32_t)by_slot[nt];
      }
      if (nb == 0) {  // :1493-1510
        res->backtrace_depth = d2;  // :1494 (the line itself: g2s_backtrace_text)
        res->backtrace_final_d = go.final_d;
        res->flags |= G2S_GAP_BACKTRACE_FAIL;
        res->count = 0;
        buf[lmf] = '\0';
        break;
      }
      const uint32_t rv = draw() >> 1;
      i = (int)(nb == 1 ? back[0] : back[rv % (uint32_t)nb]);
      t = (int)(v.segs[i].depth_len >> 16) - 1;  // a child in the closure puts the whole parent there
    }
    d2--;
  }
  res->draws = draws;
}

void seg_stop_depths(const SubView& v, int32_t* out) {
  const uint32_t n = v.n_segs;
  for (int64_t i = (int64_t)n - 1; i >= 0; i--) {  // parents first (= descending index)
    const SegRec& s = v.segs[i];
    int32_t& lo = out[2 * i];
    int32_t& hi = out[2 * i + 1];
    lo = -2; hi = -2;
    if (seg_tt(s) < 0) continue;
    if (s.flags & G2S_SUB_SOURCE) { lo = hi = (int32_t)(s.depth_len & 0xFFFFu); continue; }
    uint32_t ps[4];
    const int np = seg_parents(s, ps);
    if (np == 0) { lo = -1; hi = 1 << 30; continue; }
    int l = 1 << 30, h = -1;
    for (int x = 0; x < np; x++) {
      const int32_t ql = out[2 * ps[x]], qh = out[2 * ps[x] + 1];
      if (ql < 0) { l = -1; h = 1 << 30; } else { l = std::min(l, (int)ql); h = std::max(h, (int)qh); }
    }
    lo = l; hi = h;
  }
}

int seg_count_draws(const Graph& g, const SubView& v, const SubPrep& prep, const uint32_t* rands) {
  const GapOut& go = *v.out;
  int draws = 0;
  const int pick = (int)((rands[draws++] >> 1) % (uint32_t)go.n_len);
  int d2 = go.len[pick];
  int i = prep.start_seg[pick], t = prep.start_t[pick];
  while (d2 >= 0 && i >= 0) {
    const SegRec& s = v.segs[i];
    if (t == 0 && (s.flags & G2S_SUB_SOURCE)) break;
    {  // every traceback through this segment's entry stops at one depth: one draw per level down to it
      int lo = -1, hi = -2;
      if (prep.seg) { lo = prep.seg[i].lo; hi = prep.seg[i].hi; }
      else if (prep.stop) { lo = prep.stop[2 * i]; hi = prep.stop[2 * i + 1]; }
      if (lo >= 0 && lo == hi && d2 >= lo) { draws += d2 - lo; break; }
    }
    if (d2 > 0) {
      if (t > 0) {  // the rest of the segment is drawn base by base with one choice each
        const int run = std::min(t, d2);
        draws += run; d2 -= run; t -= run;
        continue;
      }
      uint32_t back[4];
      const int nb = seg_parents(s, back);
      if (nb == 0) break;
      if (nb > 1 && !(s.flags & G2S_SEG_ORDERED)) {
        int64_t by_slot[4] = {-1, -1, -1, -1};
        for (int x = 0; x < nb; x++) {
          const SegRec& q = v.segs[back[x]];
          by_slot[g.lastnt[seg_state(q, (int)(q.depth_len >> 16) - 1) ^ 1u]] = (int64_t)back[x];
        }
        int w = 0;
        for (int nt = 0; nt < 4; nt++) if (by_slot[nt] >= 0) back[w++] = (uint32_t)by_slot[nt];
      }
      const uint32_t rv = rands[draws++] >> 1;
      i = (int)(nb == 1 ? back[0] : back[rv % (uint32_t)nb]);
      t = (int)(v.segs[i].depth_len >> 16) - 1;
    }
    d2--;
  }
  return draws;
}

}  // namespace g2s
