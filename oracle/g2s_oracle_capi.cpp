// oracle/g2s_oracle_capi.cpp — extern "C" face of the CPU oracle for ctypes.
// TEST INFRASTRUCTURE ONLY (see g2s_oracle.hpp).  Parity unpinned by the reference.
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <sstream>
#include <thread>
#include <vector>

#include "g2s_oracle.hpp"

using namespace orc;

extern "C" {

struct orc_info {
  uint64_t sub[6];      // vertices, edges, nontrivial, size_nontrivial, vertices_final, edges_final
  uint64_t ctr[6];      // xA sA xB sB xD sD
  int32_t phaseC_count, n_lengths, lengths[2], reached_fuz, draws, q7, backtrace_failed, mem_exceeded;
  int32_t final_d, pad;
  int32_t max_border_a, max_border_b;
};

struct orc_params {
  int32_t k, solid, d_err, max_fuz;
  double max_mem_gb;
  int32_t skip_confident, unique_paths, all_paths, randseed, nb_cores;
};

struct orc_summary {
  int32_t gaps, filled, q7_gaps, pad;
  uint64_t ctr[6];
  double fill_seconds;
};

static std::vector<std::string> split_csv(const char* s) {
  std::vector<std::string> v;
  std::stringstream ss(s);
  std::string f;
  while (std::getline(ss, f, ',')) v.push_back(f);
  return v;
}

void* orc_graph_from_files(const char* reads_csv, int k, int solid) {
  return graph_from_files(split_csv(reads_csv), k, solid);
}
void* orc_graph_from_seqs(const char** seqs, int n, int k, int solid) {
  std::vector<std::string> v;
  for (int i = 0; i < n; i++) v.push_back(seqs[i]);
  return graph_from_seqs(v, k, solid);
}
void orc_graph_free(void* g) { graph_free((GraphBase*)g); }
uint64_t orc_graph_num_kmers(void* g) { return graph_num_kmers((GraphBase*)g); }

void* orc_rng_new(unsigned seed) { GlibcRand* r = new GlibcRand(); r->seed(seed); return r; }
void orc_rng_free(void* r) { delete (GlibcRand*)r; }
int orc_rng_next(void* r) { return ((GlibcRand*)r)->next(); }
// discard n draws (tests re-synchronise the oracle's stream with the product's after a gap
// whose draw count legitimately differs: SURVEY Q7)
void orc_rng_skip(void* r, unsigned long long n) { for (unsigned long long i = 0; i < n; i++) ((GlibcRand*)r)->next(); }

static void pack_info(const FillInfo& fi, orc_info* o) {
  o->sub[0] = fi.sub.vertices; o->sub[1] = fi.sub.edges; o->sub[2] = fi.sub.nontrivial_components;
  o->sub[3] = fi.sub.size_nontrivial_components; o->sub[4] = fi.sub.vertices_final; o->sub[5] = fi.sub.edges_final;
  o->ctr[0] = fi.ctr.xA; o->ctr[1] = fi.ctr.sA; o->ctr[2] = fi.ctr.xB; o->ctr[3] = fi.ctr.sB;
  o->ctr[4] = fi.ctr.xD; o->ctr[5] = fi.ctr.sD;
  o->phaseC_count = fi.phaseC_count; o->n_lengths = fi.n_lengths;
  o->lengths[0] = fi.lengths[0]; o->lengths[1] = fi.lengths[1];
  o->reached_fuz = fi.reached_fuz; o->draws = fi.draws; o->q7 = fi.q7;
  o->backtrace_failed = fi.backtrace_failed; o->mem_exceeded = fi.mem_exceeded;
  o->final_d = fi.final_d; o->pad = 0;
  o->max_border_a = fi.max_border_a; o->max_border_b = fi.max_border_b;
}

// One fill_gap call.  `fill` must hold gap_len + k + gap_err + lmf + rmf + 3 bytes (Gap2Seq.cpp:374).
int orc_fill_gap(void* g, void* rng, const char* left, const char* right, int gap_len, int gap_err, int lmf, int rmf,
                 long long max_mem, int skip_confident, int all_paths, char* fill, int* left_fuz, int* right_fuz,
                 orc_info* out) {
  GraphBase* G = (GraphBase*)g;
  FillInfo fi;
  SubgraphStats st;
  *left_fuz = 0;
  *right_fuz = 0;
  int r = fill_gap(G, *(GlibcRand*)rng, left, right, gap_len, graph_k(G), gap_err, lmf, rmf, left_fuz, right_fuz,
                   max_mem, fill, skip_confident != 0, all_paths != 0, &st, &fi);
  if (out) pack_info(fi, out);
  return r;
}

// Same as orc_fill_gap, plus a malloc'ed text dump of the left DP table
// ("ORIENTED_KMER depth count" per line) for unit tests of host post-processing.
int orc_fill_gap_dump(void* g, void* rng, const char* left, const char* right, int gap_len, int gap_err, int lmf,
                      int rmf, long long max_mem, int skip_confident, int all_paths, char* fill, int* left_fuz,
                      int* right_fuz, orc_info* out, char** states) {
  GraphBase* G = (GraphBase*)g;
  FillInfo fi;
  std::string dump;
  fi.dump_states = &dump;
  SubgraphStats st;
  *left_fuz = 0;
  *right_fuz = 0;
  int r = fill_gap(G, *(GlibcRand*)rng, left, right, gap_len, graph_k(G), gap_err, lmf, rmf, left_fuz, right_fuz,
                   max_mem, fill, skip_confident != 0, all_paths != 0, &st, &fi);
  if (out) pack_info(fi, out);
  if (states) { *states = (char*)malloc(dump.size() + 1); memcpy(*states, dump.c_str(), dump.size() + 1); }
  return r;
}

// Timed CPU baseline: fill_gap over a list of gaps, pulled one at a time by
// `nthreads` threads (gap-level parallelism like the reference's dispatcher,
// Gap2Seq.cpp:296-306; each thread has its own rand() stream, as nothing is compared).
// Returns wall seconds of the fill loop only.
double orc_time_fill_batch(void* g, const char** lefts, const char** rights, const int* gap_lens, const int* lmfs,
                           const int* rmfs, int n, int gap_err, int skip_confident, int all_paths, int nthreads,
                           int* filled_out, uint64_t* ctr_out) {
  GraphBase* G = (GraphBase*)g;
  const int k = graph_k(G);
  if (nthreads < 1) nthreads = 1;
  std::vector<int> filled((size_t)nthreads, 0);
  std::vector<Counters> ctrs((size_t)nthreads);
  std::atomic<int> next_gap(0);
  auto work = [&](int t) {
    GlibcRand rng;
    rng.seed(1 + (unsigned)t);
    while (true) {  // shared iterator, one gap at a time (Gap2Seq.cpp:313-323)
      const int i = next_gap.fetch_add(1);
      if (i >= n) break;
      std::vector<char> buf((size_t)(gap_lens[i] + k + gap_err + lmfs[i] + rmfs[i] + 3));
      int lf = 0, rf = 0;
      SubgraphStats st;
      FillInfo fi;
      int c = fill_gap(G, rng, lefts[i], rights[i], gap_lens[i], k, gap_err, lmfs[i], rmfs[i], &lf, &rf,
                       (long long)1 << 60, buf.data(), skip_confident != 0, all_paths != 0, &st, &fi);
      if (c > 0) filled[(size_t)t]++;
      Counters& cc = ctrs[(size_t)t];
      cc.xA += fi.ctr.xA; cc.sA += fi.ctr.sA; cc.xB += fi.ctr.xB; cc.sB += fi.ctr.sB; cc.xD += fi.ctr.xD; cc.sD += fi.ctr.sD;
    }
  };
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int total = 0;
  uint64_t c6[6] = {0, 0, 0, 0, 0, 0};
  for (int t = 0; t < nthreads; t++) {
    total += filled[(size_t)t];
    c6[0] += ctrs[(size_t)t].xA; c6[1] += ctrs[(size_t)t].sA; c6[2] += ctrs[(size_t)t].xB;
    c6[3] += ctrs[(size_t)t].sB; c6[4] += ctrs[(size_t)t].xD; c6[5] += ctrs[(size_t)t].sD;
  }
  if (filled_out) *filled_out = total;
  if (ctr_out) for (int i = 0; i < 6; i++) ctr_out[i] = c6[i];
  return secs;
}

static Params to_params(const orc_params* p) {
  Params q;
  q.k = p->k; q.solid = p->solid; q.d_err = p->d_err; q.max_fuz = p->max_fuz; q.max_mem_gb = p->max_mem_gb;
  q.skip_confident = p->skip_confident != 0; q.unique_paths = p->unique_paths != 0; q.all_paths = p->all_paths != 0;
  q.randseed = p->randseed; q.nb_cores = p->nb_cores;
  return q;
}

static char* dup_str(const std::string& s) {
  char* p = (char*)malloc(s.size() + 1);
  memcpy(p, s.c_str(), s.size() + 1);
  return p;
}
void orc_free_str(char* p) { free(p); }

// Scaffold mode on in-memory FASTA text; returns malloc'ed FASTA and log texts.
int orc_execute_scaffolds(void* g, const orc_params* p, const char* reads_label, const char* filled_label,
                          const char* scaffolds_text, char** fasta, char** log, orc_summary* sum) {
  std::string fa, lg;
  ExecSummary es;
  int r = execute_scaffolds((GraphBase*)g, to_params(p), reads_label, filled_label, scaffolds_text, &fa, &lg, &es);
  if (fasta) *fasta = dup_str(fa);
  if (log) *log = dup_str(lg);
  if (sum) {
    sum->gaps = es.gaps; sum->filled = es.filled; sum->q7_gaps = es.q7_gaps; sum->pad = 0;
    sum->ctr[0] = es.ctr.xA; sum->ctr[1] = es.ctr.sA; sum->ctr[2] = es.ctr.xB; sum->ctr[3] = es.ctr.sB;
    sum->ctr[4] = es.ctr.xD; sum->ctr[5] = es.ctr.sD;
    sum->fill_seconds = es.fill_seconds;
  }
  return r;
}

int orc_execute_single(void* g, const orc_params* p, const char* reads_label, const char* filled_label,
                       const char* left, const char* right, int length, char** fasta, char** log) {
  std::string fa, lg;
  int r = execute_single((GraphBase*)g, to_params(p), reads_label, filled_label, left, right, length, &fa, &lg);
  if (fasta) *fasta = dup_str(fa);
  if (log) *log = dup_str(lg);
  return r;
}

}  // extern "C"
