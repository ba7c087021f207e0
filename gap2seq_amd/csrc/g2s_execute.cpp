// gap2seq_amd/csrc/g2s_execute.cpp — host driver around the C ABI that mirrors
// Gap2Seq::execute() after the graph exists
// (/root/reference/src/Gap2Seq.cpp:224-438) and print_statistics (:100-156):
// scaffold scanner, per-gap statistics text, splice, FASTA text.
//
// The reference calls fill_gap gap by gap; here the scanner first collects the
// gaps of ALL records into one batch for the GPU (g2s_fill_batch), then replays
// the reference's splice logic over the results in input order.  Two couplings
// between consecutive gaps of one record survive batching and are kept exact:
//  * `i += right_fuz` after a filled gap (:402,415) can make the next gap
//    ineligible -> g2s_gap.skip_if_prev_right_fuz_gt;
//  * when k < max_fuz the next gap's left_max_fuz (:349) can depend on it ->
//    the batch is cut there and the scan resumes from the exact state.
// Only this file's callers' view of fill_gap is the C ABI: it proves the ABI is
// sufficient for the reference's own driver.
#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/g2s.h"
#include "fastx.hpp"

using namespace g2s;

namespace {

char* dup_text(const std::string& s) {
  char* p = (char*)malloc(s.size() + 1);
  if (p) memcpy(p, s.c_str(), s.size() + 1);
  return p;
}

inline bool is_n(char c) { return c == 'N' || c == 'n'; }

// print_statistics (Gap2Seq.cpp:100-156); `tail` = &buf[left_max_fuz - left_fuz]
void stats_line(std::ostringstream& os, int filledStart, int gapStart, int gapEnd, int paths, const char* tail, int k,
                int lmf, int rmf, int left_fuz, int right_fuz, bool skip_confident, bool unique_paths,
                const g2s_result& r, int gap, const std::string& comment) {
  if (paths > 0 && (!unique_paths || paths == 1)) {
    const int filledLen = (int)strlen(tail) - k;
    if (!skip_confident) {
      int lower = 0, upper = 0;
      for (int j = 0; j < filledLen; j++) {
        if (isupper((unsigned char)tail[j])) upper++; else lower++;
      }
      os << "Scaffold: " << comment << " GapStart: " << gapStart << " GapEnd: " << gapEnd << " GapLength: " << gap
         << " PathsFound: " << paths << " FilledStart: " << filledStart << " FilledEnd: " << filledStart + filledLen
         << " FilledGapLength: " << filledLen << " LeftMaxFuz: " << lmf << " LeftFuz: " << left_fuz
         << " RightMaxFuz: " << rmf << " RightFuz: " << right_fuz << " ConfidentBases: " << upper
         << " TotalBases: " << (upper + lower) << "\n";
      os << "SubgraphStats: Vertices: " << r.vertices << " Edges: " << r.edges
         << " NontrivialStrongComponents: " << r.nontrivial_components
         << " SizeNontrivialStrongComponents: " << r.size_nontrivial_components
         << " VerticesFinal: " << r.vertices_final << " EdgesFinal: " << r.edges_final << "\n";
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

void echo_params(std::ostringstream& os, const g2s_run_opts& o, const g2s_params& p, const char* reads,
                 const char* filled, long long max_mem) {
  // Gap2Seq.cpp:180-191
  os << "k-mer size: " << o.k << "\n";
  os << "Solidity threshold: " << o.solid << "\n";
  os << "Reads file: " << (reads ? reads : "") << "\n";
  os << "Filled scaffolds file: " << (filled ? filled : "") << "\n";
  os << "Distance error: " << p.d_err << "\n";
  os << "Max Fuz: " << o.max_fuz << "\n";
  os << "Max memory: " << max_mem << "\n";
  os << "Skip confident: " << (p.skip_confident ? 1 : 0) << "\n";
  os << "Unique: " << (p.unique_paths ? 1 : 0) << "\n";
  os << "All paths: " << (p.all_paths ? 1 : 0) << "\n";
  os << "Random seed: " << p.randseed << "\n";
}

struct GapEvent {
  size_t rec;
  size_t istart, iend;  // N-run [istart, iend)
  int gap, lmf, rmf, kmer_start;
  bool ok;
  int job;  // index into the batch or -1
};

}  // namespace

// Internal accessors implemented in g2s_api.hip
extern "C" const g2s_graph* g2s_session_graph(const g2s_session* s);
extern "C" int g2s_session_get_params(const g2s_session* s, g2s_params* out);

// The log and the FASTA leave as the records are done: after every batch of about `chunk_gaps` gaps (cut at
// record boundaries; 0 = one batch) what has been decided so far is handed to the two callbacks, in input
// order — the reference prints a gap's statistics when the gap is done (:385), not at the end of the run.
// The rand() stream runs on across the batches and the couplings between gaps stay inside a record, so the
// text is the same for every chunk size.
extern "C" int g2s_execute_scaffolds_stream(g2s_session* s, const g2s_run_opts* o, const char* reads_label,
                                            const char* filled_label, const char* scaffolds_text, size_t chunk_gaps,
                                            g2s_text_fn on_fasta, g2s_text_fn on_log, void* user, int32_t* gaps_out,
                                            int32_t* filled_out) {
  if (!s || !o || !scaffolds_text) return G2S_ERR_ARG;
  g2s_params p;
  g2s_session_get_params(s, &p);
  const int k = g2s_graph_k(g2s_session_graph(s));
  const int max_fuz = o->max_fuz;
  std::ostringstream os;
  std::string fasta;
  const long long max_mem_total = (long long)(o->max_mem_gb * 1024 * 1024 * 1024);  // :170
  echo_params(os, *o, p, reads_label, filled_label, max_mem_total);
  // :302-303: the reference's own line, total / execution units (Gap2Seq-core hands over the host's CPU count when
  // -nb-cores is omitted, as GATB's dispatcher does).  The budget the session APPLIES is g2s_params.max_mem — the same
  // number whenever -nb-cores is given; when it is omitted the command line says on stderr what applies instead.
  os << "Max mem: " << max_mem_total / std::max(1, o->nb_cores) << "\n";

  std::vector<FastxRecord> recs;
  parse_fastx(std::string(scaffolds_text), &recs);
  int gapcount = 0, filledgapcount = 0;

  // exact scanner state (the reference's locals, Gap2Seq.cpp:330-337)
  size_t cur_rec = 0;
  size_t i_exact = 0;
  int prevGapEnd = 0;
  std::string filledSeq;
  bool done = recs.empty();

  // A batch of gaps from scan to replay.  Two may be in flight (g2s_fill_begin / g2s_fill_end): behind a cut at a
  // record boundary the next batch's scan does not depend on this one's results, so it is scanned and begun — its
  // look-ups and fill kernel queued — before this one is ended and replayed; behind a barrier (a gap whose
  // left_max_fuz depends on the previous gap's right_fuz) the next scan waits for the replay.
  struct Chunk {
    std::vector<GapEvent> events;
    std::vector<g2s_gap> jobs;
    std::vector<std::string> flanks;
    size_t sr = 0, si = 0;
    bool barrier = false, soft = false;
    g2s_result* results = nullptr;
    char* arena = nullptr;
    size_t arena_bytes = 0;
    bool begun = false;
  };
  // (result records and fill arena in page-locked memory — a list finished on the device is written there by the
  // kernels themselves; ordinary memory when that cannot be had — two pairs, grow-only, kept across the batches)
  struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    std::vector<char> plain;
    void* get(size_t bytes) {
      if (bytes <= cap && p) return p;
      if (pinned) g2s_host_free(p);
      const size_t want = bytes + bytes / 4 + 4096;
      p = g2s_host_alloc(want);
      pinned = p != nullptr;
      if (!p) { plain.resize(want); p = plain.data(); }
      cap = want;
      return p;
    }
    ~HostBuf() { if (pinned) g2s_host_free(p); }
  } res_buf[2], arena_buf[2];
  size_t nchunks = 0;

  // ---- static scan: collect gaps until the input ends, a barrier is hit or the batch is full ----
  auto scan = [&](size_t start_rec, size_t start_i, int start_prev) -> Chunk* {
    Chunk* c = new Chunk;
    std::vector<GapEvent>& events = c->events;
    std::vector<g2s_gap>& jobs = c->jobs;
    std::vector<std::string>& flanks = c->flanks;  // keeps left/right strings alive
    size_t sr = start_rec, si = start_i;
    int sprev = start_prev;
    bool prev_attempted = false;  // previous gap of this record is in this batch and eligible
    int prev_rmf = 0;
    bool barrier = false, soft = false;
    flanks.reserve(2 * 1024);
    std::vector<std::pair<size_t, size_t>> flank_idx;  // per job: indices into flanks
    while (sr < recs.size() && !barrier) {
      const std::string& seq = recs[sr].seq;
      while (si < seq.size()) {
        if (!is_n(seq[si])) { si++; continue; }
        const size_t istart = si;
        if (prev_attempted) {
          const int hi = std::min((int)istart + k - sprev, max_fuz);
          const int lo = std::min((int)istart + k - (sprev + prev_rmf), max_fuz);
          if (lo != hi) { barrier = true; break; }  // left_max_fuz would depend on the previous right_fuz
        }
        GapEvent ev;
        ev.rec = sr;
        ev.istart = istart;
        ev.lmf = std::min(((int)istart + k) - sprev, max_fuz);  // :349 (Q9)
        ev.kmer_start = (int)istart - k - ev.lmf;
        ev.gap = 0;
        while (si < seq.size() && is_n(seq[si])) { si++; ev.gap++; }
        ev.iend = si;
        ev.rmf = std::min((int)seq.size() - ((int)si + k), max_fuz);  // :360
        ev.ok = si + k + ev.rmf <= seq.size();
        if (ev.rmf < 0) ev.ok = false;  // D2 (Q10): the reference would throw from substr
        for (int j = 0; j < k + ev.rmf && ev.ok; j++)
          if (is_n(seq[si + j])) ev.ok = false;
        ev.job = -1;
        const bool eligible = ev.kmer_start >= sprev && ev.ok;
        if (eligible) {
          ev.job = (int)jobs.size();
          g2s_gap gj;
          memset(&gj, 0, sizeof gj);
          flanks.push_back(seq.substr((size_t)ev.kmer_start, (size_t)(k + ev.lmf)));
          flanks.push_back(seq.substr(si, (size_t)(k + ev.rmf)));
          flank_idx.emplace_back(flanks.size() - 2, flanks.size() - 1);
          gj.left_len = k + ev.lmf;
          gj.right_len = k + ev.rmf;
          gj.gap_len = ev.gap;
          gj.lmf = ev.lmf;
          gj.rmf = ev.rmf;
          gj.skip_if_prev_right_fuz_gt = -1;
          if (prev_attempted && ev.kmer_start - sprev < prev_rmf) gj.skip_if_prev_right_fuz_gt = ev.kmer_start - sprev;
          jobs.push_back(gj);
        }
        events.push_back(ev);
        sprev = (int)si;
        prev_attempted = eligible;
        prev_rmf = std::max(0, ev.rmf);
      }
      if (barrier) break;
      sr++;
      si = 0;
      sprev = 0;
      prev_attempted = false;
      if (chunk_gaps && jobs.size() >= chunk_gaps && sr < recs.size()) { soft = true; break; }  // enough for one batch
    }
    for (size_t j = 0; j < jobs.size(); j++) {
      jobs[j].left = flanks[flank_idx[j].first].c_str();
      jobs[j].right = flanks[flank_idx[j].second].c_str();
    }

    c->sr = sr; c->si = si; c->barrier = barrier; c->soft = soft;
    return c;
  };
  // ---- the hot path: the batch's look-ups and fill kernel queued on the GPU -----------------------------------
  auto begin = [&](Chunk* c) -> int {
    const size_t q = nchunks++ & 1u;
    c->results = (g2s_result*)res_buf[q].get(std::max<size_t>(1, c->jobs.size()) * sizeof(g2s_result));
    c->arena_bytes = c->jobs.empty() ? 0 : g2s_team_arena_bytes(s, c->jobs.data(), c->jobs.size());
    c->arena = (char*)arena_buf[q].get(std::max<size_t>(1, c->arena_bytes));
    if (c->jobs.empty()) return G2S_OK;
    // (long lists go through the session's team in g2s_fill_end: g2s_session_set_team)
    const int rc = g2s_fill_begin(s, c->jobs.data(), c->jobs.size(), c->results, c->arena, c->arena_bytes);
    c->begun = rc == G2S_OK;
    return rc;
  };
  auto finish = [&](Chunk* c) -> int {
    if (c->begun) {
      const int rc = g2s_fill_end(s);
      if (rc != G2S_OK) return rc;
    }
    const std::vector<GapEvent>& events = c->events;
    const std::vector<g2s_gap>& jobs = c->jobs;
    const g2s_result* results = c->results;
    const char* arena = c->arena;
    const size_t sr = c->sr, si = c->si;
    const bool barrier = c->barrier, soft = c->soft;
    // ---- exact replay of the reference's splice logic (:340-423) ---------------
    auto finish_record = [&]() {
      const std::string& seq = recs[cur_rec].seq;
      filledSeq += seq.substr((size_t)prevGapEnd);  // :423
      append_fasta(&fasta, recs[cur_rec].comment, filledSeq);  // :426-431
      filledSeq.clear();
      prevGapEnd = 0;
      i_exact = 0;
      cur_rec++;
    };
    for (const GapEvent& ev : events) {
      while (cur_rec < ev.rec) finish_record();
      const std::string& seq = recs[cur_rec].seq;
      const std::string& comment = recs[cur_rec].comment;
      gapcount++;
      size_t i = ev.iend;
      const int lmf = ev.lmf, rmf = ev.rmf, kmer_start = ev.kmer_start, gap = ev.gap;
      const bool eligible = kmer_start >= prevGapEnd && ev.ok;
      if (eligible) {
        const g2s_result& r = results[(size_t)ev.job];
        const int sres = r.count;
        const char* tail = arena + r.fill_off;
        if (r.flags & G2S_GAP_BACKTRACE_FAIL) {
          char msg[192];
          if (g2s_backtrace_text(&jobs[(size_t)ev.job], &r, k, msg, sizeof msg)) os << msg << "\n";
        }
        const int filledStart = (int)filledSeq.length() + kmer_start + k + lmf - r.left_fuz - prevGapEnd;
        const int gapStart = kmer_start + k + lmf;
        stats_line(os, filledStart, gapStart, (int)i, sres, tail, k, lmf, rmf, r.left_fuz, r.right_fuz,
                   p.skip_confident != 0, p.unique_paths != 0, r, gap, comment);
        if (sres > 0 && (!p.unique_paths || sres == 1)) {
          filledgapcount++;
          // :396 assignment, not append (Q8); :399 drop the right k-mer
          filledSeq = seq.substr((size_t)prevGapEnd, (size_t)(kmer_start + k + lmf - r.left_fuz - prevGapEnd)) +
                      std::string(tail);
          filledSeq.resize(filledSeq.size() - (size_t)k);
          i += (size_t)r.right_fuz;  // :402
        } else {
          filledSeq += seq.substr((size_t)prevGapEnd, (size_t)(kmer_start + k + lmf + gap - prevGapEnd));  // :406
        }
      } else {
        filledSeq += seq.substr((size_t)prevGapEnd, (size_t)(kmer_start + k + lmf + gap - prevGapEnd));  // :412
      }
      prevGapEnd = (int)i;  // :415
      i_exact = i;
    }
    if (barrier) {
      while (cur_rec < sr) finish_record();
      // resume exactly at the barrier gap of record sr: N-run boundaries do not move,
      // only prevGapEnd (already exact) matters
      i_exact = si;
    } else if (soft) {
      while (cur_rec < sr) finish_record();  // (record sr starts the next batch)
    } else {
      while (cur_rec < recs.size()) finish_record();
      done = true;
    }
    if (done) os << "Filled " << filledgapcount << " gaps out of " << gapcount << "\n";  // :437
    {  // what is decided so far leaves now
      const std::string lg = os.str();
      if (on_log && !lg.empty()) on_log(lg.data(), lg.size(), user);
      os.str(std::string());
      if (on_fasta && !fasta.empty()) on_fasta(fasta.data(), fasta.size(), user);
      fasta.clear();
    }
    return G2S_OK;
  };
  std::unique_ptr<Chunk> cur(done ? nullptr : scan(cur_rec, i_exact, prevGapEnd));
  if (cur) { const int rc = begin(cur.get()); if (rc != G2S_OK) return rc; }
  while (cur) {
    std::unique_ptr<Chunk> nxt;
    if (cur->soft) {  // (record cur->sr starts the next batch, whatever this one's results are)
      nxt.reset(scan(cur->sr, 0, 0));
      const int rc = begin(nxt.get());
      if (rc != G2S_OK) { if (cur->begun) (void)g2s_fill_end(s); return rc; }
    }
    {
      const int rc = finish(cur.get());
      if (rc != G2S_OK) { if (nxt && nxt->begun) (void)g2s_fill_end(s); return rc; }
    }
    if (!nxt && !done) {  // behind a barrier: the scan resumes where the replay stands
      nxt.reset(scan(cur_rec, i_exact, prevGapEnd));
      const int rc = begin(nxt.get());
      if (rc != G2S_OK) return rc;
    }
    cur = std::move(nxt);
  }
  if (recs.empty()) {
    os << "Filled " << filledgapcount << " gaps out of " << gapcount << "\n";
    const std::string lg = os.str();
    if (on_log) on_log(lg.data(), lg.size(), user);
  }
  if (gaps_out) *gaps_out = gapcount;
  if (filled_out) *filled_out = filledgapcount;
  return G2S_OK;
}

namespace {
struct TwoTexts { std::string fasta, log; };
void collect_fasta(const char* t, size_t n, void* u) { ((TwoTexts*)u)->fasta.append(t, n); }
void collect_log(const char* t, size_t n, void* u) { ((TwoTexts*)u)->log.append(t, n); }
}  // namespace

extern "C" int g2s_execute_scaffolds(g2s_session* s, const g2s_run_opts* o, const char* reads_label,
                                     const char* filled_label, const char* scaffolds_text, char** fasta_out,
                                     char** log_out, int32_t* gaps_out, int32_t* filled_out) {
  TwoTexts tt;
  const int rc = g2s_execute_scaffolds_stream(s, o, reads_label, filled_label, scaffolds_text, 0, collect_fasta, collect_log,
                                              &tt, gaps_out, filled_out);
  if (rc != G2S_OK) return rc;
  if (fasta_out) *fasta_out = dup_text(tt.fasta);
  if (log_out) *log_out = dup_text(tt.log);
  return G2S_OK;
}

// Single-gap mode (Gap2Seq.cpp:227-283)
extern "C" int g2s_execute_single(g2s_session* s, const g2s_run_opts* o, const char* reads_label,
                                  const char* filled_label, const char* left, const char* right, int32_t length,
                                  char** fasta_out, char** log_out) {
  if (!s || !o || !left || !right) return G2S_ERR_ARG;
  g2s_params p;
  g2s_session_get_params(s, &p);
  const int k = g2s_graph_k(g2s_session_graph(s));
  std::ostringstream os;
  std::string fasta;
  echo_params(os, *o, p, reads_label, filled_label, (long long)(o->max_mem_gb * 1024 * 1024 * 1024));
  const std::string left_flank(left), right_flank(right);
  if ((int)left_flank.length() < k || (int)right_flank.length() < k) {
    fprintf(stderr, "Flanks need to be at least k length\n");  // :233-236
    if (fasta_out) *fasta_out = dup_text("");
    if (log_out) *log_out = dup_text(os.str());
    return G2S_OK;
  }
  const int lmf = std::min((int)left_flank.length() - k, o->max_fuz);
  const int rmf = std::min((int)right_flank.length() - k, o->max_fuz);
  g2s_gap gj;
  memset(&gj, 0, sizeof gj);
  gj.left = left_flank.c_str();
  gj.right = right_flank.c_str();
  gj.left_len = (int)left_flank.size();
  gj.right_len = (int)right_flank.size();
  gj.gap_len = length;
  gj.lmf = lmf;
  gj.rmf = rmf;
  gj.skip_if_prev_right_fuz_gt = -1;
  g2s_result r;
  std::vector<char> arena((size_t)(length + k + p.d_err + lmf + rmf + 3));
  int rc = g2s_fill_batch(s, &gj, 1, &r, arena.data(), arena.size());
  if (rc != G2S_OK) return rc;
  if (r.flags & G2S_GAP_BACKTRACE_FAIL) {
    char msg[192];
    if (g2s_backtrace_text(&gj, &r, k, msg, sizeof msg)) os << msg << "\n";
  }
  const char* tail = arena.data() + r.fill_off;
  const int filledStart = (int)left_flank.length() - lmf - r.left_fuz;  // :257
  stats_line(os, filledStart, (int)left_flank.length(), (int)left_flank.length() + length, r.count, tail, k, lmf, rmf,
             r.left_fuz, r.right_fuz, p.skip_confident != 0, p.unique_paths != 0, r, length, "");
  std::string filledSeq;
  if (r.count > 0 && (!p.unique_paths || r.count == 1)) {
    filledSeq = left_flank.substr(0, left_flank.length() - (size_t)r.left_fuz) + std::string(tail);  // :268
  } else {
    filledSeq = left_flank + std::string((size_t)length, 'N') + right_flank;  // :270-272
  }
  append_fasta(&fasta, "", filledSeq);
  if (fasta_out) *fasta_out = dup_text(fasta);
  if (log_out) *log_out = dup_text(os.str());
  return G2S_OK;
}
