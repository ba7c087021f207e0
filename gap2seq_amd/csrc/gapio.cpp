// gap2seq_amd/csrc/gapio.cpp — the file formats either side of Gap2Seq-core in the reference's
// pipeline, so that the unmodified wrapper flow cut -> fill -> merge
// (/root/reference/src/Gap2Seq.py:294-326) can run around this core:
//   g2s_cut_scaffolds   = what GapCutter writes (/root/reference/src/GapCutter.cpp:119-321):
//                         one record per gap (flanks of at most k+fuz bases around the N run,
//                         comment "<name> scaffold S contig C gap G[ split 1| split 2 k]"), the
//                         contigs in between, and a BED line per gap record;
//   g2s_merge_scaffolds = what GapMerger reads back (/root/reference/src/GapMerger.cpp:142-235):
//                         contigs in order, each followed by its (filled) gap record(s), the
//                         markers stripped from the comment.
// Pure host string work: nothing here touches the GPU.  g2s_cut_scaffolds tokenises a record ONCE into
// runs (bases / N's) and walks the runs with a cursor (ScaffoldRuns below): what a cut emits is decided by
// the lengths of the runs ahead of the cursor, classified into one of five outcomes (Cut) and emitted by
// outcome.  The outcomes and every offset in them are the file format the wrapper and GapMerger depend on;
// tests/test_gapio.py holds them against an independent Python restatement.
// g2s_merge_scaffolds keeps an index over the gap records instead of the reference's rescan per contig.
#include <cstdlib>
#include <cstring>
#include <map>
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

inline bool is_gap_char(char c) { return c == 'N' || c == 'n'; }

// A scaffold record as alternating runs of bases and of N/n, with a cursor.  The cursor may stand inside a
// run of bases (behind a right flank that became part of a gap record); it never stands inside a run of N's.
struct ScaffoldRuns {
  struct Run { size_t start, len; bool gap; };
  std::vector<Run> runs;
  size_t cur = 0;      // run the cursor stands in
  size_t at = 0;       // the cursor's position in the record
  size_t size = 0;

  explicit ScaffoldRuns(const std::string& s) : size(s.size()) {
    for (size_t i = 0; i < s.size();) {
      size_t e = i;
      const bool gap = is_gap_char(s[i]);
      while (e < s.size() && is_gap_char(s[e]) == gap) e++;
      runs.push_back({i, e - i, gap});
      i = e;
    }
  }
  bool done() const { return at >= size; }
  // index of the first run of N's at or behind the cursor
  size_t gap_run() const { return cur < runs.size() && runs[cur].gap ? cur : cur + 1; }
  // bases between the cursor and that run
  size_t bases_ahead() const { return cur < runs.size() && !runs[cur].gap ? runs[cur].start + runs[cur].len - at : 0; }
  size_t len(size_t r) const { return r < runs.size() ? runs[r].len : 0; }
  size_t start(size_t r) const { return r < runs.size() ? runs[r].start : size; }
  void move_to(size_t pos) {
    at = pos;
    while (cur < runs.size() && runs[cur].start + runs[cur].len <= at) cur++;
  }
};

// what the runs ahead of the cursor amount to
enum class Cut {
  kTail,        // no N's ahead: the rest of the record is a contig (GapCutter.cpp:189-195)
  kNoLeftFlank, // fewer than k bases in front of the N's: contig up to their end (:198-204)
  kPlain,       // a gap with a flank of its own on either side (:212-233, and :236-277 when not split)
  kSplit,       // two gaps that share the bases between them as flank (:236-277)
  kCluster,     // N runs around stretches too short to be flanks (:279-319)
};

const char* kScaffold = " scaffold ";
const char* kContig = " contig ";
const char* kGap = " gap ";
const char* kSplit = " split ";

// the number that follows `marker` in a comment, up to `until` (or the end); -1 without the marker
int marker_index(const std::string& comment, const char* marker, const char* until) {
  const size_t at = comment.find(marker);
  if (at == std::string::npos) return -1;
  const size_t from = at + strlen(marker);
  const size_t to = until ? comment.find(until) : std::string::npos;
  const std::string digits = to == std::string::npos ? comment.substr(from) : comment.substr(from, to > from ? to - from : 0);
  return atoi(digits.c_str());
}

}  // namespace

extern "C" int g2s_cut_scaffolds(const char* scaffolds_text, int k_, int fuz_, int mask, int no_split,
                                 const char* scaffolds_label, const char* contigs_label, const char* gaps_label,
                                 const char* bed_label, char** contigs_out, char** gaps_out, char** bed_out,
                                 char** log_out) {
  if (!scaffolds_text || k_ < 1 || fuz_ < 0) return G2S_ERR_ARG;
  const size_t k = (size_t)k_, reach = (size_t)k_ + (size_t)fuz_;  // a flank is at most k + fuz bases
  std::ostringstream log, bed;
  std::string contigs, gaps;
  // GapCutter.cpp:141-148
  log << "Scaffolds file: " << (scaffolds_label ? scaffolds_label : "") << "\n";
  log << "Contigs file: " << (contigs_label ? contigs_label : "") << "\n";
  log << "Gaps file: " << (gaps_label ? gaps_label : "") << "\n";
  log << "BED file: " << (bed_label ? bed_label : "") << "\n";
  log << "k-mer size: " << k << "\n";
  log << "Fuz: " << fuz_ << "\n";
  log << "Mask: " << (mask ? 1 : 0) << "\n";
  log << "Split: " << (no_split ? 0 : 1) << "\n";

  std::vector<FastxRecord> recs;
  parse_fastx(std::string(scaffolds_text), &recs);
  int n_contig = 0, n_gap = 0, n_scaffold = 0;
  for (const FastxRecord& rec : recs) {
    const std::string& s = rec.seq;
    // BED lines carry the record's name = the comment up to its first blank (:173-175; the reference
    // asserts that there is one, here a comment without a blank is its own name)
    const std::string name = rec.comment.substr(0, rec.comment.find(' '));
    const std::string tag = rec.comment + kScaffold + std::to_string(n_scaffold) + kContig;
    ScaffoldRuns sc(s);
    auto contig = [&](const std::string& comment, size_t from, size_t to) {
      append_fasta(&contigs, comment, s.substr(from, to - from));
      n_contig++;
    };
    auto bed_line = [&](size_t from, size_t to) { bed << name << "\t" << from << "\t" << to << "\n"; };
    while (!sc.done()) {
      const std::string piece = tag + std::to_string(n_contig);
      const std::string gap_piece = piece + kGap + std::to_string(n_gap);
      const size_t g = sc.gap_run();              // the N's ahead, then: g+1 bases, g+2 N's, g+3 bases ...
      const size_t left = sc.bases_ahead();
      const size_t mid = sc.len(g + 1), after = sc.len(g + 3);
      // ---- classify
      Cut what;
      size_t far = g + 3;                          // kCluster: the first later run of bases that can be a flank
      if (g >= sc.runs.size()) what = Cut::kTail;
      else if (left < k) what = Cut::kNoLeftFlank;
      else if (mid >= 2 * k || (mid >= k && after == 0)) what = Cut::kPlain;
      else if (mid >= k) what = (!no_split && after >= k) ? Cut::kSplit : Cut::kPlain;
      else {
        what = Cut::kCluster;
        while (far < sc.runs.size() && sc.len(far) < k) far += 2;
        if (far >= sc.runs.size()) what = Cut::kTail;   // no flank anywhere behind: everything left is a contig
      }
      // ---- emit
      const size_t gap_at = sc.start(g), gap_end = gap_at + sc.len(g);
      const size_t lflank = std::min(left, reach);
      switch (what) {
        case Cut::kTail:
          contig(piece, sc.at, sc.size);
          sc.move_to(sc.size);
          break;
        case Cut::kNoLeftFlank:
          contig(piece, sc.at, gap_end);
          sc.move_to(gap_end);
          break;
        case Cut::kPlain: {
          // the scaffold's last gap takes whatever follows it; otherwise at most k + fuz bases
          const size_t rflank = (mid < 2 * k && after == 0) ? mid : std::min(mid, reach);
          append_fasta(&gaps, gap_piece, s.substr(gap_at - lflank, lflank + (gap_end - gap_at) + rflank));
          bed_line(gap_at - lflank, gap_end + rflank);
          contig(gap_piece, sc.at, gap_at - lflank);
          n_gap++;
          sc.move_to(gap_end + rflank);
          break;
        }
        case Cut::kSplit: {
          // the first half gets the whole middle as right flank, the second its last k bases as left flank
          const size_t gap2_at = sc.start(g + 2), gap2_end = gap2_at + sc.len(g + 2);
          const size_t flank3 = std::min(after, reach);
          append_fasta(&gaps, gap_piece + kSplit + "1", s.substr(gap_at - lflank, gap2_at - (gap_at - lflank)));
          append_fasta(&gaps, gap_piece + kSplit + "2 " + std::to_string(k), s.substr(gap2_at - k, gap2_end + flank3 - (gap2_at - k)));
          bed_line(gap_at - lflank, gap2_at);
          bed_line(gap_end, gap2_end + flank3);
          contig(gap_piece, sc.at, gap_at - lflank);
          n_gap++;
          sc.move_to(gap2_end + flank3);
          break;
        }
        case Cut::kCluster: {
          const size_t far_at = sc.start(far);      // everything in [gap_at, far_at) lies between two flanks
          if (mask) {                               // and becomes one gap of n's
            const size_t rflank = std::min(sc.len(far), reach);
            append_fasta(&gaps, gap_piece,
                         s.substr(gap_at - lflank, lflank) + std::string(far_at - gap_at, 'n') + s.substr(far_at, rflank));
            bed_line(gap_at - lflank, far_at + rflank);
            contig(gap_piece, sc.at, gap_at - lflank);
            n_gap++;
            sc.move_to(far_at + rflank);
          } else {                                  // or stays inside a contig
            contig(piece, sc.at, far_at);
            sc.move_to(far_at);
          }
          break;
        }
      }
    }
    n_scaffold++;
  }
  log << "Cut " << n_scaffold << " scaffolds into " << n_contig << " contigs and " << n_gap << " gaps\n";  // :323
  if (contigs_out) *contigs_out = dup_text(contigs);
  if (gaps_out) *gaps_out = dup_text(gaps);
  if (bed_out) *bed_out = dup_text(bed.str());
  if (log_out) *log_out = dup_text(log.str());
  return G2S_OK;
}

extern "C" int g2s_merge_scaffolds(const char* contigs_text, const char* gaps_text, const char* scaffolds_label,
                                   const char* contigs_label, const char* gaps_label, char** scaffolds_out, char** log_out) {
  if (!contigs_text || !gaps_text) return G2S_ERR_ARG;
  std::ostringstream log;
  log << "Scaffolds file: " << (scaffolds_label ? scaffolds_label : "") << "\n";  // GapMerger.cpp:148-150
  log << "Contigs file: " << (contigs_label ? contigs_label : "") << "\n";
  log << "Gaps file: " << (gaps_label ? gaps_label : "") << "\n";
  std::vector<FastxRecord> contigs, gaps;
  parse_fastx(std::string(contigs_text), &contigs);
  parse_fastx(std::string(gaps_text), &gaps);
  // gap records by gap index, in file order (the reference scans the whole gap file per contig, :192-219)
  std::map<int, std::vector<size_t>> by_gap;
  for (size_t g = 0; g < gaps.size(); g++) {
    const int gi = marker_index(gaps[g].comment, kGap, kSplit);
    if (gi >= 0) by_gap[gi].push_back(g);
  }
  std::string out, scaffold, scaffold_comment = contigs.empty() ? std::string() : contigs[0].comment;
  auto emit = [&]() {  // markers are stripped: the comment up to " scaffold " (:124-138)
    const size_t m = scaffold_comment.find(kScaffold);
    append_fasta(&out, m == std::string::npos ? scaffold_comment : scaffold_comment.substr(0, m), scaffold);
  };
  int n_contigs = 0, n_gaps = 0, current = 0;
  for (const FastxRecord& c : contigs) {
    const int sc = marker_index(c.comment, kScaffold, kContig);
    const int gi = marker_index(c.comment, kGap, kSplit);
    n_contigs++;
    if (sc != current) {  // the previous scaffold is complete (:170-177)
      emit();
      scaffold.clear();
      scaffold_comment = c.comment;
      current = sc;
    }
    scaffold += c.seq;
    if (gi != -1) {  // the gap behind this contig: one record, or the two halves of a split (:181-222)
      std::string first, second;
      auto it = by_gap.find(gi);
      if (it != by_gap.end()) {
        for (size_t g : it->second) {
          const std::string& gc = gaps[g].comment;
          const size_t sp = gc.find(kSplit);
          if (sp == std::string::npos) { first = gaps[g].seq; break; }
          const size_t num_at = sp + strlen(kSplit);
          const int half = num_at < gc.size() ? atoi(gc.substr(num_at, 1).c_str()) : 0;
          if (half == 1) first = gaps[g].seq;
          else {  // the second half starts with the k bases it shares with the first: dropped
            const size_t cut = num_at + 2 <= gc.size() ? (size_t)std::max(0, atoi(gc.substr(num_at + 2).c_str())) : 0;
            second = cut <= gaps[g].seq.size() ? gaps[g].seq.substr(cut) : std::string();
          }
          if (!first.empty() && !second.empty()) break;
        }
      }
      scaffold += first + second;
      n_gaps++;
    }
  }
  int n_scaffolds = current;
  if (!scaffold.empty()) { emit(); n_scaffolds++; }  // :226-229
  log << "Merged " << n_contigs << " contigs and " << n_gaps << " gaps into " << n_scaffolds << " scaffolds\n";  // :233
  if (scaffolds_out) *scaffolds_out = dup_text(out);
  if (log_out) *log_out = dup_text(log.str());
  return G2S_OK;
}
