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
// Pure host string work: nothing here touches the GPU.  g2s_cut_scaffolds follows the reference's scan
// (GapCutter.cpp:160-319) case for case: its chain of distances from the current position (bases, N run,
// bases, N run, bases) is the same chain here under other names, because the three cases and every offset
// in them ARE the file format the wrapper and GapMerger depend on.  g2s_merge_scaffolds is arranged
// differently (an index over the gap records instead of the reference's rescan per contig).
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

// length of the stretch of bases (gap = false) or of N/n (gap = true) that starts at `from`
size_t stretch(const std::string& s, size_t from, bool gap) {
  size_t e = from;
  while (e < s.size() && is_gap_char(s[e]) == gap) e++;
  return e - from;
}

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
    size_t at = 0;
    while (at < s.size()) {
      const std::string piece = rec.comment + kScaffold + std::to_string(n_scaffold) + kContig + std::to_string(n_contig);
      const std::string gap_piece = piece + kGap + std::to_string(n_gap);
      const size_t left = stretch(s, at, false);          // bases before the next gap
      const size_t hole = stretch(s, at + left, true);    // that gap
      if (left > 0 && hole == 0) {                        // no gap left: the rest is a contig (:189-195)
        append_fasta(&contigs, piece, s.substr(at));
        n_contig++;
        break;
      }
      if (left < k) {                                     // no room for a left flank: contig up to the gap's end (:198-204)
        append_fasta(&contigs, piece, s.substr(at, left + hole));
        n_contig++;
        at += left + hole;
        continue;
      }
      const size_t lflank = std::min(left, reach);
      const size_t gap_at = at + left;                    // first N of the gap
      const size_t mid = stretch(s, gap_at + hole, false);               // bases between this gap and the next
      const size_t hole2 = stretch(s, gap_at + hole + mid, true);        // the next gap
      const size_t after = stretch(s, gap_at + hole + mid + hole2, false);  // bases after the next gap
      // ---- case 1: enough sequence on both sides (:212-233)
      if (mid >= 2 * k || (mid >= k && after == 0)) {
        const size_t rflank = mid >= 2 * k ? std::min(mid, reach) : mid;  // the last gap takes the rest of the scaffold
        append_fasta(&gaps, gap_piece, s.substr(gap_at - lflank, lflank + hole + rflank));
        append_fasta(&contigs, gap_piece, s.substr(at, left - lflank));
        bed << name << "\t" << gap_at - lflank << "\t" << gap_at + hole + rflank << "\n";
        n_gap++;
        n_contig++;
        at = gap_at + hole + rflank;
        continue;
      }
      // ---- case 2: two gaps share the sequence between them as flank (:236-277)
      if (mid >= k) {
        if (!no_split && after >= k) {
          const size_t flank3 = std::min(after, reach);
          // the first gap gets the whole middle as right flank, the second only its last k bases as left flank
          append_fasta(&gaps, gap_piece + kSplit + "1", s.substr(gap_at - lflank, lflank + hole + mid));
          append_fasta(&gaps, gap_piece + kSplit + "2 " + std::to_string(k), s.substr(gap_at + hole + mid - k, k + hole2 + flank3));
          append_fasta(&contigs, gap_piece, s.substr(at, left - lflank));
          bed << name << "\t" << gap_at - lflank << "\t" << gap_at + hole + mid << "\n";
          bed << name << "\t" << gap_at + hole << "\t" << gap_at + hole + mid + hole2 + flank3 << "\n";
          n_gap++;
          n_contig++;
          at = gap_at + hole + mid + hole2 + flank3;
        } else {
          const size_t rflank = std::min(mid, reach);
          append_fasta(&gaps, gap_piece, s.substr(gap_at - lflank, lflank + hole + rflank));
          append_fasta(&contigs, gap_piece, s.substr(at, left - lflank));
          bed << name << "\t" << gap_at - lflank << "\t" << gap_at + hole + rflank << "\n";
          n_gap++;
          n_contig++;
          at = gap_at + hole + rflank;
        }
        continue;
      }
      // ---- case 3: gaps around sequence too short to be a flank: look for the next usable flank (:279-319)
      size_t span = hole + mid + hole2, next = after;
      while (next > 0 && next < k) {
        span += next + stretch(s, gap_at + span + next, true);
        next = stretch(s, gap_at + span, false);
      }
      if (next < k) {  // the last stretch is too short as well: everything left is a contig
        append_fasta(&contigs, piece, s.substr(at));
        n_contig++;
        break;
      }
      if (mask) {      // everything between the two flanks becomes one gap of n's
        const size_t rflank = std::min(next, reach);
        append_fasta(&gaps, gap_piece, s.substr(gap_at - lflank, lflank) + std::string(span, 'n') + s.substr(gap_at + span, rflank));
        append_fasta(&contigs, gap_piece, s.substr(at, left - lflank));
        bed << name << "\t" << gap_at - lflank << "\t" << gap_at + span + rflank << "\n";
        n_gap++;
        n_contig++;
        at = gap_at + span + rflank;
      } else {
        append_fasta(&contigs, piece, s.substr(at, left + span));
        n_contig++;
        at = gap_at + span;
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
