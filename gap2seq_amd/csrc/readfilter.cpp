// gap2seq_amd/csrc/readfilter.cpp — the reference's `ReadFilter` (/root/reference/src/ReadFilter.cpp:344-415):
// from the aligned reads of one library, the reads that can belong to ONE gap — the unmapped mates of reads
// aligned an insert size away from the breakpoint, the reads overlapping the flanks — or every unmapped read.
// Host string work on the wrapper's side of Gap2Seq-core (Gap2Seq.py:64-72,145-149); BAM input through bam.hpp
// (zlib) instead of htslib, no GATB.  Output order, names ("/1", "/2"), reverse-complementing and the name filter
// follow the reference line by line in MEANING; the passes over the file are arranged differently (two instead
// of five, see below).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/g2s.h"
#include "bam.hpp"

namespace {

thread_local std::string rf_error;

// The name filter.  The reference uses GATB's BloomSynchronized<std::string>(5 * number of records) with the
// default four hash functions, and supplies hash1(key, seed) = std::hash<std::string>(key), which ignores the
// seed (ReadFilter.cpp:28-33,371): the four probes hit the same bit, so the filter is ONE bit per name at
// std::hash(name) % (5 * records), and a name that was never inserted is "contained" whenever its bit collides.
// (GATB's Bloom layout — a bit array of `tai` bits indexed by hash % tai — is recalled, not checked; the
// reference library is not in the image.)  Reproduced as such so that small inputs show the same collisions.
class NameFilter {
 public:
  explicit NameFilter(uint64_t bits) : bits_(bits), words_((size_t)(bits / 64 + 1), 0) {}
  void insert(const std::string& s) {
    if (!bits_) return;
    const uint64_t h = (uint64_t)std::hash<std::string>{}(s) % bits_;
    words_[(size_t)(h >> 6)] |= (uint64_t)1 << (h & 63);
  }
  bool contains(const std::string& s) const {
    if (!bits_) return false;
    const uint64_t h = (uint64_t)std::hash<std::string>{}(s) % bits_;
    return (words_[(size_t)(h >> 6)] >> (h & 63)) & 1;
  }
 private:
  uint64_t bits_;
  std::vector<uint64_t> words_;
};

// ReadFilter.cpp:165-173: the read's name with its end, and its mate's
inline std::string own_name(const g2s::BamRec& r) {
  return std::string(r.name, strnlen(r.name, r.l_name)) + ((r.flag & g2s::BAM_READ1) ? "/1" : "/2");
}
inline std::string mate_name(const g2s::BamRec& r) {
  return std::string(r.name, strnlen(r.name, r.l_name)) + ((r.flag & g2s::BAM_READ1) ? "/2" : "/1");
}

// ReadFilter.cpp:105-161: the read as sequenced (reverse strand alignments are complemented back); every code
// other than A, C, G, T becomes N
void append_fasta(const g2s::BamRec& r, std::string* out) {
  static const char fwd[16] = {'N', 'A', 'C', 'N', 'G', 'N', 'N', 'N', 'T', 'N', 'N', 'N', 'N', 'N', 'N', 'N'};
  static const char rev[16] = {'N', 'T', 'G', 'N', 'C', 'N', 'N', 'N', 'A', 'N', 'N', 'N', 'N', 'N', 'N', 'N'};
  out->push_back('>');
  out->append(own_name(r));
  out->push_back('\n');
  const size_t at = out->size();
  out->resize(at + (size_t)r.l_seq);
  char* d = &(*out)[at];
  if (!(r.flag & g2s::BAM_REVERSE))
    for (int32_t i = 0; i < r.l_seq; i++) d[i] = fwd[r.base4(i)];
  else
    for (int32_t i = 0; i < r.l_seq; i++) d[i] = rev[r.base4(r.l_seq - 1 - i)];
  out->push_back('\n');
}

// htslib's region iterator as the reference calls it (sam_itr_queryi, ReadFilter.cpp:184-191): a negative
// start is 0; an end in front of the start gives NO iterator (the reference then prints a warning and reads
// nothing, :188-190,213-217) — which is what happens to its right-hand window, whose bounds are written
// the wrong way round (:388-389), whenever the standard deviation is not 0.
struct Region {
  int tid;
  int64_t beg, end;
  bool valid;
};
Region make_region(int tid, int64_t beg, int64_t end, std::string* warn) {
  Region q{tid, beg < 0 ? 0 : beg, end, true};
  if (tid < 0 || q.end < q.beg) {
    q.valid = false;
    warn->append("WARNING: SAM iterator is NULL!\n");
  }
  return q;
}
// (an empty region [x, x) yields nothing and no warning: htslib's reg2bins returns no bin for beg >= end, so the
// iterator exists and ends at once — it is not a point query)
inline bool overlaps(const g2s::BamRec& r, const Region& q) {
  return q.valid && q.beg < q.end && r.ref_id == q.tid && (int64_t)r.pos < q.end && r.end_pos() > q.beg;
}

char* dup_text(const std::string& s) {
  char* p = (char*)malloc(s.size() + 1);
  if (!p) return nullptr;
  memcpy(p, s.data(), s.size());
  p[s.size()] = 0;
  return p;
}

int run_filter(g2s::BamFile& bam, const g2s_filter_opts* o, char** fasta_out, char** log_out, char** warn_out,
               int64_t* extracted_out, int64_t* total_out) {
  std::string err, warn, fasta, region_fasta;
  bam.set_threads(o->threads > 0 ? o->threads : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency())));
  // pass 1 (:225-241): the number of records and the longest read.  The windows of pass 2 depend on that length, so
  // they cannot be applied yet — but only reads of the gap's scaffold whose mate is unmapped can ever be in them:
  // those are remembered (position, end, name), which saves the reference's pass over the file for them unless
  // there are millions.
  uint64_t total = 0;
  int32_t read_length = 0;
  const int tid = o->unmapped_only ? -1 : bam.ref_id(o->scaffold ? o->scaffold : "");  // :381
  struct Cand { int64_t pos, end; std::string name; };
  std::vector<Cand> cands;
  // (G2S_FILTER_MAX_CANDS: the tests make it small to take the pass over the file instead)
  const size_t kMaxCands = getenv("G2S_FILTER_MAX_CANDS") ? (size_t)atoll(getenv("G2S_FILTER_MAX_CANDS")) : (size_t)2 << 20;
  bool cands_complete = true;
  if (!bam.for_each([&](const g2s::BamRec& r) {
        total++;
        read_length = std::max(read_length, r.l_seq);
        if (tid >= 0 && r.ref_id == tid && (r.flag & g2s::BAM_MATE_UNMAPPED)) {
          if (cands.size() < kMaxCands) cands.push_back({(int64_t)r.pos, r.end_pos(), own_name(r)});
          else cands_complete = false;
        }
        return true;
      }, &err)) {
    rf_error = err;
    return G2S_ERR_IO;
  }
  NameFilter names(5 * total);  // :371
  int64_t extracted = 0;
  if (!o->unmapped_only) {
    // :383-390: where a read must align for its mate to fall in the gap (the right-hand window as written there)
    const int64_t bp = o->breakpoint, mu = o->mean_insert, sd = o->std_dev, gl = o->gap_length, rl = read_length;
    const Region left = make_region(tid, bp - (mu + 3 * sd + 2 * rl), bp - (mu - 3 * sd + rl), &warn);
    const Region right = make_region(tid, bp + (mu + 3 * sd + rl) + gl, bp + (mu - 3 * sd + rl) + gl, &warn);
    // pass 2 (:300-310): names of reads in the windows whose mate is unmapped
    auto in_window = [](int64_t pos, int64_t end, const Region& q) { return q.valid && pos < q.end && end > q.beg; };
    if ((left.valid || right.valid) && cands_complete) {
      for (const Cand& c : cands)
        if (in_window(c.pos, c.end, left) || in_window(c.pos, c.end, right)) names.insert(c.name);
    } else if (left.valid || right.valid) {
      if (!bam.for_each([&](const g2s::BamRec& r) {
            if (!(r.flag & g2s::BAM_MATE_UNMAPPED)) return true;
            if (overlaps(r, left) || overlaps(r, right)) names.insert(own_name(r));
            return true;
          }, &err)) {
        rf_error = err;
        return G2S_ERR_IO;
      }
    }
    std::vector<Cand>().swap(cands);
    // pass 3: (:313-323) every record whose MATE's name is in the filter, then (:395-399, :283-297) the reads
    // overlapping the flanks whose own name is not — the reference makes two passes and writes as it goes;
    // here the second list is collected beside the first and appended
    const bool flanks = o->flank_length != -1;
    Region around{0, 0, 0, false};
    if (flanks) around = make_region(tid, bp - (int64_t)o->flank_length, bp + (int64_t)o->flank_length + gl, &warn);
    if (!bam.for_each([&](const g2s::BamRec& r) {
          if (names.contains(mate_name(r))) { append_fasta(r, &fasta); extracted++; }
          if (overlaps(r, around) && !names.contains(own_name(r))) { append_fasta(r, &region_fasta); extracted++; }
          return true;
        }, &err)) {
      rf_error = err;
      return G2S_ERR_IO;
    }
    fasta += region_fasta;
  } else {
    // :326-337: every unmapped read (the filter is empty here)
    if (!bam.for_each([&](const g2s::BamRec& r) {
          if ((r.flag & g2s::BAM_UNMAPPED) && !names.contains(own_name(r))) { append_fasta(r, &fasta); extracted++; }
          return true;
        }, &err)) {
      rf_error = err;
      return G2S_ERR_IO;
    }
  }
  const std::string log = "Extracted " + std::to_string(extracted) + " out of " + std::to_string(total) + " reads\n";  // :406
  if (fasta_out) *fasta_out = dup_text(fasta);
  if (log_out) *log_out = dup_text(log);
  if (warn_out) *warn_out = dup_text(warn);
  if ((fasta_out && !*fasta_out) || (log_out && !*log_out) || (warn_out && !*warn_out)) {  // (out of memory)
    if (fasta_out) { free(*fasta_out); *fasta_out = nullptr; }
    if (log_out) { free(*log_out); *log_out = nullptr; }
    if (warn_out) { free(*warn_out); *warn_out = nullptr; }
    rf_error = "out of memory";
    return G2S_ERR_NOMEM;
  }
  if (extracted_out) *extracted_out = extracted;
  if (total_out) *total_out = (int64_t)total;
  return G2S_OK;
}

}  // namespace

extern "C" {

const char* g2s_filter_last_error(void) { return rf_error.c_str(); }

int g2s_filter_reads(const char* bam_path, const g2s_filter_opts* o, char** fasta_out, char** log_out, char** warn_out,
                     int64_t* extracted, int64_t* total) {
  if (!bam_path || !o) return G2S_ERR_ARG;
  g2s::BamFile bam;
  std::string err;
  if (!bam.open_path(bam_path, &err)) { rf_error = err; return G2S_ERR_IO; }
  return run_filter(bam, o, fasta_out, log_out, warn_out, extracted, total);
}

int g2s_filter_reads_mem(const void* bam_bytes, size_t n, const g2s_filter_opts* o, char** fasta_out, char** log_out,
                         char** warn_out, int64_t* extracted, int64_t* total) {
  if (!bam_bytes || !o) return G2S_ERR_ARG;
  g2s::BamFile bam;
  std::string err;
  if (!bam.open_mem(bam_bytes, n, &err)) { rf_error = err; return G2S_ERR_IO; }
  return run_filter(bam, o, fasta_out, log_out, warn_out, extracted, total);
}

}  // extern "C"
