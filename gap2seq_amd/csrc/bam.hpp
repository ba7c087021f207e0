// gap2seq_amd/csrc/bam.hpp — a BAM reader for the read filter (readfilter.cpp): BGZF blocks
// inflated with zlib, several blocks at a time on a few threads, records handed to a callback
// in file order.  It replaces the part of htslib the reference's ReadFilter uses
// (/root/reference/src/ReadFilter.cpp:66-101 io_t, :176-222 sam_iterator): open, header,
// "every record" and "records overlapping [beg, end) of one reference".  No index is read:
// the reference's run makes two passes over the whole file anyway (:225-241 count_reads,
// :313-323 find_mates), so region queries are answered by the same kind of pass with htslib's
// overlap test (tid equal, pos < end, end position > beg).
// Formats: SAM/BAM specification, sections 4.1 (BGZF) and 4.2 (BAM).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

namespace g2s {

enum : uint32_t {
  BAM_PAIRED = 1, BAM_UNMAPPED = 4, BAM_MATE_UNMAPPED = 8, BAM_REVERSE = 16, BAM_READ1 = 64, BAM_READ2 = 128
};

// one alignment record, pointing into the reader's buffer (valid during the callback only)
struct BamRec {
  int32_t ref_id, pos, next_ref_id, next_pos, tlen, l_seq;
  uint32_t flag, n_cigar, l_name;
  uint8_t mapq;
  const char* name;        // l_name bytes, NUL terminated by the format
  const uint8_t* cigar;    // n_cigar little-endian words: length << 4 | op
  const uint8_t* seq;      // 4-bit codes "=ACMGRSVTWYHKDBN", high nibble first
  // htslib's bam_endpos: position after the last reference base the alignment covers; an unmapped record
  // or one whose CIGAR consumes no reference counts as one base
  int64_t end_pos() const {
    int64_t rlen = 0;
    if (!(flag & BAM_UNMAPPED))
      for (uint32_t i = 0; i < n_cigar; i++) {
        uint32_t w;
        memcpy(&w, cigar + 4 * (size_t)i, 4);
        const uint32_t op = w & 15;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += w >> 4;  // M D N = X
      }
    return (int64_t)pos + (rlen ? rlen : 1);
  }
  uint8_t base4(int32_t i) const { return (uint8_t)((seq[i >> 1] >> ((~i & 1) << 2)) & 15); }
};

class BamFile {
 public:
  BamFile() {}
  ~BamFile();
  BamFile(const BamFile&) = delete;
  BamFile& operator=(const BamFile&) = delete;
  // map the file / adopt the caller's bytes, index the BGZF blocks, read the header
  bool open_path(const std::string& path, std::string* err);
  bool open_mem(const void* bytes, size_t n, std::string* err);
  const std::vector<std::string>& ref_names() const { return ref_names_; }
  int ref_id(const std::string& name) const;  // -1 when the header has no such reference
  // every record in file order; stops early (returning true) when fn returns false.  False = broken file.
  bool for_each(const std::function<bool(const BamRec&)>& fn, std::string* err) const;
  void set_threads(int t) { threads_ = t < 1 ? 1 : t; }
  size_t blocks() const { return blk_off_.size(); }

 private:
  bool index_blocks(std::string* err);
  bool read_header(std::string* err);
  struct Stream;
  const uint8_t* data_ = nullptr;
  size_t size_ = 0;
  void* map_ = nullptr;
  size_t map_len_ = 0;
  std::vector<uint64_t> blk_off_;     // offset of each BGZF block
  std::vector<uint32_t> blk_csize_;   // its size in the file
  std::vector<uint32_t> blk_isize_;   // and inflated
  std::vector<uint16_t> blk_dataoff_; // offset of the deflate stream inside the block
  std::vector<std::string> ref_names_;
  uint64_t first_rec_ = 0;            // inflated offset of the first alignment record
  int threads_ = 4;
};

}  // namespace g2s
