// gap2seq_amd/csrc/bam.cpp — see bam.hpp.
#include "bam.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <thread>

namespace g2s {

namespace {
inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }  // (little-endian host)
// inflated bytes per refill (G2S_BAM_CHUNK: the tests make it small so that records span refills)
size_t chunk_bytes() {
  static const size_t v = [] {
    const char* e = getenv("G2S_BAM_CHUNK");
    const long long n = e ? atoll(e) : 0;
    return n > 0 ? (size_t)n : (size_t)32 << 20;
  }();
  return v;
}
}  // namespace

BamFile::~BamFile() {
  if (map_) munmap(map_, map_len_);
}

bool BamFile::open_path(const std::string& path, std::string* err) {
  const int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) { *err = "cannot open " + path; return false; }
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size <= 0) { ::close(fd); *err = "cannot read " + path; return false; }
  void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  ::close(fd);
  if (m == MAP_FAILED) { *err = "cannot map " + path; return false; }
  map_ = m;
  map_len_ = (size_t)st.st_size;
  madvise(m, map_len_, MADV_SEQUENTIAL);
  return open_mem(m, map_len_, err);
}

bool BamFile::open_mem(const void* bytes, size_t n, std::string* err) {
  data_ = (const uint8_t*)bytes;
  size_ = n;
  return index_blocks(err) && read_header(err);
}

// BGZF (SAM specification 4.1): gzip members with an extra subfield 'B','C' holding the member's size - 1
bool BamFile::index_blocks(std::string* err) {
  blk_off_.clear(); blk_csize_.clear(); blk_isize_.clear(); blk_dataoff_.clear();
  size_t o = 0;
  while (o < size_) {
    if (o + 18 > size_ || data_[o] != 0x1f || data_[o + 1] != 0x8b || data_[o + 2] != 8 || !(data_[o + 3] & 4)) {
      *err = "not a BGZF file (block " + std::to_string(blk_off_.size()) + ")";
      return false;
    }
    const size_t xlen = rd16(data_ + o + 10);
    if (o + 12 + xlen > size_) { *err = "truncated BGZF header"; return false; }
    size_t bsize = 0;
    for (size_t x = o + 12; x + 4 <= o + 12 + xlen;) {
      const size_t slen = rd16(data_ + x + 2);
      if (data_[x] == 'B' && data_[x + 1] == 'C' && slen == 2 && x + 6 <= o + 12 + xlen) bsize = (size_t)rd16(data_ + x + 4) + 1;
      x += 4 + slen;
    }
    if (bsize < 12 + xlen + 8 || o + bsize > size_) { *err = "truncated BGZF block"; return false; }
    const uint32_t isize = rd32(data_ + o + bsize - 4);
    if (isize > 65536) { *err = "BGZF block larger than 64 KiB"; return false; }
    blk_off_.push_back(o);
    blk_csize_.push_back((uint32_t)bsize);
    blk_isize_.push_back(isize);
    blk_dataoff_.push_back((uint16_t)(12 + xlen));
    o += bsize;
  }
  if (blk_off_.empty()) { *err = "empty file"; return false; }
  return true;
}

// the inflated stream, a window at a time
struct BamFile::Stream {
  const BamFile& f;
  size_t next_blk = 0;
  std::vector<uint8_t> buf;
  size_t lo = 0, hi = 0;
  std::string err;
  explicit Stream(const BamFile& file) : f(file) {}

  static bool inflate_block(z_stream* zs, const BamFile& f, size_t b, uint8_t* out) {
    const uint8_t* blk = f.data_ + f.blk_off_[b];
    const uint32_t isize = f.blk_isize_[b];
    if (isize == 0) return true;  // (the end-of-file marker, or an empty block)
    if (inflateReset(zs) != Z_OK) return false;
    zs->next_in = const_cast<Bytef*>(blk + f.blk_dataoff_[b]);
    zs->avail_in = f.blk_csize_[b] - f.blk_dataoff_[b] - 8;
    zs->next_out = out;
    zs->avail_out = isize;
    if (inflate(zs, Z_FINISH) != Z_STREAM_END || zs->avail_out != 0) return false;
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), out, isize) == rd32(blk + f.blk_csize_[b] - 8);
  }

  bool refill() {
    const size_t nb = f.blk_off_.size();
    if (next_blk >= nb || !err.empty()) return false;
    if (lo) { memmove(buf.data(), buf.data() + lo, hi - lo); hi -= lo; lo = 0; }
    size_t e = next_blk, bytes = 0;
    while (e < nb && (e == next_blk || bytes + f.blk_isize_[e] <= chunk_bytes())) bytes += f.blk_isize_[e++];
    if (buf.size() < hi + bytes) buf.resize(hi + bytes);
    std::vector<size_t> at(e - next_blk);
    for (size_t b = next_blk, o = hi; b < e; b++) { at[b - next_blk] = o; o += f.blk_isize_[b]; }
    const int T = (int)std::min<size_t>((size_t)f.threads_, std::max<size_t>(1, (e - next_blk) / 16));
    std::atomic<size_t> bad{(size_t)-1};
    auto work = [&](int t) {
      z_stream zs;
      memset(&zs, 0, sizeof zs);
      if (inflateInit2(&zs, -15) != Z_OK) { bad = next_blk; return; }
      for (size_t b = next_blk + (size_t)t; b < e; b += (size_t)T)
        if (!inflate_block(&zs, f, b, buf.data() + at[b - next_blk])) { bad = b; break; }
      inflateEnd(&zs);
    };
    if (T <= 1) work(0);
    else {
      std::vector<std::thread> th;
      for (int t = 1; t < T; t++) th.emplace_back(work, t);
      work(0);
      for (auto& x : th) x.join();
    }
    if (bad.load() != (size_t)-1) { err = "corrupt BGZF block " + std::to_string(bad.load()); return false; }
    hi += bytes;
    next_blk = e;
    return true;
  }
  // n contiguous bytes at the read position, or null at the end of the stream / on an error
  const uint8_t* need(size_t n) {
    while (hi - lo < n)
      if (!refill()) return nullptr;
    return buf.data() + lo;
  }
  void consume(size_t n) { lo += n; }
  bool skip(uint64_t n) {
    while (n) {
      const size_t step = (size_t)std::min<uint64_t>(n, 1 << 16);
      if (!need(step)) return false;
      consume(step);
      n -= step;
    }
    return true;
  }
  size_t left() const { return hi - lo; }
};

// BAM header (SAM specification 4.2): magic, SAM text, reference names and lengths
bool BamFile::read_header(std::string* err) {
  Stream s(*this);
  auto fail = [&](const char* what) { *err = s.err.empty() ? what : s.err; return false; };
  const uint8_t* p = s.need(12);
  if (!p || memcmp(p, "BAM\1", 4) != 0) return fail("not a BAM file");
  const uint32_t l_text = rd32(p + 4);
  s.consume(8);
  uint64_t used = 8;
  if (!s.skip(l_text)) return fail("truncated BAM header");
  used += l_text;
  p = s.need(4);
  if (!p) return fail("truncated BAM header");
  const uint32_t n_ref = rd32(p);
  s.consume(4);
  used += 4;
  ref_names_.clear();
  for (uint32_t r = 0; r < n_ref; r++) {
    p = s.need(4);
    if (!p) return fail("truncated BAM header");
    const uint32_t l_name = rd32(p);
    if (l_name == 0 || l_name > (1u << 20)) return fail("bad reference name in the BAM header");
    p = s.need(4 + (size_t)l_name + 4);
    if (!p) return fail("truncated BAM header");
    ref_names_.emplace_back((const char*)p + 4, strnlen((const char*)p + 4, l_name));
    s.consume(4 + (size_t)l_name + 4);
    used += 4 + (uint64_t)l_name + 4;
  }
  first_rec_ = used;
  return true;
}

int BamFile::ref_id(const std::string& name) const {
  for (size_t i = 0; i < ref_names_.size(); i++)
    if (ref_names_[i] == name) return (int)i;
  return -1;
}

bool BamFile::for_each(const std::function<bool(const BamRec&)>& fn, std::string* err) const {
  Stream s(*this);
  auto fail = [&](const char* what) { *err = s.err.empty() ? what : s.err; return false; };
  if (!s.skip(first_rec_)) return fail("truncated BAM header");
  for (;;) {
    const uint8_t* p = s.need(4);
    if (!p) {
      if (!s.err.empty() || s.left() != 0) return fail("truncated BAM record");
      return true;
    }
    const uint32_t bs = rd32(p);
    if (bs < 32 || bs > (1u << 30)) return fail("bad BAM record size");
    p = s.need(4 + (size_t)bs);
    if (!p) return fail("truncated BAM record");
    p += 4;
    BamRec r;
    r.ref_id = (int32_t)rd32(p);
    r.pos = (int32_t)rd32(p + 4);
    r.l_name = p[8];
    r.mapq = p[9];
    r.n_cigar = rd16(p + 12);
    r.flag = rd16(p + 14);
    r.l_seq = (int32_t)rd32(p + 16);
    r.next_ref_id = (int32_t)rd32(p + 20);
    r.next_pos = (int32_t)rd32(p + 24);
    r.tlen = (int32_t)rd32(p + 28);
    if (r.l_seq < 0 || r.l_name == 0 ||
        32 + (uint64_t)r.l_name + 4 * (uint64_t)r.n_cigar + ((uint64_t)r.l_seq + 1) / 2 + (uint64_t)r.l_seq > bs)
      return fail("bad BAM record layout");
    r.name = (const char*)p + 32;
    r.cigar = p + 32 + r.l_name;
    r.seq = r.cigar + 4 * (size_t)r.n_cigar;
    if (!fn(r)) return true;
    s.consume(4 + (size_t)bs);
  }
}

}  // namespace g2s
