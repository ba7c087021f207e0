// gap2seq_amd/csrc/fastx.cpp — see fastx.hpp.
#include "fastx.hpp"

#include <cstdio>

namespace g2s {

bool read_text_file(const std::string& path, std::string* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  out->clear();
  char buf[1 << 16];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, got);
  fclose(f);
  return true;
}

namespace {
struct LineReader {
  const std::string& t;
  size_t pos = 0;
  explicit LineReader(const std::string& text) : t(text) {}
  bool next(const char** p, size_t* len) {
    if (pos >= t.size()) return false;
    size_t e = t.find('\n', pos);
    if (e == std::string::npos) e = t.size();
    *p = t.data() + pos;
    *len = e - pos;
    if (*len > 0 && (*p)[*len - 1] == '\r') (*len)--;
    pos = e + 1;
    return true;
  }
};
}  // namespace

void parse_fastx(const std::string& text, std::vector<FastxRecord>* out) {
  LineReader lr(text);
  const char* p;
  size_t len;
  FastxRecord* cur = nullptr;
  int fastq_line = -1;  // >=0 while inside a 4-line FASTQ record
  while (lr.next(&p, &len)) {
    if (fastq_line >= 0) {
      if (fastq_line == 0) cur->seq.assign(p, len);
      if (++fastq_line == 3) { fastq_line = -1; cur = nullptr; }
      continue;
    }
    if (len == 0) continue;
    if (p[0] == '>') {
      out->emplace_back();
      cur = &out->back();
      cur->comment.assign(p + 1, len - 1);
    } else if (p[0] == '@' && cur == nullptr) {
      out->emplace_back();
      cur = &out->back();
      cur->comment.assign(p + 1, len - 1);
      fastq_line = 0;
    } else if (cur) {
      cur->seq.append(p, len);
    }
  }
}

}  // namespace g2s
