// gap2seq_amd/csrc/fastx.hpp — FASTA/FASTQ text handling for the host side.
// Stands in for gatb BankFasta / BankAlbum as used at
// /root/reference/src/Gap2Seq.cpp:213 (reads), :224,:276-279,:426-431 (output),
// :287-289,:316-318 (scaffold iterator): multi-line records are joined, case is
// kept, the comment is the header line without '>', output is one line per
// sequence.
#pragma once
#include <algorithm>
#include <string>
#include <vector>

namespace g2s {

struct FastxRecord {
  std::string comment;
  std::string seq;
};

bool read_text_file(const std::string& path, std::string* out);
void parse_fastx(const std::string& text, std::vector<FastxRecord>* out);
inline void append_fasta(std::string* out, const std::string& comment, const std::string& seq) {
  out->push_back('>');
  out->append(comment);
  out->push_back('\n');
  out->append(seq);
  out->push_back('\n');
}

// FASTA text whose records are ">comment\nSEQ\n" -> the same records with SEQ broken into lines of `width`
// characters (width <= 0: unchanged).  GATB's BankFasta, which the reference writes its output through
// (Gap2Seq.cpp:224,426-431), is recalled to break data lines at 70 columns unless told otherwise; that cannot
// be checked here (gatb-core is not part of the reference tree), so the command lines keep one line per
// record by default and offer -fasta-width 70.  Every FASTA reader of the pipeline accepts both.
inline std::string wrap_fasta(const char* text, size_t n, int width) {
  if (width <= 0) return std::string(text, n);
  std::string out;
  out.reserve(n + n / (size_t)width + 16);
  size_t i = 0;
  while (i < n) {
    size_t e = i;
    while (e < n && text[e] != '\n') e++;
    if (text[i] == '>') out.append(text + i, e - i);
    else {
      for (size_t p = i; p < e; p += (size_t)width) {
        if (p > i) out.push_back('\n');
        out.append(text + p, std::min<size_t>((size_t)width, e - p));
      }
    }
    out.push_back('\n');
    i = e + 1;
  }
  return out;
}

}  // namespace g2s
