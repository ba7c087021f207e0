// gap2seq_amd/csrc/fastx.hpp — FASTA/FASTQ text handling for the host side.
// Stands in for gatb BankFasta / BankAlbum as used at
// /root/reference/src/Gap2Seq.cpp:213 (reads), :224,:276-279,:426-431 (output),
// :287-289,:316-318 (scaffold iterator): multi-line records are joined, case is
// kept, the comment is the header line without '>', output is one line per
// sequence.
#pragma once
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

}  // namespace g2s
