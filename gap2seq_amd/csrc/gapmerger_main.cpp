// gap2seq_amd/csrc/gapmerger_main.cpp — `GapMerger` command line, drop-in for the reference's
// options (/root/reference/src/GapMerger.cpp:30-33,66-71; called by Gap2Seq.py:321-326 as
// GapMerger -scaffolds OUT -gaps FILLED -contigs C).  The work is g2s_merge_scaffolds (gapio.cpp).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "../../include/g2s.h"
#include "fastx.hpp"

static bool slurp(const std::string& path, std::string* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[1 << 16];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, got);
  fclose(f);
  return true;
}

int main(int argc, char** argv) {
  std::string scaffolds, contigs, gaps;
  int fasta_width = 0;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto val = [&]() -> const char* { return (i + 1 < argc) ? argv[++i] : ""; };
    if (a == "-scaffolds") scaffolds = val();
    else if (a == "-contigs") contigs = val();
    else if (a == "-gaps") gaps = val();
    else if (a == "-fasta-width") fasta_width = atoi(val());
    else if (a == "-nb-cores" || a == "-verbose") (void)val();
    else { std::cout << "EXCEPTION: Unknown parameter '" << a << "'" << std::endl; return EXIT_FAILURE; }
  }
  if (scaffolds.empty() || contigs.empty() || gaps.empty()) {
    std::cout << "EXCEPTION: missing mandatory option (-scaffolds, -contigs, -gaps)" << std::endl;
    return EXIT_FAILURE;
  }
  std::string ctext, gtext;
  if (!slurp(contigs, &ctext)) { std::cout << "EXCEPTION: cannot open " << contigs << std::endl; return EXIT_FAILURE; }
  if (!slurp(gaps, &gtext)) { std::cout << "EXCEPTION: cannot open " << gaps << std::endl; return EXIT_FAILURE; }
  char *out = nullptr, *log = nullptr;
  if (g2s_merge_scaffolds(ctext.c_str(), gtext.c_str(), scaffolds.c_str(), contigs.c_str(), gaps.c_str(), &out, &log) != G2S_OK) {
    std::cout << "EXCEPTION: bad arguments" << std::endl;
    return EXIT_FAILURE;
  }
  std::cout << log;
  FILE* f = fopen(scaffolds.c_str(), "wb");
  if (!f) { std::cout << "EXCEPTION: cannot write " << scaffolds << std::endl; return EXIT_FAILURE; }
  {
    const std::string w = g2s::wrap_fasta(out, strlen(out), fasta_width);  // (see fastx.hpp)
    fwrite(w.data(), 1, w.size(), f);
  }
  fclose(f);
  g2s_free(out); g2s_free(log);
  return EXIT_SUCCESS;
}
