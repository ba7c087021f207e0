// gap2seq_amd/csrc/gapcutter_main.cpp — `GapCutter` command line, drop-in for the reference's
// options (/root/reference/src/GapCutter.cpp:36-43,71-79; called by Gap2Seq.py:301-309 as
// GapCutter -k K -fuz F -scaffolds S -gaps G -contigs C -bed B [-mask]).  The work is
// g2s_cut_scaffolds (gapio.cpp).
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
static bool spill(const std::string& path, const char* text, int fasta_width = 0) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  if (fasta_width > 0) {  // (see fastx.hpp: GATB's BankFasta is recalled to break data lines at 70 columns)
    const std::string w = g2s::wrap_fasta(text, strlen(text), fasta_width);
    fwrite(w.data(), 1, w.size(), f);
  } else {
    fputs(text, f);
  }
  fclose(f);
  return true;
}

int main(int argc, char** argv) {
  int k = 31, fuz = 10, mask = 0, no_split = 0, fasta_width = 0;
  std::string scaffolds, contigs, gaps, bed;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto val = [&]() -> const char* { return (i + 1 < argc) ? argv[++i] : ""; };
    if (a == "-k") k = atoi(val());
    else if (a == "-fuz") fuz = atoi(val());
    else if (a == "-scaffolds") scaffolds = val();
    else if (a == "-contigs") contigs = val();
    else if (a == "-gaps") gaps = val();
    else if (a == "-bed") bed = val();
    else if (a == "-mask") mask = 1;
    else if (a == "-no-split") no_split = 1;
    else if (a == "-fasta-width") fasta_width = atoi(val());
    else if (a == "-nb-cores" || a == "-verbose") (void)val();
    else { std::cout << "EXCEPTION: Unknown parameter '" << a << "'" << std::endl; return EXIT_FAILURE; }
  }
  if (scaffolds.empty() || contigs.empty() || gaps.empty() || bed.empty()) {
    std::cout << "EXCEPTION: missing mandatory option (-scaffolds, -contigs, -gaps, -bed)" << std::endl;
    return EXIT_FAILURE;
  }
  std::string text;
  if (!slurp(scaffolds, &text)) { std::cout << "EXCEPTION: cannot open " << scaffolds << std::endl; return EXIT_FAILURE; }
  char *c = nullptr, *g = nullptr, *b = nullptr, *log = nullptr;
  if (g2s_cut_scaffolds(text.c_str(), k, fuz, mask, no_split, scaffolds.c_str(), contigs.c_str(), gaps.c_str(), bed.c_str(), &c, &g,
                        &b, &log) != G2S_OK) {
    std::cout << "EXCEPTION: bad arguments" << std::endl;
    return EXIT_FAILURE;
  }
  std::cout << log;
  const bool ok = spill(contigs, c, fasta_width) && spill(gaps, g, fasta_width) && spill(bed, b);
  g2s_free(c); g2s_free(g); g2s_free(b); g2s_free(log);
  if (!ok) { std::cout << "EXCEPTION: cannot write the output files" << std::endl; return EXIT_FAILURE; }
  return EXIT_SUCCESS;
}
