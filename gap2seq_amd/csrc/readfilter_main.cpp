// gap2seq_amd/csrc/readfilter_main.cpp — `ReadFilter` command line, drop-in for the reference's options
// (/root/reference/src/ReadFilter.cpp:49-62,259-276; called by Gap2Seq.py:64-72 and :145-149).  The work is
// g2s_filter_reads (readfilter.cpp).  Like the reference, a run that extracts nothing leaves no output file
// (Gap2Seq.py:151-153 relies on it) and a BAM that cannot be read is reported on stderr with exit status 0
// (ReadFilter.cpp:361-365,417-421).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "../../include/g2s.h"
#include "fastx.hpp"

int main(int argc, char** argv) {
  g2s_filter_opts o;
  memset(&o, 0, sizeof o);
  o.gap_length = -1;
  o.flank_length = -1;
  std::string bam, reads, scaffold;
  int fasta_width = 0;
  bool saw_mean = false, saw_sd = false, saw_scaffold = false, saw_bp = false;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto val = [&]() -> const char* { return (i + 1 < argc) ? argv[++i] : ""; };
    if (a == "-bam") bam = val();
    else if (a == "-reads") reads = val();
    else if (a == "-mean") { o.mean_insert = atoi(val()); saw_mean = true; }
    else if (a == "-std-dev") { o.std_dev = atoi(val()); saw_sd = true; }
    else if (a == "-scaffold") { scaffold = val(); saw_scaffold = true; }
    else if (a == "-breakpoint") { o.breakpoint = atoi(val()); saw_bp = true; }
    else if (a == "-gap-length") o.gap_length = atoi(val());
    else if (a == "-flank-length") o.flank_length = atoi(val());
    else if (a == "-unmapped-only") o.unmapped_only = 1;
    else if (a == "-fasta-width") fasta_width = atoi(val());
    else if (a == "-nb-cores") o.threads = atoi(val());
    else if (a == "-verbose") (void)val();
    else { std::cout << "EXCEPTION: Unknown parameter '" << a << "'" << std::endl; return EXIT_FAILURE; }
  }
  if (bam.empty() || reads.empty() || !saw_mean || !saw_sd || !saw_scaffold || !saw_bp) {
    std::cout << "EXCEPTION: missing mandatory option (-bam, -reads, -mean, -std-dev, -scaffold, -breakpoint)" << std::endl;
    return EXIT_FAILURE;
  }
  o.scaffold = scaffold.c_str();
  char *fasta = nullptr, *log = nullptr, *warn = nullptr;
  int64_t extracted = 0, total = 0;
  if (g2s_filter_reads(bam.c_str(), &o, &fasta, &log, &warn, &extracted, &total) != G2S_OK) {
    std::cerr << "Error loading alignments" << std::endl;  // :362-365
    std::cerr << g2s_filter_last_error() << std::endl;
    return EXIT_SUCCESS;
  }
  std::cerr << warn;
  if (extracted > 0) {
    FILE* f = fopen(reads.c_str(), "wb");
    if (!f) { std::cout << "EXCEPTION: cannot write " << reads << std::endl; return EXIT_FAILURE; }
    const std::string w = g2s::wrap_fasta(fasta, strlen(fasta), fasta_width);
    fwrite(w.data(), 1, w.size(), f);
    fclose(f);
  }
  std::cout << log;
  g2s_free(fasta); g2s_free(log); g2s_free(warn);
  return EXIT_SUCCESS;
}
