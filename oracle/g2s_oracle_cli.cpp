// oracle/g2s_oracle_cli.cpp — Gap2Seq-core-shaped command line over the CPU
// oracle.  TEST INFRASTRUCTURE ONLY; parity unpinned by the reference.
// Accepts the argv the reference wrapper builds (Gap2Seq.py:178-188,230-241).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <thread>

#include "g2s_oracle.hpp"

using namespace orc;

int main(int argc, char** argv) {
  Params p;
  std::string reads, scaffolds, filled, left, right;
  int length = -1;
  bool saw_left = false, saw_right = false, saw_len = false;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto val = [&]() -> const char* { return (i + 1 < argc) ? argv[++i] : ""; };
    if (a == "-k") p.k = atoi(val());
    else if (a == "-solid") p.solid = atoi(val());
    else if (a == "-reads") reads = val();
    else if (a == "-scaffolds") scaffolds = val();
    else if (a == "-filled") filled = val();
    else if (a == "-dist-error") p.d_err = atoi(val());
    else if (a == "-fuz") p.max_fuz = atoi(val());
    else if (a == "-max-mem") p.max_mem_gb = atof(val());
    else if (a == "-all-upper") p.skip_confident = true;
    else if (a == "-best-only") p.all_paths = false;
    else if (a == "-unique") p.unique_paths = true;
    else if (a == "-randseed") p.randseed = atoi(val());
    else if (a == "-nb-cores") {  // 0 = all cores (GATB Tool); only divides -max-mem (Gap2Seq.cpp:302)
      p.nb_cores = atoi(val());
      if (p.nb_cores <= 0) p.nb_cores = (int)std::max(1u, std::thread::hardware_concurrency());
    }
    else if (a == "-left") { left = val(); saw_left = true; }
    else if (a == "-right") { right = val(); saw_right = true; }
    else if (a == "-length") { length = atoi(val()); saw_len = true; }
    else if (a == "-verbose") (void)val();
  }
  if (reads.empty() || filled.empty()) {
    std::cout << "EXCEPTION: missing mandatory option (-reads, -filled)" << std::endl;
    return EXIT_FAILURE;
  }
  std::vector<std::string> files;
  { std::stringstream ss(reads); std::string f; while (std::getline(ss, f, ',')) files.push_back(f); }
  GraphBase* g = graph_from_files(files, p.k, p.solid);
  if (!g) { std::cout << "DBG building failed: cannot read " << reads << std::endl; return EXIT_FAILURE; }
  std::string fasta, log;
  int q7_gaps = 0;  // gaps whose outcome depends on libstdc++'s hash-set order, reported on stderr
  if (saw_left && saw_right && saw_len) {
    execute_single(g, p, reads, filled, left, right, length, &fasta, &log);
  } else {
    std::string text;
    if (!read_file(scaffolds, &text)) { std::cout << "EXCEPTION: cannot open " << scaffolds << std::endl; return EXIT_FAILURE; }
    ExecSummary es;
    execute_scaffolds(g, p, reads, filled, text, &fasta, &log, &es);
    q7_gaps = es.q7_gaps;
  }
  std::cout << log;
  std::cerr << "# oracle: q7_gaps " << q7_gaps << std::endl;
  std::ofstream out(filled.c_str());
  out << fasta;
  graph_free(g);
  return EXIT_SUCCESS;
}
