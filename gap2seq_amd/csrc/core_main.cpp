// gap2seq_amd/csrc/core_main.cpp — `Gap2Seq-core` command line, drop-in for the
// reference binary's options (/root/reference/src/Gap2Seq.cpp:51-67,75-90 and
// GATB Tool's -nb-cores/-verbose; main: /root/reference/src/main.cpp:25-35).
// It accepts exactly the argv the reference wrapper builds
// (/root/reference/src/Gap2Seq.py:178-188,230-241).  Everything below the option
// parsing goes through the C ABI in include/g2s.h.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../include/g2s.h"
#include "fastx.hpp"

static bool readable(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fclose(f);
  return true;
}

int main(int argc, char** argv) {
  g2s_run_opts o;
  o.k = 31; o.solid = 2; o.max_fuz = 10; o.nb_cores = 0; o.max_mem_gb = 20.0;
  g2s_params p;
  memset(&p, 0, sizeof p);
  p.d_err = 500; p.all_paths = 1;
  int randseed = 0, device = 0, streams = 2, stream_gaps = 8192;
  bool streams_given = false;
  static int fasta_width = 0;  // (static: read by the output callback)
  std::string devices;  // "0,1,2": GPUs sharing the gap list (the graph is replicated)
  std::string reads, scaffolds, filled, left, right;
  int length = 0;
  bool saw_left = false, saw_right = false, saw_len = false;
  if (const char* e = getenv("G2S_DEVICE")) device = atoi(e);
  if (const char* e = getenv("G2S_DEVICES")) devices = e;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto val = [&]() -> const char* { return (i + 1 < argc) ? argv[++i] : ""; };
    if (a == "-k") o.k = atoi(val());
    else if (a == "-solid") o.solid = atoi(val());
    else if (a == "-reads") reads = val();
    else if (a == "-scaffolds") scaffolds = val();
    else if (a == "-filled") filled = val();
    else if (a == "-dist-error") p.d_err = atoi(val());
    else if (a == "-fuz") o.max_fuz = atoi(val());
    else if (a == "-max-mem") o.max_mem_gb = atof(val());
    else if (a == "-all-upper") p.skip_confident = 1;
    else if (a == "-best-only") p.all_paths = 0;
    else if (a == "-unique") p.unique_paths = 1;
    else if (a == "-randseed") randseed = atoi(val());
    else if (a == "-nb-cores") o.nb_cores = atoi(val());
    else if (a == "-left") { left = val(); saw_left = true; }
    else if (a == "-right") { right = val(); saw_right = true; }
    else if (a == "-length") { length = atoi(val()); saw_len = true; }
    else if (a == "-verbose") (void)val();
    else if (a == "-version") { std::cout << "Gap2Seq-core (MI355X) ABI " << G2S_ABI_VERSION << std::endl; return EXIT_SUCCESS; }
    else if (a == "-device") device = atoi(val());
    else if (a == "-devices") devices = val();
    else if (a == "-streams") { streams = atoi(val()); streams_given = true; }
    else if (a == "-stream-gaps") stream_gaps = atoi(val());
    else if (a == "-fasta-width") fasta_width = atoi(val());
    else if (a == "-help" || a == "-h") {
      std::cout << "Gap2Seq-core (MI355X) -reads a.fq[,b.fq] -filled out.fa (-scaffolds in.fa | -left S -right S -length N)\n"
                   "  [-k 31] [-solid 2] [-dist-error 500] [-fuz 10] [-max-mem 20] [-randseed 0]\n"
                   "  [-all-upper] [-best-only] [-unique] [-nb-cores N] [-device D | -devices D0,D1,...] [-streams 2]\n"
                   "  [-stream-gaps 8192] [-fasta-width 0]\n";
      return EXIT_SUCCESS;
    }
    else {  // GATB's OptionsParser rejects what it does not know; main.cpp:29-31 prints the message
      std::cout << "EXCEPTION: Unknown parameter '" << a << "'" << std::endl;
      return EXIT_FAILURE;
    }
  }
  if (reads.empty() || filled.empty()) {
    std::cout << "EXCEPTION: missing mandatory option (-reads, -filled)" << std::endl;  // main.cpp:29-31
    return EXIT_FAILURE;
  }
  // -max-mem is divided by the number of threads (:302) — by an EXPLICIT -nb-cores here: with the option omitted
  // (GATB: all cores) the reference's per-gap budget depends on the host's CPU count, which says nothing about
  // what a GPU can hold; the whole budget then applies per gap (the device-budget analogue D3)
  const int mem_div = std::max(1, o.nb_cores);
  const bool cores_given = o.nb_cores > 0;
  if (!cores_given) o.nb_cores = (int)std::max(1u, std::thread::hardware_concurrency());
  p.max_mem = (int64_t)(o.max_mem_gb * 1024 * 1024 * 1024) / mem_div;
  if (!cores_given)  // (the "Max mem:" line of the log keeps the reference's formula, :302-303; this is what applies)
    std::cerr << "[g2s] -nb-cores not given: the whole -max-mem (" << (long long)p.max_mem
              << " bytes) is the per-gap device budget; the log's \"Max mem:\" line is -max-mem / " << o.nb_cores
              << " host CPUs as the reference prints it" << std::endl;
  // the session seeds with time(NULL) when this is 0 (:178); the echo prints the user's value (:191)
  p.randseed = randseed > 0 ? (uint32_t)randseed : 0u;
  p.host_threads = 0;

  {  // the graph is built on the first GPU this run uses
    int build_dev = device;
    if (!devices.empty()) build_dev = atoi(devices.c_str());
    setenv("G2S_DEVICE", std::to_string(build_dev).c_str(), 1);
  }
  g2s_graph* g = nullptr;
  const std::string cache = reads + ".g2s";  // the reference reuses "<reads>.h5" (:171,195-197)
  int rc;
  if (readable(cache)) {
    std::cout << "Loading from " << cache << std::endl;
    rc = g2s_graph_load(cache.c_str(), &g);
    if (rc == G2S_OK && (g2s_graph_k(g) != o.k || g2s_graph_solid(g) != o.solid)) {  // a cache of another -k or -solid is not this run's graph
      g2s_graph_free(g);
      g = nullptr;
      rc = g2s_graph_build_files(reads.c_str(), o.k, o.solid, 0, &g);
    }
  } else {
    rc = g2s_graph_build_files(reads.c_str(), o.k, o.solid, 0, &g);
    // like Graph::create leaving "<reads>.h5" behind for the next run (:195-197); opt-in here
    // because the file is as large as the graph (32 B per k-mer)
    if (rc == G2S_OK && getenv("G2S_SAVE_GRAPH")) (void)g2s_graph_save(g, cache.c_str());
  }
  if (rc != G2S_OK) {
    std::cout << "DBG building failed: " << g2s_last_error() << std::endl;  // :215-218
    return EXIT_FAILURE;
  }
  g2s_session* s = nullptr;
  rc = g2s_session_create(g, device, &p, &s);
  if (rc != G2S_OK) {
    std::cout << "EXCEPTION: " << g2s_last_error() << std::endl;
    return EXIT_FAILURE;
  }
  // the dispatcher: `streams` sessions on every listed device pull groups of gaps from one
  // list (g2s_team_fill); the first session of the first device owns the rand() stream
  std::vector<int> devs;
  for (size_t pos = 0; pos < devices.size();) {
    size_t e = devices.find(',', pos);
    if (e == std::string::npos) e = devices.size();
    if (e > pos) devs.push_back(atoi(devices.substr(pos, e - pos).c_str()));
    pos = e + 1;
  }
  if (devs.empty()) devs.push_back(device);
  else if (devs[0] != device) {  // the lead session sits on the first listed device
    g2s_session_destroy(s);
    s = nullptr;
    rc = g2s_session_create(g, devs[0], &p, &s);
    if (rc != G2S_OK) { std::cout << "EXCEPTION: " << g2s_last_error() << std::endl; return EXIT_FAILURE; }
  }
  // Several GPUs: one session per GPU, and every batch of the stream is -stream-gaps gaps PER GPU, cut into one share
  // per GPU — each GPU fills, traces and writes a whole share, the rand() stream chained from share to share (phase D3
  // sharded) — unless -streams asks for the round-2 dispatcher (several sessions a GPU pulling small groups).
  const bool per_gpu_shares = devs.size() > 1 && !streams_given;
  if (per_gpu_shares) streams = 1;
  std::vector<g2s_session*> helpers;
  for (size_t d = 0; d < devs.size(); d++)
    for (int t = (d == 0 ? 1 : 0); t < std::max(1, streams); t++) {
      g2s_session* h = nullptr;
      rc = g2s_session_create(g, devs[d], &p, &h);
      if (rc != G2S_OK) { std::cout << "EXCEPTION: " << g2s_last_error() << std::endl; return EXIT_FAILURE; }
      helpers.push_back(h);
    }
  if (!helpers.empty()) g2s_session_set_team(s, helpers.data(), (int)helpers.size(), per_gpu_shares ? G2S_GROUP_PER_SESSION : 0);
  if (per_gpu_shares) stream_gaps = (int)std::min<long long>((long long)std::max(0, stream_gaps) * (long long)devs.size(), 1 << 24);
  char *fasta = nullptr, *log = nullptr;
  if (saw_left && saw_right && saw_len) {
    rc = g2s_execute_single(s, &o, reads.c_str(), filled.c_str(), left.c_str(), right.c_str(), length, &fasta, &log);
  } else {
    FILE* f = fopen(scaffolds.c_str(), "rb");
    if (!f) { std::cout << "EXCEPTION: cannot open " << scaffolds << std::endl; return EXIT_FAILURE; }
    std::string text;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, got);
    fclose(f);
    int32_t gaps = 0, nfilled = 0;
    // the log is printed and the records are written as the batches finish (-stream-gaps N per batch)
    // (written beside the target and renamed when the run is complete: a failed run leaves no partial -filled file)
    const std::string partial = filled + ".partial";
    FILE* out = fopen(partial.c_str(), "wb");
    if (!out) { std::cout << "EXCEPTION: cannot write " << filled << std::endl; return EXIT_FAILURE; }
    rc = g2s_execute_scaffolds_stream(
        s, &o, reads.c_str(), filled.c_str(), text.c_str(), (size_t)std::max(0, stream_gaps),
        [](const char* t, size_t n, void* u) {
          const std::string w = g2s::wrap_fasta(t, n, fasta_width);
          fwrite(w.data(), 1, w.size(), (FILE*)u);
        },
        [](const char* t, size_t n, void*) { fwrite(t, 1, n, stdout); fflush(stdout); }, out, &gaps, &nfilled);
    fclose(out);
    if (rc != G2S_OK) {
      remove(partial.c_str());
      std::cout << "EXCEPTION: " << g2s_last_error() << std::endl;
      return EXIT_FAILURE;
    }
    if (rename(partial.c_str(), filled.c_str()) != 0) { std::cout << "EXCEPTION: cannot write " << filled << std::endl; return EXIT_FAILURE; }
    g2s_session_destroy(s);
    for (g2s_session* h : helpers) g2s_session_destroy(h);
    g2s_graph_free(g);
    return EXIT_SUCCESS;
  }
  if (rc != G2S_OK) {
    std::cout << "EXCEPTION: " << g2s_last_error() << std::endl;
    return EXIT_FAILURE;
  }
  std::cout << log;
  FILE* out = fopen(filled.c_str(), "wb");
  if (!out) { std::cout << "EXCEPTION: cannot write " << filled << std::endl; return EXIT_FAILURE; }
  {
    const std::string w = g2s::wrap_fasta(fasta, strlen(fasta), fasta_width);
    fwrite(w.data(), 1, w.size(), out);
  }
  fclose(out);
  g2s_free(fasta);
  g2s_free(log);
  g2s_session_destroy(s);
  for (g2s_session* h : helpers) g2s_session_destroy(h);
  g2s_graph_free(g);
  return EXIT_SUCCESS;
}
