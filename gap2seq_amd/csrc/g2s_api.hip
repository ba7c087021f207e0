// gap2seq_amd/csrc/g2s_api.hip — implementation of the C ABI in include/g2s.h:
// graph handles, sessions, batches, and the orchestration of the fill path
// (kernels on one HIP stream -> device-to-host of the state logs -> host phase D).
//
// Replaces the call edge Gap2Seq::execute() -> Gap2Seq::fill_gap()
// (/root/reference/src/Gap2Seq.cpp:252,380 -> :858).  No CPU fallback: without a
// usable gfx950 device every fill entry point fails with G2S_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <sys/prctl.h>
#include <pthread.h>
#include <sched.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/g2s.h"
#include "../../include/g2s_test.h"
#include "envcache.hpp"
#include "d2_device.h"
#include "d3_device.h"
#include "dbg.hpp"
#include "fastx.hpp"
#include "fill_device.h"
#include "fill_launch.h"
#include "fill_seg.h"
#include "flank_lookup.h"
#include "glibc_rand.hpp"
#include "post.hpp"
#include "seg_tables.h"

using namespace g2s;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string tl_error;
static int fail(int code, const std::string& msg) { tl_error = msg; return code; }
#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(G2S_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
  } while (0)
// (resident mode: kernels of the list may be in flight on the session's streams and write the caller's pinned buffers
// — an error return waits for them first)
#define HIP_TRY_S(expr)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      (void)hipStreamSynchronize(s->stream);                                                  \
      (void)hipStreamSynchronize(s->stream2);                                                 \
      if (s->stream3) (void)hipStreamSynchronize(s->stream3);                                 \
      return fail(G2S_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    }                                                                                         \
  } while (0)

extern "C" int g2s_abi_version(void) { return G2S_ABI_VERSION; }
extern "C" size_t g2s_backtrace_text(const g2s_gap* gap, const g2s_result* r, int k, char* out, size_t cap) {
  if (!gap || !r || !out || cap == 0 || !(r->flags & G2S_GAP_BACKTRACE_FAIL)) return 0;
  std::string kmer;
  if (r->right_fuz >= 0 && gap->right && (int64_t)r->right_fuz + k <= (int64_t)gap->right_len)
    for (int x = 0; x < k; x++) kmer.push_back((char)toupper((unsigned char)gap->right[r->right_fuz + x]));
  const int len = snprintf(out, cap, "Unable to backtrace! %d %d %s", r->backtrace_depth, r->backtrace_final_d, kmer.c_str());
  return len < 0 ? 0 : std::min<size_t>((size_t)len, cap - 1);
}
extern "C" const char* g2s_last_error(void) { return tl_error.c_str(); }
extern "C" void g2s_free(void* p) { free(p); }
extern "C" void* g2s_host_alloc(size_t bytes) {
  void* p = nullptr;
  // portable: every device of the process may write it (one session per GPU fills its share of one arena)
  if (hipHostMalloc(&p, std::max<size_t>(bytes, 16), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
extern "C" void g2s_host_free(void* p) { if (p) (void)hipHostFree(p); }

extern "C" int g2s_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int d = 0; d < n; d++) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ok++;
  }
  return ok;
}

// ---------------------------------------------------------------------------
// graph
// ---------------------------------------------------------------------------
struct g2s_graph {
  Graph* g = nullptr;
};

extern "C" int g2s_graph_build_seqs(const char* const* seqs, const uint64_t* lens, int nseqs, int k, int solid,
                                    int nthreads, g2s_graph** out) {
  if (!seqs || !out || nseqs < 0) return fail(G2S_ERR_ARG, "g2s_graph_build_seqs: bad argument");
  std::vector<std::pair<const char*, uint64_t>> v;
  for (int i = 0; i < nseqs; i++) v.emplace_back(seqs[i], lens ? lens[i] : (uint64_t)strlen(seqs[i]));
  std::string err;
  Graph* g = graph_build(v, k, solid, nthreads, &err);
  if (!g) return fail(G2S_ERR_ARG, err);
  *out = new g2s_graph();
  (*out)->g = g;
  return G2S_OK;
}

extern "C" int g2s_graph_build_files(const char* reads_csv, int k, int solid, int nthreads, g2s_graph** out) {
  if (!reads_csv || !out) return fail(G2S_ERR_ARG, "g2s_graph_build_files: bad argument");
  std::vector<FastxRecord> recs;
  std::string csv(reads_csv);
  size_t pos = 0;
  while (pos <= csv.size()) {  // comma separated list (Gap2Seq.cpp:199-210)
    size_t e = csv.find(',', pos);
    if (e == std::string::npos) e = csv.size();
    std::string file = csv.substr(pos, e - pos);
    pos = e + 1;
    if (file.empty()) continue;
    std::string text;
    if (!read_text_file(file, &text)) return fail(G2S_ERR_IO, "cannot read " + file);
    parse_fastx(text, &recs);
  }
  std::vector<std::pair<const char*, uint64_t>> v;
  for (auto& r : recs) v.emplace_back(r.seq.data(), (uint64_t)r.seq.size());
  std::string err;
  Graph* g = graph_build(v, k, solid, nthreads, &err);
  if (!g) return fail(G2S_ERR_ARG, err);
  *out = new g2s_graph();
  (*out)->g = g;
  return G2S_OK;
}

extern "C" int g2s_graph_save(const g2s_graph* g, const char* path) {
  std::string err;
  if (!g || !path) return fail(G2S_ERR_ARG, "g2s_graph_save: bad argument");
  return graph_save(*g->g, path, &err) ? G2S_OK : fail(G2S_ERR_IO, err);
}
extern "C" int g2s_graph_load(const char* path, g2s_graph** out) {
  std::string err;
  if (!path || !out) return fail(G2S_ERR_ARG, "g2s_graph_load: bad argument");
  Graph* g = graph_load(path, &err);
  if (!g) return fail(G2S_ERR_IO, err);
  *out = new g2s_graph();
  (*out)->g = g;
  return G2S_OK;
}
extern "C" void g2s_graph_free(g2s_graph* g) {
  if (!g) return;
  if (g->g) {
    for (auto& kv : g->g->dev) {
      if (hipSetDevice(kv.first) == hipSuccess) {
        if (kv.second.succ) (void)hipFree(kv.second.succ);
        if (kv.second.pred) (void)hipFree(kv.second.pred);
        if (kv.second.ustart) (void)hipFree(kv.second.ustart - kUstartPad);
        if (kv.second.rem) (void)hipFree(kv.second.rem);
        if (kv.second.urec) (void)hipFree(kv.second.urec);
      }
    }
    delete g->g;
  }
  delete g;
}
extern "C" int g2s_graph_k(const g2s_graph* g) { return g ? g->g->k : 0; }
extern "C" int g2s_graph_solid(const g2s_graph* g) { return g ? g->g->solid : 0; }
extern "C" uint64_t g2s_graph_num_kmers(const g2s_graph* g) { return g ? g->g->n : 0; }
extern "C" uint64_t g2s_graph_num_unitigs(const g2s_graph* g) { return g ? g->g->n_unitigs : 0; }
extern "C" uint32_t g2s_graph_node(const g2s_graph* g, const char* kmer) {
  if (!g || !kmer || (int)strnlen(kmer, (size_t)g->g->k) < g->g->k) return G2S_INVALID_NODE;
  return g->g->node_of(kmer);
}
extern "C" int g2s_graph_successors(const g2s_graph* g, uint32_t node, uint32_t out[4]) {
  if (!g || (uint64_t)node >= 2 * g->g->n) return 0;
  int c = 0;
  for (int nt = 0; nt < 4; nt++) {
    uint32_t w = g->g->succ_of(node, nt);
    if (w != kInvalidNode) out[c++] = w;
  }
  return c;
}
extern "C" int g2s_graph_predecessors(const g2s_graph* g, uint32_t node, uint32_t out[4]) {
  if (!g || (uint64_t)node >= 2 * g->g->n) return 0;
  int c = 0;
  for (int nt = 0; nt < 4; nt++) {
    uint32_t w = g->g->pred_of(node, nt);
    if (w != kInvalidNode) out[c++] = w;
  }
  return c;
}
extern "C" int g2s_graph_node_string(const g2s_graph* g, uint32_t node, char* out) {
  if (!g || !out || (uint64_t)node >= 2 * g->g->n) return fail(G2S_ERR_ARG, "g2s_graph_node_string: bad node");
  std::string s = g->g->node_string(node);
  memcpy(out, s.c_str(), s.size() + 1);
  return G2S_OK;
}

extern "C" int g2s_graph_upload(g2s_graph* gh, int device) {
  g2s_env_sync();
  if (!gh) return fail(G2S_ERR_ARG, "g2s_graph_upload: null graph");
  Graph& g = *gh->g;
  if (g.dev.count(device)) return G2S_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return fail(G2S_ERR_NO_DEVICE, "no HIP device " + std::to_string(device) + " (this library has no CPU fallback)");
  HIP_TRY(hipSetDevice(device));
  DeviceGraph dg;
  const size_t bytes = g.succ.size() * sizeof(uint32_t);
  // the LDS tier reads whole 512 B blocks: pad the table with INVALID entries
  HIP_TRY(hipMalloc((void**)&dg.succ, bytes + 1024));
  HIP_TRY(hipMemset(dg.succ, 0xFF, bytes + 1024));
  HIP_TRY(hipMemcpy(dg.succ, g.succ.data(), bytes, hipMemcpyHostToDevice));
  dg.bytes = bytes;
  if (!g.pred.empty()) {
    HIP_TRY(hipMalloc((void**)&dg.pred, bytes));
    HIP_TRY(hipMemcpy(dg.pred, g.pred.data(), bytes, hipMemcpyHostToDevice));
    dg.bytes += bytes;
  }
  {  // unitig-start bitmap between two pads of all-ones words (scans past either end stop there)
    const size_t words = g.ustart.size();
    uint64_t* base = nullptr;
    HIP_TRY(hipMalloc((void**)&base, (words + 2 * kUstartPad) * 8));
    HIP_TRY(hipMemset(base, 0xFF, (words + 2 * kUstartPad) * 8));
    HIP_TRY(hipMemcpy(base + kUstartPad, g.ustart.data(), words * 8, hipMemcpyHostToDevice));
    dg.ustart = base + kUstartPad;
    dg.bytes += (words + 2 * kUstartPad) * 8;
  }
  g.dev[device] = dg;
  return G2S_OK;
}
extern "C" uint64_t g2s_graph_device_bytes(const g2s_graph* g, int device) {
  if (!g) return 0;
  auto it = g->g->dev.find(device);
  return it == g->g->dev.end() ? 0 : it->second.bytes;
}

// ---------------------------------------------------------------------------
// session
// ---------------------------------------------------------------------------
namespace {

static inline void cpu_relax() { __builtin_ia32_pause(); }

struct DevBuf {  // grow-only device allocation
  void* p = nullptr;
  size_t cap = 0;
  size_t clean = 0;  // leading bytes known to hold their reset value (set by whoever resets them ahead of time)
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    clean = 0;
    if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { e = hipMalloc(&p, bytes); want = bytes; }
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; clean = 0; }
};
struct PinBuf {  // grow-only pinned host allocation
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    // mapped + coherent: the LDS tier's kernels write their results here directly
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

inline uint32_t pow2ceil(uint64_t x) {
  uint64_t p = 1;
  while (p < x) p <<= 1;
  return (uint32_t)std::min<uint64_t>(p, 1u << 30);
}

}  // namespace

// diagnostics: G2S_PROGRESS_FILE=path appends one line per kernel launch / completion (opened and closed per
// line, so that the file is complete even when the process has to be abandoned)
static void progress_note(const char* fmt, ...) {
  const char* path = GENV("G2S_PROGRESS_FILE");
  if (!path) return;
  if (FILE* f = fopen(path, "a")) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(f, fmt, ap);
    va_end(ap);
    fputc('\n', f);
    fclose(f);
  }
}

namespace {
struct TierData {  // what came back from one launch group (pinned buffers live in the session)
  PinBuf outs, subs;
  PinBuf done;  // LDS tier: gap indices in completion order, written by the kernel as gaps finish
  std::vector<SubRec> conv;        // HBM tier: closures converted to the host's 16-byte records
  std::vector<uint64_t> conv_xp;
  std::vector<uint32_t> gap_ids;
  // segment tier: the closures arrive as segments and are expanded into per-state records here,
  // each gap taking its share with one atomic add (analysis threads work on different gaps)
  std::vector<SubRec> exp;
  std::atomic<size_t> exp_cursor{0};
};
}  // namespace

// CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota
// (cpu.max) when there is one — a 256-thread host often hands a container 16 CPUs' worth.
static int usable_cpus() {
  int n = (int)std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, CPU_COUNT(&set));
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32];
    long long period = 0;
    if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
      n = std::min(n, (int)std::max(1ll, (atoll(q) + period - 1) / period));
    fclose(f);
  } else {
    long long quota = -1, period = 0;
    if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = -1; fclose(fq); }
    if (FILE* fp2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp2, "%lld", &period) != 1) period = 0; fclose(fp2); }
    if (quota > 0 && period > 0) n = std::min(n, (int)std::max(1ll, (quota + period - 1) / period));
  }
  return std::max(1, n);
}

// CPUs of the NUMA node the calling thread runs on, intersected with its affinity mask.
static bool local_node_cpus(cpu_set_t* out) {
  const int cpu = sched_getcpu();
  cpu_set_t allowed;
  if (cpu < 0 || sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
  for (int node = 0; node < 64; node++) {
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return false;  // no more nodes: not found
    char buf[4096];
    const bool got = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!got) continue;
    cpu_set_t set;
    CPU_ZERO(&set);
    bool mine = false;
    for (char* tok = strtok(buf, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
      int a = 0, b = 0;
      if (sscanf(tok, "%d-%d", &a, &b) == 2) {}
      else if (sscanf(tok, "%d", &a) == 1) b = a;
      else continue;
      for (int c = a; c <= b && c < CPU_SETSIZE; c++) {
        if (CPU_ISSET(c, &allowed)) CPU_SET(c, &set);
        if (c == cpu) mine = true;
      }
    }
    if (mine) {
      if (CPU_COUNT(&set) < 2) return false;
      *out = set;
      return true;
    }
  }
  return false;
}

// Persistent host workers for the per-gap post-processing (spawning threads per
// batch costs more than the work at 500 gaps per batch).
class WorkerPool {
 public:
  explicit WorkerPool(int nthreads) {
    for (int t = 0; t < nthreads; t++) th_.emplace_back([this]() { loop(); });
    // keep the workers on the NUMA node of the thread that owns the session: the closures
    // they read were copied into pinned memory of that node, and what they write is read
    // back by that thread
    cpu_set_t node;
    if (!GENV("G2S_NO_NUMA_BIND") && local_node_cpus(&node))
      for (auto& t : th_) pthread_setaffinity_np(t.native_handle(), sizeof node, &node);
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; gen_.fetch_add(1); }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  int size() const { return (int)th_.size(); }
  // f(i) for i in [0,n); the calling thread helps.  Completion is counted in tasks, not in
  // workers: a worker that wakes up late finds nothing to do and nobody waits for it.
  void run(size_t n, const std::function<void(size_t)>& f) {
    post(n, f);
    finish();
  }
  // post: hand the tasks to the workers and return (f must stay alive until finish());
  // finish: help with what is left and wait for the last task.  One job at a time.
  void post(size_t n, const std::function<void(size_t)>& f) {
    posted_n_ = n;
    posted_f_ = &f;
    if (n == 0) return;
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &f; n_ = n; done_.store(0);
      posted_g_ = gen_.load() + 1;
      next_.store(posted_g_ << 32);  // the generation tags every ticket: stale workers cannot take one
      gen_.store(posted_g_);
    }
    // wake no more workers than there are tasks besides the caller's share (every wake-up of a
    // sleeping thread costs microseconds of CPU on a host that is usually shared)
    if (n > th_.size()) cv_.notify_all();
    else for (size_t i = 0; i + 1 < n; i++) cv_.notify_one();
  }
  // nothing posted, or every task of the posted job done (finish() will not wait)
  bool idle() const { return posted_n_ == 0 || done_.load(std::memory_order_acquire) == posted_n_; }
  void finish() {
    if (posted_n_ == 0) return;
    drain(posted_g_, posted_n_, posted_f_);
    // tail of the last tasks (a few microseconds): the only place a thread waits without sleeping
    while (done_.load(std::memory_order_acquire) != posted_n_) __builtin_ia32_pause();
    posted_n_ = 0;
  }
 private:
  void drain(uint64_t g, size_t n, const std::function<void(size_t)>* f) {
    size_t mine = 0;
    uint64_t t = next_.load(std::memory_order_acquire);
    while (true) {
      // a ticket is only taken when it belongs to this generation (a plain fetch_add by a
      // late worker would swallow a ticket of the next run)
      if ((t >> 32) != (g & 0xFFFFFFFFull) || (t & 0xFFFFFFFFull) >= n) break;
      if (!next_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel)) continue;
      (*f)((size_t)(t & 0xFFFFFFFFull));
      mine++;
      t = next_.load(std::memory_order_acquire);
    }
    if (mine) done_.fetch_add(mine, std::memory_order_acq_rel);
  }
  void loop() {
    uint64_t seen = 0;
    while (true) {
      uint64_t g;
      size_t n;
      const std::function<void(size_t)>* f;
      // a short look for the next job before going to sleep: the jobs of one list follow each other within
      // microseconds (sizes, flank text, host-finished gaps), and waking a sleeping thread costs tens of them
      for (int spin = 0; spin < 1500 && gen_.load(std::memory_order_acquire) == seen; spin++) __builtin_ia32_pause();
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&]() { return gen_.load() != seen; });
        if (stop_) return;
        g = gen_.load(); n = n_; f = fn_;
      }
      seen = g;
      drain(g, n, f);
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_;
  const std::function<void(size_t)>* fn_ = nullptr;
  size_t n_ = 0;
  std::atomic<uint64_t> next_{0};
  std::atomic<size_t> done_{0};
  std::atomic<uint64_t> gen_{0};
  bool stop_ = false;
  size_t posted_n_ = 0;
  uint64_t posted_g_ = 0;
  const std::function<void(size_t)>* posted_f_ = nullptr;
};

// The session's rand() stream, materialised ahead of use so that tracebacks of
// different gaps can read their draws at known offsets in parallel (value = word >> 1).
struct RandCache {
  GlibcRandStream st;
  // the generator's state and the next values as one window of the flat stream (resident mode hands it to the device)
  const uint32_t* window(size_t words) { return st.window(words); }
  void jump(size_t n, const uint32_t state31[31]) { st.jump(n, state31); }
  void seed(uint32_t s) { st.seed(s); }
  void ensure(size_t n) { st.ensure(n); }  // at least n upcoming values materialised
  int32_t at(size_t off) { st.ensure(off + 1); return st.value(off); }
  const uint32_t* ptr(size_t off) const { return st.raw() + off; }
  int32_t at_const(size_t off) const { return st.value(off); }  // already materialised
  bool would_grow(size_t n) const { return st.would_grow(n); }  // ensure(n) would move the buffer
  void consume(size_t n) { st.consume(n); }
  void skip(uint64_t n) { st.skip(n); }  // n values drawn elsewhere, the state behind them not handed over
  void swap(RandCache& o) { st.swap(o.st); }
};

// One helper thread per session for the work that runs beside a batch (materialising rand() values):
// started once, handed a job per batch — creating a thread per call cost ~20 us of a 500 us step.
struct BgWorker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, busy = false, quit = false;
  void loop() {
    std::unique_lock<std::mutex> lk(mu);
    while (true) {
      cv.wait(lk, [&] { return has_job || quit; });
      if (quit) return;
      std::function<void()> j = std::move(job);
      has_job = false;
      lk.unlock();
      j();
      lk.lock();
      busy = false;
      cv.notify_all();
    }
  }
  void submit(std::function<void()> j) {
    std::unique_lock<std::mutex> lk(mu);
    if (!th.joinable()) th = std::thread([this] { loop(); });
    cv.wait(lk, [&] { return !busy; });
    job = std::move(j);
    has_job = true;
    busy = true;
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return !busy; });
  }
  ~BgWorker() {
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return !busy; });
      quit = true;
      cv.notify_all();
    }
    if (th.joinable()) th.join();
  }
};

struct g2s_session {
  g2s_graph* graph = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;  // resident mode: the rand() stream is generated beside the fill kernel
  hipStream_t stream3 = nullptr;  // resident mode, deep lists: the large variant's early launch (resident_launch_fill)
  hipEvent_t ev_desc = nullptr;  // the fill kernel's descriptors and launch order are in device memory (copied on the third stream)
  hipEvent_t ev_pre = nullptr, ev_early = nullptr;  // in front of the fill kernel; behind the early launch
  hipEvent_t ev_fill = nullptr, ev_d2 = nullptr;    // behind the fill kernels (what g2s_d2_* waits for on its stream); behind g2s_d2_*
  DevBuf d_segx1;                 // the early launch's scratch
  hipEvent_t ev_rand = nullptr;
  g2s_params params;
  RandCache rcache;
  BgWorker bg;
  WorkerPool* pool = nullptr;
  WorkerPool* team_pool = nullptr;  // (a lead session's) one thread per other session of a team
  bool no_lds_tier = false;  // G2S_NO_LDS_TIER=1: force the general HBM tier (tests, A/B timing)
  size_t mem_budget = 0;  // bytes of HBM this session may use for work areas
  DevBuf d_gaps, d_ids, d_flank, d_outs, d_rs, d_rlog, d_keys, d_cnt, d_mark, d_slog, d_subscr, d_subout, d_counter;
  DevBuf d_xcd;  // segment tier, batched announcements: 8 ticket counters (64 bytes), then 8 lists of finished gaps
  // the two largest per-gap arrays of a batch, kept from one batch to the next: allocated afresh they are new
  // pages every time (3 MB per 10 000 gaps: 0.2-0.3 ms of page faults per call)
  std::vector<SubView> spare_views;
  std::vector<GapJob> spare_jobs;            // per-gap arrays of the last batch that was freed (taken by the next one)
  std::vector<uint32_t> spare_u32[3];
  std::vector<size_t> spare_sz;
  std::vector<SubPrep> spare_prep;
  DevBuf d_log, d_lvl, d_plk, d_xl, d_xo;  // LDS tier: state log, level offsets, parent links, closure side lists
  DevBuf d_segx;  // large variant of the segment tier: segment arrays and queues of its persistent workgroups
  DevBuf d_ovf;   // resident mode: the gaps of a launch that outgrew the regular tier (the large variant's list)
  // resident mode, phase D2 on the device for the closures the fill kernels leave (d2_device.hip): the gaps they list
  // (two lists of n: the small instantiation's, then what it passes on), the per-gap results, the runs, the kernels'
  // scratch; d_hops: where the trace kernel keeps the segments a traceback enters when the closure is too large for
  // its LDS (a traceback enters a segment once: as many entries as the closure has segments, at the closure's offset)
  DevBuf d_d2list, d_d2out, d_d2runs, d_d2scr_small, d_d2scr_big, d_hops;
  bool spec_on = false;      // the last fill launch's waves left guessed tracebacks in d_textout / d_resout (G2S_DEVA_SPEC)
  bool d2_launched = false;  // the last fill launch had g2s_d2_* behind it
  bool d2_wait = false;      // ... and phase D3's hand-off waits for it (deep lists: their trace waves would keep its
                             // large instantiation off the compute units); else the trace waves of its gaps do
  uint32_t d2_wgs = 0;       // workgroups of g2s_d2_* behind the last fill launch
  uint32_t d2_lists = 0;     // lists whose fill kernels tagged their entries for a polling g2s_d2_small (the tag's low bits)
  unsigned long long d2_done_total = 0;  // gaps those fill launches counted as through, all lists (the counters only grow)
  const void* d2_ctr_seen = nullptr;     // the counter buffer those counts live in
  size_t d2_ticks_n = 0;     // (G2S_D2_LOG) gaps of the last list whose listing times are in d_d2list
  bool d2_prof_on = false;   // (G2S_D2_PROF) the section counters behind the cursors have been zeroed
  int num_cus = 256;
  DevBuf d_rspool;
  DevBuf d_logpool;  // chunks for state logs that outgrow their slice of d_log (LDS tier)                          // LDS tier: spill pool for right sets
  std::vector<void*> tier_pool;  // recycled TierData (pinned host buffers)
  size_t tier_cursor = 0;        // next free slot of tier_pool in the current run
  const void* flank_owner = nullptr;  // batch whose flank nodes d_flank holds
  const void* desc_owner = nullptr;   // batch whose resident-mode descriptors h_gaps / h_d3 hold
  std::vector<g2s_session*> helpers;  // g2s_session_set_team
  size_t team_group = 0;
  PinBuf h_gaps;                 // staging for the GapDev upload
  std::vector<PinBuf*> pin_free; // pinned buffers of finished batches (flank text in, node ids out), reused
  FlankLookup lookup;            // device copy of the sorted k-mer set for the flank look-up kernel
  DevBuf d_lk_kmers, d_lk_bucket, d_lk_rank2id, d_lk_flip;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  g2s_timing last_timing;        // of the last g2s_fill_batch / g2s_batch_run (g2s_session_last_timing)
  // resident mode (run_resident): closures, phase D3 work areas and the rand() stream stay on the device
  DevBuf d_d2log, d_sub, d_d3, d_rnd, d_lastch, d_rtab, d_resout, d_textout, d_dgap;
  DevBuf d_fstage;  // a long list's look-up descriptors and flank text, copied in front of the look-up kernel
  DevBuf d_outs_all, d_sub_all;  // lead of a team: the groups' records and closure records, gathered for phase D3
  PinBuf h_d3all;                // and the list's D3Gap array, summary and stream window
  PinBuf h_gfn;                  // a group of a sharded list: its group function (deviation behind it by deviation in front)
  bool team_sharded = false;     // this session's group of a team's list stays on its device through phase D3
  PinBuf h_d3;                   // D3Gap per gap | summary | stream window; staging of results / text when the caller's are not pinned
  PinBuf h_res, h_text, h_side;
  // resident mode, deep lists: the closures the host analyses arrive while the large variant still runs (SegEarly):
  // pinned items | GapOut copies | segments, the device's two counters, and what the analysis of an item leaves for
  // its traceback behind phase D3's hand-over
  PinBuf h_early;
  DevBuf d_early_ctr;
  SegEarly early_host;             // host pointers into h_early (cap_items 0: not in use for the list in flight)
  std::vector<SubPrep> early_prep;
  std::vector<std::vector<uint64_t>> early_scratch;
  std::vector<int32_t> early_of_gap;  // by gap: its early item, or -1
  std::vector<std::pair<uint32_t, uint32_t>> early_inline_done;  // (lists that are not deep) the (gap, item) pairs the waiting thread analysed
  RandTables rtab;
  std::vector<uint32_t> res_ids, res_at;  // launch order of a resident list and its counting sort, kept between lists
  bool in_team_list = false;     // the session is filling a group of a team's list (team_resident)
  bool team_shares_device = false;  // ... and another session of the team sits on the same device
  int team_sessions = 1;            // ... sessions of that team (they share the host's threads)
  int peer_asked_for = -1;       // the lead device this session asked direct access to (team lists)
  bool self_cleaned = false;     // the last list's trace kernel zeroed records, summary and cursors behind itself
  std::vector<uint32_t> host_order;  // resident mode: the handed-over items, largest closure first
  size_t side_dirty = SIZE_MAX;  // items of h_side whose ready word may be set (resident mode's hand-over)...
  size_t side_layout_n = 0;      // ...under the layout of a list of this many gaps (the arrays behind the items move with it)
  uint32_t timed_seq = 0;        // resident launches so far (one in eight is bracketed with HIP events)
  // g2s_timing.host_us: time points of the last resident list on this session (g2s_fill_batch computes the laps)
  std::chrono::steady_clock::time_point lap_fill_queued, lap_d3_queued, lap_handed, lap_finished, lap_synced, lap_end;
  bool laps_valid = false;
  int resident_strikes = 0;      // lists that had to be run again on the host path; three in a row switch the mode off
  // The large variant behind the fill kernel of a resident launch (for the gaps that outgrow the regular tier) is an
  // empty launch on most lists, and an empty launch costs a 500-gap list 5 us of its 210: after eight lists in a row
  // without such a gap it is left out, until a list has one (that list takes the host path once) or is deep by its
  // parameters.
  int segw_quiet = 0;
  hipEvent_t ev_segw = nullptr;  // behind the large variant's launch (timed launches)
  // g2s_fill_begin / g2s_fill_end: up to G2S_MAX_IN_FLIGHT lists in flight, each on this session or a twin of it on the
  // same device (own streams and buffers; created on first use).  The rand() stream is this session's: a list that
  // ends on a twin borrows it.
  g2s_session* twins[G2S_MAX_IN_FLIGHT - 1] = {};
  // (lists in flight) the generator's state behind the last draw of the list whose phase D3 ran here last, in device
  // memory (D3Work.link), and the event behind the kernel that writes it: the next list's stream starts from it
  // without the host in between
  DevBuf d_link;
  hipEvent_t ev_chain = nullptr;
  void* d3_pending = nullptr;  // (D3Pending) phase D3 of a list queued on this session's stream and not waited for yet
  struct InFlight { g2s_batch* b = nullptr; g2s_session* on = nullptr; g2s_result* results = nullptr; char* arena = nullptr; size_t cap = 0;
                    const g2s_gap* gaps = nullptr; size_t n = 0; double ms_prepare = 0;
                    bool d3_queued = false;
                    bool chained = false; };  // its rand() stream continues the older list's on the device
  InFlight inflight[G2S_MAX_IN_FLIGHT];
  int n_inflight = 0;
  // g2s_fill_begin / g2s_fill_end calling the other entry points for a list of their own: those refuse callers
  // while lists are in flight (the lists' kernels still read and write this session's buffers)
  int internal_calls = 0;
  uint64_t begun = 0;
  // g2s_share_*: this rank's share of a list sharded over processes, between g2s_share_begin and g2s_share_end
  g2s_batch* share_batch = nullptr;
  int share_step = 0;  // 1 begun, 2 tables, 3 traced
  bool share_timed = false, share_two = false;
  uint32_t share_win[G2S_RAND_WINDOW];
  bool resident_off = false;
};

static void d3_pending_drop(g2s_session* s);  // (D3Pending is defined with phase D3's launch code below)
namespace {
// Between g2s_fill_begin and the matching g2s_fill_end the session's pinned and device buffers belong to the lists in
// flight: every other entry point that would queue work on the session, or move its rand() stream, refuses.
inline bool lists_in_flight(const g2s_session* s) { return s && s->n_inflight > 0 && s->internal_calls == 0; }
struct InternalCall {
  g2s_session* s;
  explicit InternalCall(g2s_session* s_) : s(s_) { s->internal_calls++; }
  ~InternalCall() { s->internal_calls--; }
};
#define REFUSE_IN_FLIGHT(s, who) \
  do { if (lists_in_flight(s)) return fail(G2S_ERR_STATE, who ": lists are in flight on this session (g2s_fill_end them first)"); } while (0)
}  // namespace

extern "C" int g2s_session_create(g2s_graph* g, int device, const g2s_params* p, g2s_session** out) {
  g2s_env_sync();
  if (!g || !p || !out) return fail(G2S_ERR_ARG, "g2s_session_create: bad argument");
  int rc = g2s_graph_upload(g, device);
  if (rc != G2S_OK) return rc;
  HIP_TRY(hipSetDevice(device));
  {  // the segment tier's table, derived on the device from the bitmap that is already there
    DeviceGraph& dg = g->g->dev.at(device);
    if (!dg.rem && !dg.pred && g->g->n > 0) {
      HIP_TRY(build_rem_table(dg.ustart, g->g->n, &dg.rem));
      HIP_TRY(build_urec_table(dg.succ, dg.rem, g->g->n, &dg.urec));
      dg.bytes += g->g->n * (8 + 64);
    }
  }
  g2s_session* s = new g2s_session();
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) s->num_cus = cus;
  }
  {  // what the flank look-up kernel searches: the sorted k-mer set, its prefix index, rank -> node
    const Graph& gr = *g->g;
    const size_t kb = gr.wide ? gr.kmers128.size() * 16 : gr.kmers64.size() * 8;
    hipError_t e = s->d_lk_kmers.ensure(std::max<size_t>(kb, 16));
    if (e == hipSuccess && kb) e = hipMemcpy(s->d_lk_kmers.p, gr.wide ? (const void*)gr.kmers128.data() : (const void*)gr.kmers64.data(), kb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = s->d_lk_bucket.ensure(std::max<size_t>(gr.bucket.size() * 4, 16));
    if (e == hipSuccess && !gr.bucket.empty()) e = hipMemcpy(s->d_lk_bucket.p, gr.bucket.data(), gr.bucket.size() * 4, hipMemcpyHostToDevice);
    // (rank -> the oriented node of the canonical k-mer, one word: 2 * id | flip — one random access of the look-up
    // instead of two)
    if (e == hipSuccess) e = s->d_lk_rank2id.ensure(std::max<size_t>(gr.rank2id.size() * 4, 16));
    if (e == hipSuccess && !gr.rank2id.empty()) {
      std::vector<uint32_t> r2n(gr.rank2id.size());
      for (size_t q = 0; q < r2n.size(); q++) r2n[q] = 2u * gr.rank2id[q] | (uint32_t)(gr.flip[q] & 1u);
      e = hipMemcpy(s->d_lk_rank2id.p, r2n.data(), r2n.size() * 4, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) { delete s; return fail(G2S_ERR_HIP, std::string("session setup (look-up tables): ") + hipGetErrorString(e)); }
    s->lookup.kmers = s->d_lk_kmers.p; s->lookup.bucket = (const uint32_t*)s->d_lk_bucket.p;
    s->lookup.rank2node = (const uint32_t*)s->d_lk_rank2id.p;
    s->lookup.k = gr.k; s->lookup.bucket_bits = gr.bucket_bits; s->lookup.wide = gr.wide ? 1 : 0;
  }
  {  // phase D3 on the device (d3_device.hip): the last base of every oriented k-mer as text, the generator's jump tables
    const Graph& gr = *g->g;
    gr.ensure_lastch();
    hipError_t e = s->d_lastch.ensure(std::max<size_t>(2 * (size_t)gr.n, 16));
    if (e == hipSuccess && gr.n) e = hipMemcpy(s->d_lastch.p, gr.lastch_up.data(), (size_t)gr.n, hipMemcpyHostToDevice);
    if (e == hipSuccess && gr.n) e = hipMemcpy((char*)s->d_lastch.p + gr.n, gr.lastch_dn.data(), (size_t)gr.n, hipMemcpyHostToDevice);
    static std::vector<uint32_t> tables;  // seed independent: once per process
    static std::once_flag tables_once;
    std::call_once(tables_once, []() { tables.resize((128 + 256 + 64) * 31); rand_tables_host(tables.data(), tables.data() + 128 * 31, tables.data() + (128 + 256) * 31); });
    if (e == hipSuccess) e = s->d_rtab.ensure(tables.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(s->d_rtab.p, tables.data(), tables.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete s; return fail(G2S_ERR_HIP, std::string("session setup (phase D3 tables): ") + hipGetErrorString(e)); }
    s->rtab.hi = (const uint32_t*)s->d_rtab.p;
    s->rtab.mid = s->rtab.hi + 128 * 31;
    s->rtab.lane = s->rtab.mid + 256 * 31;
    if (const char* env = GENV("G2S_RESIDENT")) s->resident_off = atoi(env) == 0;
  }
  s->graph = g;
  s->device = device;
  s->params = *p;
  memset(&s->last_timing, 0, sizeof s->last_timing);
  // srand((randseed > 0) ? randseed : time(NULL)) (Gap2Seq.cpp:178); params keep the user's value for the echo (:191)
  s->rcache.seed(p->randseed > 0 ? p->randseed : (uint32_t)time(nullptr));
  if (const char* env = GENV("G2S_NO_LDS_TIER")) s->no_lds_tier = atoi(env) != 0;
  hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_rand, hipEventDisableTiming);
  for (int i = 0; i < 5 && e == hipSuccess; i++) e = hipEventCreate(&s->ev[i]);
  if (e == hipSuccess) e = hipEventCreate(&s->ev_segw);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_chain, hipEventDisableTiming);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream3, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_pre, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_desc, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_early, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_fill, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_d2, hipEventDisableTiming);
  if (e == hipSuccess) e = s->d_link.ensure(32 * 4);
  size_t free_b = 0, total_b = 0;
  if (e == hipSuccess) e = hipMemGetInfo(&free_b, &total_b);
  if (e != hipSuccess) { delete s; return fail(G2S_ERR_HIP, std::string("session setup: ") + hipGetErrorString(e)); }
  s->mem_budget = (size_t)((double)free_b * 0.6);
  {
    int nth = p->host_threads > 0 ? p->host_threads : usable_cpus();
    // default cap 32: beyond that the wake-up of the pool costs more than it saves at 500 gaps
    // per batch; an explicit host_threads is honoured up to 256
    nth = std::max(1, std::min(nth, p->host_threads > 0 ? 256 : 32));
    if (p->host_threads <= 0) nth = std::max(nth, std::min(32, 2 * nth));  // short bursts: a CPU quota is an average
    s->pool = new WorkerPool(nth - 1);
  }
  *out = s;
  return G2S_OK;
}

extern "C" void g2s_session_destroy(g2s_session* s) {
  if (!s) return;
  if (s->d2_prof_on && s->d_counter.p) {  // (G2S_D2_PROF: what g2s_d2_* spent where, over the session's lists)
    unsigned long long both[64] = {0};
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    if (hipMemcpy(both, (char*)s->d_counter.p + 1024, 512, hipMemcpyDeviceToHost) == hipSuccess)
      for (int k = 0; k < 2; k++) {  // ([0 .. 31]: g2s_d2_small, [32 .. 63]: g2s_d2_big; s_memrealtime ticks of 10 ns)
        const unsigned long long* pr = both + 32 * k;
        static const char* names[11] = {"load+sort", "dag", "chain intervals", "cuts", "runs", "edges", "csr", "components", "statistics", "order", "verdicts"};
        fprintf(stderr, "[g2s] %s: %llu closures taken, %llu through on runs (%.0f nodes, %.0f rounds each), %llu as a DAG; ticks per closure taken %.0f, the longest %llu (%llu records); by section, summed:",
                k ? "g2s_d2_big" : "g2s_d2_small", pr[17], pr[11], pr[11] ? (double)pr[12] / (double)pr[11] : 0.0,
                pr[11] ? (double)pr[13] / (double)pr[11] : 0.0, pr[19], pr[17] ? (double)pr[18] / (double)pr[17] : 0.0, pr[16], pr[20]);
        for (int q = 0; q < 11; q++) fprintf(stderr, " %s %llu", names[q], pr[q]);
        fprintf(stderr, " | beyond the capacities %llu, given up %llu\n", pr[14], pr[15]);
      }
    if (const char* path = GENV("G2S_D2_LOG"))  // (one line per closure taken: tools/d2_log.py)
      if (s->d_d2log.p) {
        std::vector<unsigned long long> lg(16u * 8192u);
        if (hipMemcpy(lg.data(), s->d_d2log.p, lg.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
          if (FILE* f = fopen(path, "w")) {
            for (int k = 0; k < 2; k++)
              for (unsigned long long q = 0; q < std::min<unsigned long long>(both[32 * k + 21], 4096ull); q++) {
                const unsigned long long* e = lg.data() + 16ull * (4096ull * k + q);
                for (int w = 0; w < 16; w++) fprintf(f, "%llu%c", e[w], w == 15 ? '\n' : ' ');
              }
            std::vector<uint32_t> tk(s->d2_ticks_n);  // (the last list's: "T gap tick" — s_memrealtime, 10 ns)
            if (s->d2_ticks_n && s->d_d2list.p && hipMemcpy(tk.data(), (uint32_t*)s->d_d2list.p + 2 * s->d2_ticks_n, tk.size() * 4, hipMemcpyDeviceToHost) == hipSuccess)
              for (size_t q = 0; q < tk.size(); q++) if (tk[q]) fprintf(f, "T %zu %u\n", q, tk[q]);
            fclose(f);
          }
      }
  }
  // lists begun and never ended: their kernels first — on all three streams of every session that carries one (the
  // rand() stream and the descriptors run on the second, the large variant's early launch on the third) — and what
  // was queued for their phase D3 (it points into the batches freed below)
  for (int q = 0; q < s->n_inflight; q++)
    if (g2s_session* on = s->inflight[q].on) {
      (void)hipSetDevice(on->device);
      for (hipStream_t st : {on->stream, on->stream2, on->stream3}) if (st) (void)hipStreamSynchronize(st);
    }
  d3_pending_drop(s);
  for (g2s_session* t : s->twins) if (t) d3_pending_drop(t);
  for (int q = 0; q < s->n_inflight; q++) g2s_batch_free(s->inflight[q].b);
  s->n_inflight = 0;
  for (g2s_session*& t : s->twins) if (t) { g2s_session_destroy(t); t = nullptr; }
  (void)hipSetDevice(s->device);
  for (hipStream_t st : {s->stream, s->stream2, s->stream3}) if (st) (void)hipStreamSynchronize(st);
  DevBuf* bufs[] = {&s->d_gaps, &s->d_ids, &s->d_flank, &s->d_outs, &s->d_rs, &s->d_rlog, &s->d_keys,
                    &s->d_cnt, &s->d_mark, &s->d_slog, &s->d_subscr, &s->d_subout, &s->d_counter, &s->d_xcd,
                    &s->d_log, &s->d_lvl, &s->d_plk, &s->d_xl, &s->d_xo, &s->d_rspool, &s->d_logpool, &s->d_segx, &s->d_ovf};
  for (DevBuf* b : bufs) b->release();
  delete s->pool;
  delete s->team_pool;
  for (PinBuf* pb : s->pin_free) { pb->release(); delete pb; }
  s->d_lk_kmers.release(); s->d_lk_bucket.release(); s->d_lk_rank2id.release(); s->d_lk_flip.release();
  for (void* v : s->tier_pool) { TierData* t = (TierData*)v; t->outs.release(); t->subs.release(); t->done.release(); delete t; }
  s->h_gaps.release();
  for (int i = 0; i < 5; i++) if (s->ev[i]) (void)hipEventDestroy(s->ev[i]);
  s->d_outs_all.release(); s->d_sub_all.release(); s->h_d3all.release();
  s->d_resout.release(); s->d_textout.release(); s->d_dgap.release(); s->d_d2log.release(); s->d_sub.release(); s->d_d3.release(); s->d_rnd.release(); s->d_lastch.release(); s->d_rtab.release();
  s->h_d3.release(); s->h_res.release(); s->h_text.release(); s->h_side.release(); s->h_gfn.release();
  s->h_early.release(); s->d_early_ctr.release(); s->d_fstage.release();
  if (s->ev_rand) (void)hipEventDestroy(s->ev_rand);
  if (s->ev_segw) (void)hipEventDestroy(s->ev_segw);
  if (s->ev_chain) (void)hipEventDestroy(s->ev_chain);
  if (s->ev_pre) (void)hipEventDestroy(s->ev_pre);
  if (s->ev_desc) (void)hipEventDestroy(s->ev_desc);
  if (s->ev_early) (void)hipEventDestroy(s->ev_early);
  if (s->ev_fill) (void)hipEventDestroy(s->ev_fill);
  if (s->ev_d2) (void)hipEventDestroy(s->ev_d2);
  if (s->stream3) (void)hipStreamDestroy(s->stream3);
  s->d_segx1.release();
  s->d_d2list.release(); s->d_d2out.release(); s->d_d2runs.release(); s->d_d2scr_small.release(); s->d_d2scr_big.release(); s->d_hops.release();
  s->d_link.release();
  if (s->stream2) (void)hipStreamDestroy(s->stream2);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

extern "C" int g2s_session_srand(g2s_session* s, uint32_t seed) {
  if (!s) return fail(G2S_ERR_ARG, "g2s_session_srand: bad argument");
  // (a list in flight that continues an older one's stream on the device would not see the new seed)
  REFUSE_IN_FLIGHT(s, "g2s_session_srand");
  s->rcache.seed(seed);
  return G2S_OK;
}
extern "C" int g2s_session_skip_draws(g2s_session* s, uint64_t n) {
  if (!s) return fail(G2S_ERR_ARG, "g2s_session_skip_draws: bad argument");
  REFUSE_IN_FLIGHT(s, "g2s_session_skip_draws");
  while (n) {  // in pieces: the stream is materialised before it is consumed
    const size_t c = (size_t)std::min<uint64_t>(n, 1u << 20);
    s->rcache.ensure(c);
    s->rcache.consume(c);
    n -= c;
  }
  return G2S_OK;
}

// ---------------------------------------------------------------------------
// batch
// ---------------------------------------------------------------------------

struct g2s_batch {
  g2s_session* s = nullptr;
  std::vector<GapJob> jobs;
  std::vector<uint32_t> flank_off;
  // pinned, from the session's pool: [FlankDesc per valid gap][flank text][node ids of every gap]
  PinBuf* pin = nullptr;
  FlankDesc* desc = nullptr;
  char* text = nullptr;
  uint32_t* nodes = nullptr;   // oriented flank nodes of every gap (written by the look-up kernel)
  size_t n_desc = 0, n_nodes = 0;
  bool host_lookup = false;    // flanks too long for the kernel's staging buffer: resolved on the host
  // what resident mode needs of the list, gathered by the preparation's own passes
  bool has_skip = false, seg_tier_all = true;  // a gap carries a skip rule; every gap with flanks fits the segment tier
  int gmax = 0, dmax = 0;                       // longest gap, deepest search
  size_t rnd_cap = 0, n_valid = 0;              // rand() values the list can draw at most; gaps with complete flanks
  bool fast_desc = false;                       // GapDev / D3Gap of every gap are in the session's pinned buffers (desc_owner)
  // g2s_fill_begin: the fill kernel of this list has been queued already (resident_launch_fill); g2s_batch_run goes on
  // with phase D3
  bool pre_launched = false, pre_two = false, pre_timed = false, pre_segw = false;
  bool d3_queued = false;  // ...and its phase D3 too (resident_queue_d3): g2s_batch_run only waits
  bool chain_broken = false;  // ...from the device state of a list in front of it which then did not end on the device
  bool others_in_flight = false;  // begun while another list was in flight: the device is shared (resident_launch_fill)
  bool through_begin = false;     // handed over by g2s_fill_begin: other lists' kernels will run beside this one's
  uint64_t pre_units = 0;
  // Flank look-ups: by the look-up kernel in front of the fill kernels (upload_flanks queues it), or — allow_inline, a
  // list resident mode is about to launch in the regular segment tier — by the fill kernel's own waves (fill_seg.hip):
  // then only the text is staged and inline_pending says that d_flank does not hold the ids yet.  Whoever needs the ids
  // in d_flank without that launch (the host path taking over) calls upload_flanks() again: the kernel after all.
  bool inline_ok = false, inline_pending = false;
  const char* inline_text_dev = nullptr;  // device-readable flank text of the list (pinned memory, or its copy in d_fstage)
  uint32_t text_stride = 0;               // not 0: gap i's text at i * text_stride (a list without a bad flank)
  // the fill kernel resolved the flanks and wrote the pinned copy of the node ids only for the gaps resident mode hands
  // to the host: whoever reads `nodes` of other gaps — the host path, when the list is given back — fetches the table first
  bool nodes_dev_only = false;
  int fetch_nodes();
  int upload_flanks(bool allow_inline = false);
  size_t arena_bytes = 0;
  std::vector<size_t> arena_off;  // of each gap's fill buffer within the batch's share of the arena
  char* arena = nullptr;          // the batch's share of the caller's fill arena (stage 1 writes the
  size_t arena_base = 0;          // fills that do not depend on rand() values); base = its offset there
  g2s_timing timing;
  std::vector<TierData*> tiers;
  TierData* seg_td = nullptr;  // the segment tier's launch of this run (closures expanded into its buffer)
  bool force_host_d2 = false;  // G2S_HOST_D2=1 (tests): phase D2 on the host even where the device did it
  // stage 1 results (GPU passes + per-gap analysis), consumed by stage 2 (offsets + tracebacks)
  std::vector<SubView> views;
  std::vector<SubPrep> prep;
  std::vector<char> mem_exceeded;
  // per gap, written by the parallel analysis and read by the in-order offset pass (16 B per
  // gap, contiguous: the sequential part of a batch touches nothing else)
  struct GapInfo {
    int32_t fixed[2];   // draws when pathLengths[pick] is chosen, or -1 when that depends on the draws
    int16_t n_len;      // 0: no phase D
    int16_t reached_j;
    uint8_t kind;       // 0 normal, 1 bad flank, 2 -max-mem verdict
    uint8_t filled;     // bit 0: count > 0 (&& == 1 with -unique); bit 1: fill already written by the analysis
    int16_t skip_thr;   // skip_if_prev_right_fuz_gt clamped to int16 (-1: never skip)
  };
  std::vector<GapInfo> info;
  void drop_tiers();
  ~g2s_batch() {
    if (s) {
      drop_tiers();
      if (s->flank_owner == this) s->flank_owner = nullptr;
      if (s->desc_owner == this) s->desc_owner = nullptr;
      if (pin) s->pin_free.push_back(pin);
      if (views.capacity() > s->spare_views.capacity()) { views.clear(); s->spare_views.swap(views); }
      if (jobs.capacity() > s->spare_jobs.capacity()) s->spare_jobs.swap(jobs);
      if (flank_off.capacity() > s->spare_u32[0].capacity()) s->spare_u32[0].swap(flank_off);
      if (arena_off.capacity() > s->spare_sz.capacity()) s->spare_sz.swap(arena_off);
      if (prep.capacity() > s->spare_prep.capacity()) s->spare_prep.swap(prep);  // (elements kept: analyze_gap resets what it uses)
    }
  }
};

// pinned result buffers are recycled through the session: allocating page-locked memory
// costs more than a whole 500-gap batch.  The k-th launch group of a run always uses the
// session's k-th slot, so buffer sizes settle after the first run.
void g2s_batch::drop_tiers() { tiers.clear(); }
static TierData* take_tier(g2s_session* s, size_t /*unused*/) {
  const size_t slot = s->tier_cursor++;
  while (s->tier_pool.size() <= slot) s->tier_pool.push_back(new TierData());
  TierData* t = (TierData*)s->tier_pool[slot];
  t->gap_ids.clear();
  return t;
}

// Can a list of n gaps be finished on the device (run_resident)?  The checks that do not look at the gaps.
static bool resident_applicable(const g2s_session* s, size_t n) {
  if (s->resident_off || n == 0) return false;
  const int forced = GENV("G2S_RESIDENT") ? atoi(GENV("G2S_RESIDENT")) : -1;  // (1: lists of any length; 0: never)
  // (lists of a few dozen gaps: the host analyses gaps while the launch's stragglers run and is done before four
  // more launches would be; measured on config 2's 500 gaps: 0.32 ms on the device against 0.35-0.39 ms.  The
  // groups of a team's list may be short: the list is what counts)
  if (forced == 0 || (forced != 1 && !s->in_team_list && n < 256)) return false;
  if (GENV("G2S_NO_SEG_TIER") || GENV("G2S_FORCE_SEGX") || GENV("G2S_HOST_D2") || GENV("G2S_SEG_DUMP") ||
      GENV("G2S_DUMP_STATS") || GENV("G2S_NO_LDS_TIER") || GENV("G2S_STATE_D2"))
    return false;
  const Graph& g = *s->graph->g;
  auto it = g.dev.find(s->device);
  if (it == g.dev.end()) return false;
  const DeviceGraph& dg = it->second;
  return dg.pred == nullptr && dg.rem != nullptr && !s->no_lds_tier && g.n < (1ull << 28) - 1;
}

extern "C" int g2s_batch_prepare(g2s_session* s, const g2s_gap* gaps, size_t n, g2s_batch** out) {
  g2s_env_sync();
  if (!s || (!gaps && n) || !out) return fail(G2S_ERR_ARG, "g2s_batch_prepare: bad argument");
  REFUSE_IN_FLIGHT(s, "g2s_batch_prepare");
  const Graph& g = *s->graph->g;
  const int k = g.k;
  const auto tp0 = std::chrono::steady_clock::now();
  g2s_batch* b = new g2s_batch();
  b->s = s;
  memset(&b->timing, 0, sizeof b->timing);
  // (the per-gap arrays of a batch come from the session and go back to it: 0.5 MB of fresh pages per 10 000-gap
  // list cost more than filling them)
  b->jobs.swap(s->spare_jobs);
  b->flank_off.swap(s->spare_u32[0]);
  b->arena_off.swap(s->spare_sz);
  b->jobs.resize(n);
  b->flank_off.resize(n);
  b->arena_off.resize(n);
  const int d_err = s->params.d_err;
  // ---- sizes: what fill_gap reads of the flanks is the first k+lmf characters of the left one and the
  // first and last k+rmf of the right one; the k-mer -> node look-ups run on the device (flank_lookup.hip)
  // (a long list: in parallel — every task its stretch of the list with offsets from the stretch's start, the
  // stretches' bases from a pass over the <= 16 tasks, added where the second pass below visits the gap.  One
  // thread took 56 us for 10 000 gaps, in front of everything else the list needs.)
  size_t text_bytes = 0, n_nodes = 0, n_desc = 0, tb_max = 0;
  std::vector<uint32_t>& text_off = s->spare_u32[1];
  std::vector<uint32_t>& didx = s->spare_u32[2];  // (didx: descriptor index of every valid gap)
  text_off.resize(n);
  didx.resize(n);
  const bool force_host_lookup = GENV("G2S_HOST_LOOKUP") != nullptr;
  static const size_t max_tasks = getenv("G2S_PREP_TASKS") ? (size_t)std::max(1, atoi(getenv("G2S_PREP_TASKS"))) : 16;
  const size_t per_task = std::max<size_t>(256, (n + max_tasks - 1) / max_tasks);  // (at most 16 tasks: every task wakes a thread)
  const size_t ntasks = (n + per_task - 1) / per_task;
  struct Part {
    size_t n_nodes = 0, text_bytes = 0, n_desc = 0, arena_bytes = 0, rnd_cap = 0;
    uint64_t flank_bytes = 0;
    int gmax = 0, dmax = 0;
    size_t tb_max = 0;
    bool seg_all = true, host_lookup = false, has_skip = false;
  };
  std::vector<Part> parts(std::max<size_t>(ntasks, 1));
  auto size_range = [&](size_t t) {
    Part& P = parts[t];
    const size_t lo = t * per_task, hi = std::min(n, lo + per_task);
    for (size_t i = lo; i < hi; i++) {
      GapJob& j = b->jobs[i];
      const g2s_gap& in = gaps[i];
      j.g = in.gap_len;
      j.lmf = in.lmf;
      j.rmf = in.rmf;
      j.skip_if_prev_right_fuz_gt = in.skip_if_prev_right_fuz_gt;
      // D2: flanks too short would make the reference throw from substr
      j.bad_flank = !in.left || !in.right || in.lmf < 0 || in.rmf < 0 || in.gap_len < 0 ||
                    in.left_len < k + in.lmf || in.right_len < k + in.rmf;
      b->flank_off[i] = (uint32_t)P.n_nodes;
      didx[i] = 0xFFFFFFFFu;
      if (!j.bad_flank) {
        const size_t tb = (size_t)(k + j.lmf) + 2 * (size_t)(k + j.rmf);
        if (tb > G2S_FLANK_TEXT_MAX || j.lmf > 65535 || j.rmf > 65535 || force_host_lookup) P.host_lookup = true;
        text_off[i] = (uint32_t)P.text_bytes;
        P.text_bytes += (tb + 3) & ~(size_t)3;
        P.tb_max = std::max(P.tb_max, tb);
        P.n_nodes += (size_t)(j.lmf + 1) + 2 * (size_t)(j.rmf + 1);
        didx[i] = (uint32_t)P.n_desc++;
        P.flank_bytes += (uint64_t)in.left_len + (uint64_t)in.right_len;
        const int dd = j.lmf + j.rmf + j.g + d_err;
        if (j.rmf > 31 || j.lmf > 31 || dd >= 32767) P.seg_all = false;
        P.gmax = std::max(P.gmax, j.g);
        P.dmax = std::max(P.dmax, dd);
        P.rnd_cap += (size_t)(dd + 2);
      } else {
        j.lmf = std::max(0, j.lmf);
        j.rmf = std::max(0, j.rmf);
        j.g = std::max(0, j.g);
      }
      if (j.skip_if_prev_right_fuz_gt >= 0) P.has_skip = true;
      b->arena_off[i] = P.arena_bytes;
      P.arena_bytes += j.buf_bytes(k, d_err);
    }
  };
  // (on the pool although one thread takes this pass in 42 us against 55-67: the pass behind it then finds the workers
  // awake — on one thread here that pass took 150-370 us instead of 35, LAB_NOTES round 6)
  if (ntasks > 4) s->pool->run(ntasks, size_range);
  else for (size_t t = 0; t < ntasks; t++) size_range(t);
  std::vector<size_t> base_nodes(ntasks + 1), base_text(ntasks + 1), base_desc(ntasks + 1), base_arena(ntasks + 1);
  for (size_t t = 0; t < ntasks; t++) {
    const Part& P = parts[t];
    base_nodes[t] = n_nodes; base_text[t] = text_bytes; base_desc[t] = n_desc; base_arena[t] = b->arena_bytes;
    n_nodes += P.n_nodes; text_bytes += P.text_bytes; n_desc += P.n_desc; b->arena_bytes += P.arena_bytes;
    b->rnd_cap += P.rnd_cap;
    b->timing.flank_bytes += P.flank_bytes;
    b->gmax = std::max(b->gmax, P.gmax);
    b->dmax = std::max(b->dmax, P.dmax);
    tb_max = std::max(tb_max, P.tb_max);
    if (!P.seg_all) b->seg_tier_all = false;
    if (P.host_lookup) b->host_lookup = true;
    if (P.has_skip) b->has_skip = true;
  }
  b->n_valid = n_desc;
  const auto tp1 = std::chrono::steady_clock::now();
  if (hipSetDevice(s->device) != hipSuccess) { delete b; return fail(G2S_ERR_NO_DEVICE, "cannot select device"); }
  // resident mode is likely to take this list: its descriptors are filled by the pass below as well
  GapLite* fast_gd = nullptr;  // (resident mode's descriptors: 32 bytes a gap — fill_device.h — the kernels expand them)
  D3Gap* fast_dg = nullptr;
  if (resident_applicable(s, n) && b->seg_tier_all && !b->host_lookup) {
    if (s->h_gaps.ensure(n * sizeof(GapDev) + n * 4 + 16) == hipSuccess &&
        s->h_d3.ensure(n * sizeof(D3Gap) + 2048 + 64 * 128 + G2S_RAND_WINDOW * 4) == hipSuccess) {
      fast_gd = (GapLite*)s->h_gaps.p;
      fast_dg = (D3Gap*)s->h_d3.p;
      s->desc_owner = b;
      b->fast_desc = true;
    }
  }
  // (the regular segment tier's waves resolve their own flanks when resident mode takes the list — not a deep list,
  // whose longest gaps start in the large variant; G2S_FLANK_KERNEL=1: the look-up kernel as until round 5.  A list
  // without a bad flank gets its text at a FIXED STRIDE: when the launch takes the gaps in list order, a wave knows
  // where its text is before its descriptor has arrived — on a short list both come over the link.)
  b->inline_ok = fast_gd != nullptr && b->dmax < 2500 && !GENV("G2S_FLANK_KERNEL");
  uint32_t tstride = 0;
  if (b->inline_ok && n_desc == n && tb_max <= 508 && !GENV("G2S_NO_TEXT_STRIDE")) {
    tstride = (uint32_t)((tb_max + 3) & ~(size_t)3);
    if ((uint64_t)n * tstride < (1ull << 31)) text_bytes = (size_t)n * tstride; else tstride = 0;
  }
  b->text_stride = tstride;
  if (!s->pin_free.empty()) { b->pin = s->pin_free.back(); s->pin_free.pop_back(); }
  else b->pin = new PinBuf();
  const size_t desc_bytes = (n_desc * sizeof(FlankDesc) + 15) & ~(size_t)15;
  const size_t text_pad = (text_bytes + 16 + 15) & ~(size_t)15;
  if (b->pin->ensure(desc_bytes + text_pad + n_nodes * 4 + 16) != hipSuccess) { delete b; return fail(G2S_ERR_NOMEM, "pinned flank buffer"); }
  b->desc = (FlankDesc*)b->pin->p;
  b->text = (char*)b->pin->p + desc_bytes;
  b->nodes = (uint32_t*)((char*)b->pin->p + desc_bytes + text_pad);
  b->n_desc = n_desc;
  b->n_nodes = n_nodes;
  // ---- flank text and descriptors into the pinned buffer (long lists: on the pool)
  {
    // (what: 1 = flank text and look-up descriptors, 2 = the fill kernel's and phase D3's descriptors, 3 = both)
    auto do_range = [&](size_t t, int what) {
      const size_t lo = t * per_task, hi = std::min(n, lo + per_task);
      for (size_t i = lo; i < hi; i++) {
        GapJob& j = b->jobs[i];
        if (what & 1) {  // (the first visit: the offsets within the task's stretch become offsets within the list)
          b->flank_off[i] += (uint32_t)base_nodes[t];
          b->arena_off[i] += base_arena[t];
          if (!j.bad_flank) { text_off[i] = tstride ? (uint32_t)i * tstride : text_off[i] + (uint32_t)base_text[t]; didx[i] += (uint32_t)base_desc[t]; }
        }
        j.nodes = b->nodes + b->flank_off[i];
        if (fast_gd && (what & 2)) {
          GapLite& d = fast_gd[i];
          D3Gap& q = fast_dg[i];
          q.arena_off = (uint64_t)b->arena_off[i];  // (within the batch's share of the arena: D3Params.arena_base is added on the device)
          q.skip_thr = std::max(-1, std::min(j.skip_if_prev_right_fuz_gt, 32767));
          q.lmf = (uint16_t)j.lmf;
          q.kind = j.bad_flank ? 1 : 0;
          q.pad = 0;
          d.g = j.g; d.lmf = (uint16_t)j.lmf; d.rmf = (uint16_t)j.rmf;
          d.flank_off = b->flank_off[i];
          d.text_off = j.bad_flank ? 0u : text_off[i];                // (where the gap's flank text starts: look-ups in the kernel)
          d.arena_off = (uint64_t)b->arena_off[i];                    // (its fill buffer in the batch's share of the arena,
          d.has_skip = j.skip_if_prev_right_fuz_gt >= 0 ? 1u : 0u;     // and whether a skip rule decides over it: tracebacks in the kernel)
          d.pad = 0u;
        }
        if (!j.bad_flank && (what & 1)) j.text_off = text_off[i];
        if (j.bad_flank || !(what & 1)) continue;
        const g2s_gap& in = gaps[i];
        char* t = b->text + text_off[i];
        const size_t ll = (size_t)(k + j.lmf), rl2 = (size_t)(k + j.rmf);
        memcpy(t, in.left, ll);
        memcpy(t + ll, in.right, rl2);
        memcpy(t + ll + rl2, in.right + ((size_t)in.right_len - rl2), rl2);
        FlankDesc& d = b->desc[didx[i]];
        d.text_off = text_off[i];
        d.flank_off = b->flank_off[i];
        d.lmf = (uint16_t)j.lmf;
        d.rmf = (uint16_t)j.rmf;
        if (b->host_lookup) {  // node ids on the host (Graph::node_of), uploaded by upload_flanks
          uint32_t* fn = b->nodes + b->flank_off[i];
          for (int x = 0; x <= j.lmf; x++) *fn++ = g.node_of(t + x);                          // :995,1083
          for (int x = 0; x <= j.rmf; x++) *fn++ = g.node_of(t + ll + rl2 + (rl2 - k - x));   // :878,954
          for (int x = 0; x <= j.rmf; x++) *fn++ = g.node_of(t + ll + x);                     // :1113
        }
      }
    };
    if (ntasks > 4) s->pool->run(ntasks, [&](size_t t) { do_range(t, 3); });
    else if (fast_gd) {
      // a short list on this thread: the look-up kernel is launched as soon as its input is there, the other
      // descriptors are written while it runs
      for (size_t t = 0; t < ntasks; t++) do_range(t, 1);
      const int rc0 = b->upload_flanks(true);
      if (rc0 != G2S_OK) { delete b; return rc0; }
      for (size_t t = 0; t < ntasks; t++) do_range(t, 2);
    } else for (size_t t = 0; t < ntasks; t++) do_range(t, 3);
  }
  const auto tp2 = std::chrono::steady_clock::now();
  const int rc = b->upload_flanks(true);
  if (rc != G2S_OK) { delete b; return rc; }
  if (n >= 1024 && GENV("G2S_DEBUG"))
    fprintf(stderr, "[g2s] prepare: sizes %.3f ms, text + descriptors %.3f ms, look-up launch %.3f ms\n",
            std::chrono::duration<double, std::milli>(tp1 - tp0).count(), std::chrono::duration<double, std::milli>(tp2 - tp1).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp2).count());
  *out = b;
  return G2S_OK;
}

// d_flank belongs to the session; the batch that ran last owns its contents.  The node ids are
// computed on the session's stream by the look-up kernel (no host synchronisation: the fill
// kernels follow on the same stream; the host reads them from pinned memory only after a fill
// kernel has reported gaps as done).
int g2s_batch::fetch_nodes() {
  if (!nodes_dev_only) return G2S_OK;
  if (s->flank_owner != this) return fail(G2S_ERR_STATE, "batch flank nodes: the device table belongs to another batch");
  if (hipSetDevice(s->device) != hipSuccess) return fail(G2S_ERR_NO_DEVICE, "cannot select device");
  hipError_t e = hipStreamSynchronize(s->stream);
  if (e == hipSuccess && n_nodes) e = hipMemcpy(nodes, s->d_flank.p, n_nodes * 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(G2S_ERR_HIP, std::string("batch flank nodes: ") + hipGetErrorString(e));
  nodes_dev_only = false;
  return G2S_OK;
}
int g2s_batch::upload_flanks(bool allow_inline) {
  if (s->flank_owner == this && (!inline_pending || allow_inline)) return G2S_OK;
  if (hipSetDevice(s->device) != hipSuccess) return fail(G2S_ERR_NO_DEVICE, "cannot select device");
  hipError_t e = s->d_flank.ensure(std::max<size_t>(n_nodes * 4, 16));
  const bool staged = s->flank_owner == this && inline_pending;  // (the text is where the kernel reads it already)
  inline_pending = false;
  if (e == hipSuccess && n_nodes && allow_inline && inline_ok && !host_lookup) {
    // the fill kernel's waves do the look-ups: the text where they can read it (a long list: one copy to device memory)
    void* d_text = nullptr;
    e = hipHostGetDevicePointer(&d_text, text, 0);
    if (e == hipSuccess && n_desc > 2048) {
      const size_t bytes = (size_t)((const char*)nodes - (const char*)text);
      e = s->d_fstage.ensure(bytes);
      if (e == hipSuccess) e = hipMemcpyAsync(s->d_fstage.p, text, bytes, hipMemcpyHostToDevice, s->stream);
      d_text = s->d_fstage.p;
    }
    if (e != hipSuccess) return fail(G2S_ERR_HIP, std::string("batch flank text: ") + hipGetErrorString(e));
    inline_text_dev = (const char*)d_text;
    inline_pending = true;
    s->flank_owner = this;
    return G2S_OK;
  }
  if (e == hipSuccess && n_nodes) {
    if (host_lookup) {
      e = hipMemcpyAsync(s->d_flank.p, nodes, n_nodes * 4, hipMemcpyHostToDevice, s->stream);
    } else {
      void *d_desc = nullptr, *d_text = nullptr, *d_nodes = nullptr;
      e = hipHostGetDevicePointer(&d_desc, desc, 0);
      if (e == hipSuccess) e = hipHostGetDevicePointer(&d_text, text, 0);
      if (e == hipSuccess) e = hipHostGetDevicePointer(&d_nodes, nodes, 0);
      // (a long list: descriptors and flank text go to device memory in one copy in front of the kernel — 10 000
      // workgroups that each read their descriptor and then their text over the link are two round trips of the link
      // each: 0.10 ms for config 3's list, against a 1.5 MB copy and a kernel that reads device memory)
      if (e == hipSuccess && n_desc > 2048 && !GENV("G2S_FLANKS_OVER_THE_LINK") && !staged) {
        const size_t bytes = (size_t)((const char*)nodes - (const char*)desc);  // [descriptors][text], contiguous
        e = s->d_fstage.ensure(bytes);
        if (e == hipSuccess) e = hipMemcpyAsync(s->d_fstage.p, desc, bytes, hipMemcpyHostToDevice, s->stream);
        d_desc = s->d_fstage.p;
        d_text = (char*)s->d_fstage.p + ((const char*)text - (const char*)desc);
      }
      if (e == hipSuccess)
        e = launch_resolve_flanks(s->stream, s->lookup, (uint32_t)n_desc, (const FlankDesc*)d_desc, (const char*)d_text,
                                  (uint32_t*)s->d_flank.p, (uint32_t*)d_nodes);
    }
  }
  if (e != hipSuccess) return fail(G2S_ERR_HIP, std::string("batch flank look-up: ") + hipGetErrorString(e));
  s->flank_owner = this;
  return G2S_OK;
}

extern "C" size_t g2s_batch_arena_bytes(const g2s_batch* b) { return b ? b->arena_bytes : 0; }
extern "C" int g2s_batch_timing(const g2s_batch* b, g2s_timing* out) {
  if (!b || !out) return fail(G2S_ERR_ARG, "g2s_batch_timing: bad argument");
  *out = b->timing;
  return G2S_OK;
}
extern "C" void g2s_batch_free(g2s_batch* b) { delete b; }

namespace {

struct Plan {  // per-gap capacities at one scale
  uint32_t rlog_cap, slog_cap;
  uint64_t bytes;
};

// LDS tier: capacity of the in-LDS right set for one gap, 0 when the gap is not eligible
// `room` = entries the launch can afford per gap (LDS per CU / gaps per CU).
uint32_t lds_rs_cap(const GapJob& j, int d_err, uint32_t room) {
  const int right_half = j.rmf + (j.g + d_err + 1) / 2;
  if (j.rmf > (int)fill_lds_max_fuz()) return 0;
  const uint64_t need = (uint64_t)pow2ceil(std::max<uint64_t>(512, 2ull * (uint64_t)(right_half + j.rmf + 2)));
  if (need > room) return 0;
  // small batches leave most of the 160 KB/CU unused: take it, branching right sets then stay in LDS
  return (uint32_t)std::max<uint64_t>(need, room);
}
// entries of right set each gap may have in LDS when `ngaps` gaps share the chip
uint32_t lds_room(size_t ngaps) {
  const size_t per_cu = std::max<size_t>(1, (ngaps + 255) / 256);
  const size_t bytes = (160u * 1024u) / std::min<size_t>(per_cu, 6) - fill_lds_bytes(0, 64) - 512;
  uint32_t cap = 512;
  while ((size_t)cap * 2 * 4 <= bytes && cap < 16384) cap <<= 1;
  return cap;
}

Plan plan_gap(const GapJob& j, int d_err, uint64_t scale, uint64_t max_states) {
  const int right_half = j.rmf + (j.g + d_err + 1) / 2;
  const int D = j.lmf + j.rmf + j.g + d_err;
  uint64_t r = (uint64_t)pow2ceil(std::max<uint64_t>(256, 4ull * (uint64_t)(right_half + j.rmf + 2))) * scale;
  uint64_t st = (uint64_t)pow2ceil(std::max<uint64_t>(2048, 8ull * (uint64_t)(D + 2))) * scale;
  const uint64_t lim = std::max<uint64_t>(1024, std::min<uint64_t>(max_states, 1u << 28));
  uint64_t limp = 1;
  while (limp * 2 <= lim) limp <<= 1;
  Plan p;
  p.rlog_cap = (uint32_t)std::min(r, limp);
  p.slog_cap = (uint32_t)std::min(st, limp);
  // right set (hash + log), state table (keys, counts, marks), state log, closure scratch + packed output
  p.bytes = (uint64_t)p.rlog_cap * (8 + 4) + (uint64_t)p.slog_cap * (2 * 16 + 4 + 2 * sizeof(SubState));
  return p;
}

// Launch phases A-D1 for the listed gaps and bring the results back.  lds = true: the
// LDS-resident kernels (fill_lds.hip); false: the general tier with per-gap tables in
// HBM (fill_kernels.hip) at the given table scale.
// on_done (LDS tier): called on the calling thread while the kernel runs, with the indices of
// gaps whose results have arrived in td->outs / td->subs since the last call.
using DoneFn = std::function<void(const uint32_t* gaps_done, size_t count)>;
int run_tier(g2s_batch* b, const std::vector<uint32_t>& ids, uint64_t scale, uint64_t max_states, TierData* td,
             bool lds, uint32_t lds_room_override = 0, bool rs_in_hbm = false, uint32_t fcap = 64,
             const DoneFn* on_done = nullptr,
             int seg = 0 /* segment tier (fill_seg.hip): 1 the tier proper, 2 its large variant; needs lds = true */) {
  const auto t_enter = std::chrono::steady_clock::now();
  g2s_session* s = b->s;
  uint32_t* seg_dbg = nullptr;
  const DeviceGraph& dg = s->graph->g->dev.at(s->device);
  const size_t n = b->jobs.size();
  const int d_err = s->params.d_err;
  HIP_TRY(s->h_gaps.ensure(n * sizeof(GapDev) + ids.size() * 4 + 16));
  GapDev* gd = (GapDev*)s->h_gaps.p;
  uint32_t* ids_pinned = (uint32_t*)(gd + n);  // the LDS tier reads descriptors and ids from here (no upload)
  if (!ids.empty()) memcpy(ids_pinned, ids.data(), ids.size() * 4);
  memset(gd, 0, n * sizeof(GapDev));
  uint64_t rs_total = 0, rlog_total = 0, st_total = 0, slog_total = 0, lvl_total = 0, xl_total = 0, out_states = 0, out_max = 0;
  uint32_t lds_cap_max = 0;
  td->gap_ids = ids;
  for (size_t x = 0; x < ids.size(); x++) {
    const uint32_t i = ids[x];
    const GapJob& j = b->jobs[i];
    const Plan p = plan_gap(j, d_err, scale, max_states);
    GapDev& d = gd[i];
    d.g = j.g; d.e = d_err; d.lmf = j.lmf; d.rmf = j.rmf;
    d.D = j.lmf + j.rmf + j.g + d_err;
    d.right_half = j.rmf + (j.g + d_err + 1) / 2;
    d.prune_from = j.g / 2 + d_err / 2 + j.lmf;
    d.all_paths = s->params.all_paths ? 1 : 0;
    d.flank_off = b->flank_off[i];
    d.rlog_cap = p.rlog_cap; d.rs_mask = 2 * p.rlog_cap - 1;
    d.slog_cap = p.slog_cap; d.st_mask = 2 * p.slog_cap - 1;
    d.rs_off = rs_total; rs_total += 2ull * p.rlog_cap;
    d.rlog_off = rlog_total; rlog_total += p.rlog_cap;
    d.st_off = st_total; st_total += 2ull * p.slog_cap;
    d.slog_off = slog_total; slog_total += p.slog_cap;
    d.lvl_off = lvl_total; lvl_total += (uint64_t)(d.D + 2);
    // pinned output pool of the LDS tier: 4 states per DP level and gap, plus room for one
    // gap's worst case (what does not fit is run again by the next pass)
    out_states += std::min<uint64_t>(p.slog_cap, 4ull * (uint64_t)(d.D + 2));
    out_max = std::max<uint64_t>(out_max, p.slog_cap);
    if (lds && !seg) {  // extra-parent list of merged states: a quarter of the log's capacity
      d.pad0 = std::max(64u, p.slog_cap / 4);
      d.st_off = xl_total;
      xl_total += d.pad0;
    }
    if (lds && !rs_in_hbm && !seg) {
      const uint32_t c = lds_rs_cap(j, d_err, lds_room_override ? lds_room_override : lds_room(ids.size()));
      d.rs_mask = c - 1;
      lds_cap_max = std::max(lds_cap_max, c);
    }
  }
  HIP_TRY(s->d_gaps.ensure(n * sizeof(GapDev)));
  HIP_TRY(s->d_ids.ensure(std::max<size_t>(ids.size() * 4, 16)));
  HIP_TRY(s->d_outs.ensure(n * sizeof(GapOut)));
  if (!seg) HIP_TRY(s->d_subscr.ensure(slog_total * (lds ? sizeof(SubRec) : sizeof(SubState))));
  if (!lds) HIP_TRY(s->d_subout.ensure(slog_total * sizeof(SubState)));
  HIP_TRY(s->d_counter.ensure(32));
  hipStream_t st = s->stream;
  if (!lds) {
    HIP_TRY(hipMemcpyAsync(s->d_gaps.p, gd, n * sizeof(GapDev), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->d_ids.p, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, st));
  }
  // d_outs, the cursors and the spill pool are reset right after an LDS-tier launch has
  // finished, while the host works on its results: the next launch then starts behind one
  // copy instead of behind three fill kernels (~25 us on the stream)
  size_t pool_bytes_used = 0, xcd_bytes_used = 0;
  if (s->d_outs.clean < n * sizeof(GapOut)) HIP_TRY(hipMemsetAsync(s->d_outs.p, 0, n * sizeof(GapOut), st));
  if (s->d_counter.clean < 32) HIP_TRY(hipMemsetAsync(s->d_counter.p, 0, 32, st));  // [0] output cursor, [1] completion-list cursor, [2] spill-pool cursor, [3] log-pool cursor
  s->d_outs.clean = 0;
  s->d_counter.clean = 0;
  if (lds) {
    if (!seg) {
      HIP_TRY(s->d_log.ensure(slog_total * 8));
      HIP_TRY(s->d_lvl.ensure(lvl_total * 4));
      HIP_TRY(s->d_plk.ensure(slog_total * 4));
      HIP_TRY(s->d_xl.ensure(std::max<uint64_t>(xl_total * 8, 16)));
      HIP_TRY(s->d_xo.ensure(std::max<uint64_t>(xl_total * 8, 16)));
    }
    if (rs_in_hbm) {
      HIP_TRY(s->d_rs.ensure(rs_total * 4));
      HIP_TRY(hipMemsetAsync(s->d_rs.p, 0xFF, rs_total * 4, st));
    }
    HIP_TRY(hipEventRecord(s->ev[0], st));
    HIP_TRY(hipEventRecord(s->ev[1], st));  // phases A-C are one kernel in this tier
    const uint32_t num_oriented = (uint32_t)(2 * s->graph->g->n);
    // results go straight to pinned host memory (closures packed by an atomic cursor)
    HIP_TRY(td->outs.ensure(n * sizeof(GapOut)));
    out_states += out_max;
    // (segment tier: two 16-byte units per closure segment; ~15 segments per gap, at most G2S_SEG_CAP)
    if (seg == 1) out_states = (uint64_t)ids.size() * 128u + 2u * G2S_SEG_CAP;
    // (large variant: closures of a few thousand segments; what does not fit runs in the LDS tier)
    if (seg == 2) out_states = std::min<uint64_t>((uint64_t)ids.size() * 4096u, 8ull << 20) + 2u * G2S_SEGX_CAP;  // (<= 128 MB pinned)
    HIP_TRY(td->subs.ensure(std::max<uint64_t>(out_states * sizeof(SubRec), 16)));
    HIP_TRY(td->done.ensure(std::max<size_t>(ids.size() * 4, 16)));
    memset(td->done.p, 0xFF, ids.size() * 4);
    void *d_outs_host = nullptr, *d_subs_host = nullptr, *d_done_host = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&d_outs_host, td->outs.p, 0));
    HIP_TRY(hipHostGetDevicePointer(&d_subs_host, td->subs.p, 0));
    HIP_TRY(hipHostGetDevicePointer(&d_done_host, td->done.p, 0));
    // spill pool for right sets that outgrow their LDS share (launches with the right set in LDS):
    // chunks of 32 K entries for an eighth of the gaps
    const uint32_t chunk_entries = 32768u;
    uint32_t pool_chunks = 0;
    if (!rs_in_hbm && lds_cap_max < chunk_entries && !seg) {
      pool_chunks = (uint32_t)std::min<size_t>(ids.size() / 8 + 16, (s->mem_budget / 8) / ((size_t)chunk_entries * 4));
      HIP_TRY(s->d_rspool.ensure((size_t)pool_chunks * chunk_entries * 4));
      if (s->d_rspool.clean < (size_t)pool_chunks * chunk_entries * 4)
        HIP_TRY(hipMemsetAsync(s->d_rspool.p, 0xFF, (size_t)pool_chunks * chunk_entries * 4, st));
      s->d_rspool.clean = 0;
      pool_bytes_used = (size_t)pool_chunks * chunk_entries * 4;
    }
    // log pool: a gap whose state log outgrows its slice takes a chunk and goes on (a handful
    // of repeat-rich gaps per ten thousand; without it they would run again after the launch)
    const uint32_t log_chunk_states = 131072u;
    const uint32_t log_chunks = (uint32_t)std::min<size_t>(ids.size() / 64 + 4, 256);
    if (!seg) HIP_TRY(s->d_logpool.ensure((size_t)log_chunks * fill_lds_log_chunk_bytes(log_chunk_states)));
    // short lists: gap descriptors and the id list are read once per gap, straight from pinned
    // host memory — with no upload in front of it the kernel starts ~13 us earlier (2-3 % of a
    // 500-gap step); at 10 000 gaps the reads over the link cost more (kernel +4 %) than the
    // upload, so long lists keep it
    void* d_gaps_host = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&d_gaps_host, s->h_gaps.p, 0));
    const GapDev* gaps_dev = (const GapDev*)d_gaps_host;
    const uint32_t* ids_dev = (const uint32_t*)(gaps_dev + n);
    if (ids.size() > 2048) {
      HIP_TRY(hipMemcpyAsync(s->d_gaps.p, gd, n * sizeof(GapDev), hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(s->d_ids.p, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, st));
      gaps_dev = (const GapDev*)s->d_gaps.p;
      ids_dev = (const uint32_t*)s->d_ids.p;
    }
    const char* seg_dump = seg ? GENV("G2S_SEG_DUMP") : nullptr;  // diagnostics: phase A entries + segments of every gap
    const uint32_t seg_dbg_w = seg == 2 ? fill_segx_dbg_words() : fill_seg_dbg_words();
    const bool seg_two_waves = GENV("G2S_SEG_WAVES") ? atoi(GENV("G2S_SEG_WAVES")) == 2 : ids.size() <= 2048;
    if (seg == 1 && seg_two_waves) b->timing.seg2_launches++;
    // long lists: finished gaps are announced in batches of 16 per XCD (one L2 write-back per batch instead of per
    // gap; fill_seg.hip, `publish`): config 3's launch 0.8 -> 0.48 ms.  Short lists announce every gap by itself:
    // their launch ends with its slowest gap either way (config 2: 0.152 ms with batches of 4, 0.153 without), and
    // the gaps of unfinished batches would be analysed behind the launch's end instead of under it (config 2's
    // step 0.39 ms against 0.37).  G2S_PUBLISH_BATCH=N forces (1 = every gap by itself).
    uint32_t pub_batch = GENV("G2S_PUBLISH_BATCH") ? (uint32_t)atoi(GENV("G2S_PUBLISH_BATCH")) : (ids.size() <= 2048 ? 1u : 16u);
    size_t xcd_bytes = 0;
    if (seg == 1 && pub_batch > 1) {
      xcd_bytes = 64 + 8 * ids.size() * 4;
      HIP_TRY(s->d_xcd.ensure(xcd_bytes));
      if (s->d_xcd.clean < xcd_bytes) {
        HIP_TRY(hipMemsetAsync(s->d_xcd.p, 0, 64, st));
        HIP_TRY(hipMemsetAsync((char*)s->d_xcd.p + 64, 0xFF, xcd_bytes - 64, st));
      }
      s->d_xcd.clean = 0;
      xcd_bytes_used = xcd_bytes;
    }
    if (seg_dump) {
      HIP_TRY(s->d_slog.ensure((size_t)ids.size() * seg_dbg_w * 4));
      HIP_TRY(hipMemsetAsync(s->d_slog.p, 0, (size_t)ids.size() * seg_dbg_w * 4, st));
      seg_dbg = (uint32_t*)s->d_slog.p;
    }
    if (seg == 2) {
      // one persistent workgroup per compute unit (the variant takes nearly all of a CU's LDS)
      const uint32_t wgs = (uint32_t)std::min<size_t>(ids.size(), (size_t)std::max(1, s->num_cus));
      // (G2S_SEGX_ONE_WAVE=1: round 3's kernel, one wave per compute unit — measurements only)
      static const bool one_wave = getenv("G2S_SEGX_ONE_WAVE") != nullptr;
      HIP_TRY(s->d_segx.ensure(std::max(fill_segx_scratch_bytes(wgs), fill_segw_scratch_bytes(wgs))));
      if (!one_wave)
        HIP_TRY(launch_fill_segw(st, (uint32_t)ids.size(), wgs, dg.succ, dg.urec, gaps_dev, ids_dev, (const uint32_t*)s->d_flank.p,
                                 (SubRec*)d_subs_host, (unsigned long long)out_states, (unsigned long long*)s->d_counter.p,
                                 (GapOut*)s->d_outs.p, (GapOut*)d_outs_host, (uint32_t*)d_done_host,
                                 s->params.skip_confident ? 1 : 0, seg_dbg, (uint32_t*)s->d_segx.p,
                                 (unsigned long long*)s->d_counter.p + 2));
      else
      HIP_TRY(launch_fill_segx(st, (uint32_t)ids.size(), wgs, dg.succ, dg.urec, gaps_dev, ids_dev, (const uint32_t*)s->d_flank.p,
                               (SubRec*)d_subs_host, (unsigned long long)out_states, (unsigned long long*)s->d_counter.p,
                               (GapOut*)s->d_outs.p, (GapOut*)d_outs_host, (uint32_t*)d_done_host,
                               s->params.skip_confident ? 1 : 0, seg_dbg, (uint32_t*)s->d_segx.p,
                               (unsigned long long*)s->d_counter.p + 2));
    } else if (seg)
      HIP_TRY(launch_fill_seg(st, (uint32_t)ids.size(), dg.succ, dg.urec, gaps_dev, ids_dev, (const uint32_t*)s->d_flank.p,
                              (SubRec*)d_subs_host, (unsigned long long)out_states, (unsigned long long*)s->d_counter.p,
                              (GapOut*)s->d_outs.p, (GapOut*)d_outs_host, (uint32_t*)d_done_host,
                              s->params.skip_confident ? 1 : 0, seg_dbg,
                              // short lists are latency-bound (the launch ends with its slowest gap): two waves per
                              // gap; long lists fill the chip and are throughput-bound: one (G2S_SEG_WAVES=1|2 forces)
                              seg_two_waves, (unsigned long long*)s->d_xcd.p, (uint32_t*)((char*)s->d_xcd.p + 64),
                              (uint32_t)ids.size(), pub_batch));
    else
    HIP_TRY(launch_fill_lds(st, (uint32_t)ids.size(), lds_cap_max, num_oriented, dg.succ, dg.ustart,
                            gaps_dev, ids_dev, (const uint32_t*)s->d_flank.p,
                            (uint64_t*)s->d_log.p, (uint32_t*)s->d_lvl.p, (uint32_t*)s->d_plk.p, (uint64_t*)s->d_xl.p,
                            (uint64_t*)s->d_xo.p, (SubRec*)s->d_subscr.p, (SubRec*)d_subs_host,
                            (unsigned long long)out_states,
                            (unsigned long long*)s->d_counter.p, (GapOut*)s->d_outs.p, (GapOut*)d_outs_host,
                            (uint32_t*)d_done_host, s->params.skip_confident ? 1 : 0,
                            rs_in_hbm ? (uint32_t*)s->d_rs.p : nullptr, fcap, (uint32_t*)s->d_rspool.p, pool_chunks,
                            chunk_entries, s->d_logpool.p, log_chunks, log_chunk_states));
    HIP_TRY(hipEventRecord(s->ev[2], st));
    HIP_TRY(hipEventRecord(s->ev[3], st));
  } else {
    HIP_TRY(s->d_rs.ensure(rs_total * 4));
    HIP_TRY(s->d_rlog.ensure(rlog_total * 4));
    HIP_TRY(s->d_keys.ensure(st_total * 8));
    HIP_TRY(s->d_cnt.ensure(st_total * 4));
    HIP_TRY(s->d_mark.ensure(st_total * 4));
    HIP_TRY(s->d_slog.ensure(slog_total * 4));
    HIP_TRY(hipMemsetAsync(s->d_rs.p, 0xFF, rs_total * 4, st));
    HIP_TRY(hipMemsetAsync(s->d_keys.p, 0xFF, st_total * 8, st));
    HIP_TRY(hipMemsetAsync(s->d_cnt.p, 0, st_total * 4, st));
    HIP_TRY(hipMemsetAsync(s->d_mark.p, 0, st_total * 4, st));
    HIP_TRY(hipEventRecord(s->ev[0], st));
    HIP_TRY(launch_right_bfs(st, (uint32_t)ids.size(), dg.succ, dg.pred, (const GapDev*)s->d_gaps.p,
                             (const uint32_t*)s->d_ids.p, (const uint32_t*)s->d_flank.p, (uint32_t*)s->d_rs.p,
                             (uint32_t*)s->d_rlog.p, (GapOut*)s->d_outs.p));
    HIP_TRY(hipEventRecord(s->ev[1], st));
    HIP_TRY(launch_left_dp(st, (uint32_t)ids.size(), dg.succ, (const GapDev*)s->d_gaps.p,
                           (const uint32_t*)s->d_ids.p, (const uint32_t*)s->d_flank.p, (const uint32_t*)s->d_rs.p,
                           (uint64_t*)s->d_keys.p, (uint32_t*)s->d_cnt.p, (uint32_t*)s->d_slog.p,
                           (GapOut*)s->d_outs.p));
    HIP_TRY(hipEventRecord(s->ev[2], st));
    HIP_TRY(launch_extract(st, (uint32_t)ids.size(), dg.succ, dg.pred, (const GapDev*)s->d_gaps.p,
                           (const uint32_t*)s->d_ids.p, (const uint32_t*)s->d_flank.p, (const uint64_t*)s->d_keys.p,
                           (const uint32_t*)s->d_cnt.p, (uint32_t*)s->d_mark.p, (SubState*)s->d_subscr.p,
                           (SubState*)s->d_subout.p, (unsigned long long*)s->d_counter.p, (GapOut*)s->d_outs.p,
                           s->params.skip_confident ? 1 : 0));
    HIP_TRY(hipEventRecord(s->ev[3], st));
  }

  const auto t_launched = std::chrono::steady_clock::now();
  progress_note("launched: %zu gaps, lds %d seg %d scale %llu", ids.size(), (int)lds, seg, (unsigned long long)scale);
  if (lds) {
   if (seg && on_done) {
    // Segment tier: the analysis of a gap costs a fraction of a microsecond (it runs on the closure
    // segments, or only copies what the device found).  Short lists: whatever has arrived is analysed
    // at once on this thread.  Long lists: a hand-over to the pool for every eighth of the list, so that
    // an eighth at most is left when the kernel ends.  Only this thread polls, napping between looks.
    const volatile uint32_t* done = (const volatile uint32_t*)td->done.p;
    const size_t total = ids.size();
    size_t seen = 0, given = 0;
    const int old_slack = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
    prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
    while (given < total) {
      while (seen < total && done[seen] != 0xFFFFFFFFu) seen++;
      std::atomic_thread_fence(std::memory_order_acquire);
      const bool finished = hipEventQuery(s->ev[2]) != hipErrorNotReady;
      if (finished) { while (seen < total && done[seen] != 0xFFFFFFFFu) seen++; }
      // short lists: whatever has arrived is analysed at once on this thread (a gap whose phase D2 ran on
      // the device costs a fraction of a microsecond here), so that only the last gaps are left
      // when the kernel ends; long lists: two hand-overs to the pool
      if ((total <= 2048 && seen > given) || (total > 2048 && seen - given >= std::max<size_t>(1024, total / 8)) ||
          seen == total || (finished && seen > given)) {
        (*on_done)((const uint32_t*)td->done.p + given, seen - given);
        given = seen;
      } else if (finished) {
        break;  // kernel over and nothing new: an error, reported by the sync below
      } else {
        (*on_done)(nullptr, 0);  // (lets closures that wait for the pool go as soon as it is free)
        struct timespec ts = {0, 5000};
        nanosleep(&ts, nullptr);
      }
    }
    prctl(PR_SET_TIMERSLACK, old_slack > 0 ? (unsigned long)old_slack : 50000UL, 0, 0, 0);
   } else if (on_done) {
      // gaps finish at very different times (the longest take 10x the median): hand the
      // finished ones to the caller while the kernel is still running.  Only this thread
      // polls, and it naps between looks: a busy host must not starve the HIP runtime.
      const volatile uint32_t* done = (const volatile uint32_t*)td->done.p;
      size_t seen = 0, given = 0;
      const size_t total = ids.size();
      // a few large hand-overs (every one of them wakes the worker pool) while gaps arrive in
      // numbers; once arrivals dry up — the stragglers — whatever is there goes out at once,
      // so that next to nothing is left when the last gap finishes
      const size_t chunk = std::max<size_t>(32, total / 8);
      // the naps below are a few microseconds; the default timer slack (50 us) would stretch them
      const int old_slack = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
      prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
      auto last_arrival = std::chrono::steady_clock::now();
      double dbg_first = -1, dbg_fin = -1;
      while (given < total) {
        const size_t before = seen;
        while (seen < total && done[seen] != 0xFFFFFFFFu) seen++;
        std::atomic_thread_fence(std::memory_order_acquire);
        const auto now = std::chrono::steady_clock::now();
        if (seen != before) last_arrival = now;
        const bool finished = hipEventQuery(s->ev[2]) != hipErrorNotReady;
        if (dbg_first < 0 && seen > 0) dbg_first = std::chrono::duration<double, std::milli>(now - t_launched).count();
        if (dbg_fin < 0 && finished) dbg_fin = std::chrono::duration<double, std::milli>(now - t_launched).count();
        const bool lull = seen > given && std::chrono::duration<double, std::micro>(now - last_arrival).count() > 60.0;
        if (seen - given >= chunk || lull || (finished && seen > given) || seen == total) {
          (*on_done)((const uint32_t*)td->done.p + given, seen - given);
          given = seen;
        } else if (finished) {
          while (seen < total && done[seen] != 0xFFFFFFFFu) seen++;
          if (seen == given) break;  // kernel over and nothing new: an error, reported by the sync below
        } else {
          // (short naps once only the stragglers are left: their arrival ends the launch)
          struct timespec ts = {0, seen * 10 > total * 9 ? 4000 : 20000};
          nanosleep(&ts, nullptr);
        }
      }
      prctl(PR_SET_TIMERSLACK, old_slack > 0 ? (unsigned long)old_slack : 50000UL, 0, 0, 0);
      if (GENV("G2S_DEBUG"))
        fprintf(stderr, "[g2s] run_tier: first gap seen %.3f ms after the launch call, kernel seen finished at %.3f ms\n", dbg_first, dbg_fin);
    }
    const auto t_polled = std::chrono::steady_clock::now();
    // (measured: leaving this wait to the end of the run makes the step slower, not faster — the runtime's
    // completion handling then competes with the tracebacks for the host's CPUs)
    HIP_TRY(hipStreamSynchronize(st));  // the kernels wrote td->outs / td->subs themselves
    if (seg_dbg) {
      const uint32_t W = seg == 2 ? fill_segx_dbg_words() : fill_seg_dbg_words();
      const uint32_t ecap = seg == 2 ? G2S_SEGX_EA : 64u * G2S_SEG_ASETS, scap = seg == 2 ? G2S_SEGX_CAP : G2S_SEG_CAP;
      std::vector<uint32_t> h((size_t)ids.size() * W);
      HIP_TRY(hipMemcpy(h.data(), seg_dbg, h.size() * 4, hipMemcpyDeviceToHost));
      if (FILE* f = fopen(GENV("G2S_SEG_DUMP"), "a")) {
        const uint32_t sb0 = 8u + 2u * ecap;
        for (size_t x = 0; x < ids.size(); x++) {
          const uint32_t* o = h.data() + x * W;
          fprintf(f, "gap %u nA %u nseg %u flags %#x roundsA %u roundsB %u c_count %u best %u\n", o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
          if (o[W - 4] | o[W - 3] | o[W - 2] | o[W - 1])  // (-DG2S_SEG_PROFILE builds) cycles of phase B's sections
            fprintf(f, "P %u %u %u %u %u %u %u %u %u %u\n", o[W - 4], o[W - 3], o[W - 2], o[W - 1], o[W - 8], o[W - 7], o[W - 6], o[W - 5],
                    o[W - 10], o[W - 9]);
          if (o[W - 14] | o[W - 13] | o[W - 12] | o[W - 11])  // (the same builds, one wave per gap) cycles of phase A's sections
            fprintf(f, "PA %u %u %u %u\n", o[W - 14], o[W - 13], o[W - 12], o[W - 11]);
          if (GENV("G2S_SEG_DUMP_BRIEF")) continue;  // (profiles: no entries, no segments)
          for (uint32_t e = 0; e < o[1] && e < ecap; e++) fprintf(f, "A %u %u\n", o[8 + 2 * e], o[9 + 2 * e]);
          for (uint32_t q = 0; q < o[2] && q < scap; q++)
            fprintf(f, "S %u %u %u %u %#x %#x %u\n", o[sb0 + 6 * q], o[sb0 + 6 * q + 1] & 0xFFFF, o[sb0 + 6 * q + 1] >> 16,
                    o[sb0 + 6 * q + 2], o[sb0 + 6 * q + 3], o[sb0 + 6 * q + 4], o[sb0 + 6 * q + 5]);
        }
        fclose(f);
      }
    }
    {  // resets for the next launch, off its critical path (nobody waits for them here)
      HIP_TRY(hipMemsetAsync(s->d_outs.p, 0, n * sizeof(GapOut), st));
      s->d_outs.clean = n * sizeof(GapOut);
      HIP_TRY(hipMemsetAsync(s->d_counter.p, 0, 32, st));
      s->d_counter.clean = 32;
      if (xcd_bytes_used) {
        HIP_TRY(hipMemsetAsync(s->d_xcd.p, 0, 64, st));
        HIP_TRY(hipMemsetAsync((char*)s->d_xcd.p + 64, 0xFF, xcd_bytes_used - 64, st));
        s->d_xcd.clean = xcd_bytes_used;
      }
      if (pool_bytes_used) {
        HIP_TRY(hipMemsetAsync(s->d_rspool.p, 0xFF, pool_bytes_used, st));
        s->d_rspool.clean = pool_bytes_used;
      }
    }
    if (GENV("G2S_DEBUG"))
      fprintf(stderr, "[g2s] run_tier: plan+upload+launch %.3f ms, polling/analysis %.3f ms, final sync %.3f ms\n",
              std::chrono::duration<double, std::milli>(t_launched - t_enter).count(),
              std::chrono::duration<double, std::milli>(t_polled - t_launched).count(),
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_polled).count());
  } else {
    // device -> host: per-gap results, then the packed closures
    HIP_TRY(td->outs.ensure(n * sizeof(GapOut)));
    unsigned long long total_sub = 0;
    HIP_TRY(hipMemcpyAsync(td->outs.p, s->d_outs.p, n * sizeof(GapOut), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&total_sub, s->d_counter.p, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(td->subs.ensure(std::max<uint64_t>(total_sub * sizeof(SubState), 16)));
    if (total_sub)
      HIP_TRY(hipMemcpyAsync(td->subs.p, s->d_subout.p, total_sub * sizeof(SubState), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    b->timing.ms_d2h += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  progress_note("finished: %zu gaps, lds %d seg %d", ids.size(), (int)lds, seg);
  float ms = 0;
  if (seg == 2) {
    HIP_TRY(hipEventElapsedTime(&ms, s->ev[1], s->ev[2]));
    b->timing.ms_fill_segx += ms;
    b->timing.segx_launches++;
  } else if (seg) {
    HIP_TRY(hipEventElapsedTime(&ms, s->ev[1], s->ev[2]));
    b->timing.ms_fill_seg += ms;
    b->timing.seg_launches++;
    b->timing.seg_timed_launches++;
  } else if (lds) {
    HIP_TRY(hipEventElapsedTime(&ms, s->ev[1], s->ev[2]));
    b->timing.ms_fill_lds += ms;
    b->timing.lds_launches++;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev[2], s->ev[3]));
    b->timing.ms_extract_lds += ms;
  } else {
    HIP_TRY(hipEventElapsedTime(&ms, s->ev[0], s->ev[1]));
    b->timing.ms_right_bfs += ms;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev[1], s->ev[2]));
    b->timing.ms_left_dp += ms;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev[2], s->ev[3]));
    b->timing.ms_extract += ms;
    b->timing.launches_left_dp++;
  }
  return G2S_OK;
}

}  // namespace

namespace {

FillParams fill_params_of(const g2s_session* s) {
  FillParams fp;
  fp.k = s->graph->g->k;
  fp.d_err = s->params.d_err;
  fp.skip_confident = s->params.skip_confident != 0;
  fp.all_paths = s->params.all_paths != 0;
  fp.unique_paths = s->params.unique_paths != 0;
  return fp;
}

// Everything about gap i that does not depend on the gaps before it: D2 + stop-depth
// analysis, the order-independent result fields, and the summary the offset pass reads.
// *r must be zeroed.
// (G2S_DEBUG) what the host analysis of a run was made of
static const bool dbg_analysis_stats = getenv("G2S_DEBUG") != nullptr;
static std::atomic<uint64_t> dbg_ns_seg{0}, dbg_n_seg{0}, dbg_ns_state{0}, dbg_n_state{0}, dbg_states{0};

void analyze_gap(g2s_batch* b, size_t i, const FillParams& fp, g2s_result* r) {
  b->prep[i].reset();  // (the array is recycled from batch to batch and not wiped in between: 10 000 of these are 0.4 ms)
  g2s_batch::GapInfo& gi = b->info[i];
  gi.fixed[0] = gi.fixed[1] = -1;
  gi.n_len = 0; gi.reached_j = 0; gi.kind = 0; gi.filled = 0;
  const GapJob& j = b->jobs[i];
  gi.skip_thr = (int16_t)std::max(-1, std::min(j.skip_if_prev_right_fuz_gt, 32767));
  if (j.bad_flank) { gi.kind = 1; r->flags |= G2S_GAP_BAD_FLANK; return; }
  if (b->mem_exceeded[i]) { gi.kind = 2; r->count = -1; r->flags |= G2S_GAP_MEM_EXCEEDED; return; }
  SubView& v = b->views[i];
  SubPrep& pp = b->prep[i];
  bool analysed = false;
  if (v.segs && (v.out->dflags & G2S_DEVA_ANALYSED) && !b->force_host_d2) {
    // segment tier, phase D2 and the stop depths done on the device: nothing per segment is left to do
    const GapOut& go = *v.out;
    pp.seg_mode = true;
    pp.count = go.c_count;
    pp.phase_d = go.c_count > 0 && go.n_len > 0;  // :1169
    if (pp.phase_d) {
      if (!fp.skip_confident) {
        if (fp.all_paths) pp.count = go.count_s;  // the recount (:1189-1191)
        pp.sub[0] = go.sub_vertices; pp.sub[1] = go.sub_edges; pp.sub[2] = 0; pp.sub[3] = 0;
        pp.sub[4] = go.sub_vertices; pp.sub[5] = go.sub_edges;
      }
      pp.sink_safe = (go.dflags & G2S_DEVA_SINK_SAFE) != 0;
      pp.has_choice = (go.dflags & G2S_DEVA_CHOICE) != 0;
      for (int q = 0; q < 2; q++) {
        const uint32_t sg = (go.start_seg >> (16 * q)) & 0xFFFFu;
        pp.start_seg[q] = sg == 0xFFFFu ? -1 : (int)sg;
        pp.start_t[q] = (int)((go.start_t >> (16 * q)) & 0xFFFFu);
        const int fd = go.fixed_draws[q];
        pp.stop_depth[q] = (q < go.n_len && fd >= 0) ? go.len[q] + 1 - fd : -1;
      }
      // the draw count depends on the draws (~4 % of the gaps): the in-order pass will walk the parent links;
      // the stop depths behind every segment, computed here in parallel, let that walk end early
      if (b->seg_td && ((go.n_len > 0 && go.fixed_draws[0] < 0) || (go.n_len > 1 && go.fixed_draws[1] < 0))) {
        const size_t need = ((size_t)v.n_segs * 8 + 15) / 16 + 1;
        const size_t at = b->seg_td->exp_cursor.fetch_add(need);
        if (at + need <= b->seg_td->exp.size()) {
          int32_t* st = (int32_t*)(b->seg_td->exp.data() + at);
          seg_stop_depths(v, st);
          pp.stop = st;
        }
      }
    }
    analysed = true;
  } else if (v.segs) {  // segment tier: the analysis runs on the closure segments themselves, O(segments) ...
    void* scratch = nullptr;
    if (b->seg_td) {  // 24 bytes per segment from the launch's shared buffer (no allocation per gap)
      const size_t need = ((size_t)v.n_segs * 24 + 15) / 16 + 1;
      const size_t at = b->seg_td->exp_cursor.fetch_add(need);
      if (at + need <= b->seg_td->exp.size()) scratch = b->seg_td->exp.data() + at;
    }
    const auto ta0 = std::chrono::steady_clock::now();
    analysed = seg_analyze(fp, j, v, &pp, scratch);
    if (!analysed) pp = SubPrep();
    if (dbg_analysis_stats) {
      dbg_ns_seg.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - ta0).count());
      dbg_n_seg.fetch_add(1);
    }
  }
  if (v.segs && !analysed) {  // ... unless a k-mer occurs at two depths of the closure: per-state records then
    const GapOut& go = *v.out;
    const size_t need = (size_t)go.n_sub + ((size_t)go.n_xp + 1) / 2;
    SubRec* dst = nullptr;
    if (b->seg_td) {
      const size_t at = b->seg_td->exp_cursor.fetch_add(need);
      if (at + need <= b->seg_td->exp.size()) dst = b->seg_td->exp.data() + at;
    }
    if (!dst) { pp.own.resize(need); dst = pp.own.data(); }  // (the shared buffer is sized for 4 states per DP level and gap)
    uint64_t* xp = (uint64_t*)(dst + go.n_sub);
    seg_expand(fp, j, go, v.segs, v.n_segs, dst, xp);
    v.st = dst; v.n = go.n_sub; v.xp = xp; v.n_xp = go.n_xp;
    v.segs = nullptr;
  }
  if (!analysed) {
    const auto ta0 = std::chrono::steady_clock::now();
    if (v.n_xp > 1) std::sort(const_cast<uint64_t*>(v.xp), const_cast<uint64_t*>(v.xp) + v.n_xp);  // by state (the kernel appends per level)
    sub_analyze(fp, j, v, &pp);
    if (dbg_analysis_stats) {
      dbg_ns_state.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - ta0).count());
      dbg_n_state.fetch_add(1);
      dbg_states.fetch_add(v.n);
    }
  }
  r->phaseC_count = v.out->c_count;
  r->n_lengths = v.out->n_len;
  r->lengths[0] = v.out->len[0];
  r->lengths[1] = v.out->len[1];
  if (v.out->flags & (G2S_DEV_Q7_A | G2S_DEV_Q7_B | G2S_DEV_Q7_D)) r->flags |= G2S_GAP_Q7;
  r->flags |= pp.flags;
  r->count = pp.count;
  gi.filled = pp.count > 0 && (!fp.unique_paths || pp.count == 1);
  if (pp.phase_d) {
    r->vertices = pp.sub[0]; r->edges = pp.sub[1]; r->nontrivial_components = pp.sub[2];
    r->size_nontrivial_components = pp.sub[3]; r->vertices_final = pp.sub[4]; r->edges_final = pp.sub[5];
    gi.n_len = (int16_t)v.out->n_len;
    gi.reached_j = (int16_t)v.out->reached_j;
    for (int q = 0; q < v.out->n_len && q < 2; q++) gi.fixed[q] = sub_fixed_draws(v, pp, q);
    // One path length and no state with a second parent: every rand() % 1 of the traceback is
    // 0, so the fill does not depend on where the gap's draws start in the stream.  Write it
    // now (under the kernels); the in-order pass only adds up the draw count.
    if (b->arena && v.out->n_len == 1 && (pp.seg_mode ? !pp.has_choice : v.n_xp == 0) && gi.fixed[0] >= 0) {
      if (pp.seg_mode) seg_traceback(*b->s->graph->g, fp, j, v, pp, nullptr, b->arena + b->arena_off[i], r);
      else sub_traceback(*b->s->graph->g, fp, j, v, pp, nullptr, b->arena + b->arena_off[i], r);
      if (r->draws != gi.fixed[0]) r->flags |= G2S_GAP_BACKTRACE_FAIL;  // cannot happen: the draw count was proven fixed
      r->fill_off = (uint64_t)(b->arena_base + b->arena_off[i]) + (uint64_t)(j.lmf - r->left_fuz);
      r->fill_len = (int32_t)strlen(b->arena + b->arena_off[i] + (j.lmf - r->left_fuz));
      gi.filled |= 2;
    }
  }
}

// Stage 1 of a batch: phases A-D1 on the GPU (all passes/tiers) and, on request, the per-gap
// host analysis (D2, stop depths).  Independent of every other batch.
int batch_stage1(g2s_batch* b, bool analyze, g2s_result* results) {
  g2s_session* s = b->s;
  if (hipSetDevice(s->device) != hipSuccess) return fail(G2S_ERR_NO_DEVICE, "cannot select device");
  const Graph& g = *s->graph->g;
  const size_t n = b->jobs.size();
  auto t_begin = std::chrono::steady_clock::now();
  const bool dbg_laps = GENV("G2S_DEBUG") != nullptr;
  auto t_lap = t_begin;
  auto lap = [&](const char* what) {  // (diagnostics) where stage 1 spends its time outside the launches
    if (!dbg_laps) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[g2s] stage 1 lap %s: %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_lap).count());
    t_lap = now;
  };
  { const int rc = b->upload_flanks(); if (rc != G2S_OK) return rc; }
  { const int rc = b->fetch_nodes(); if (rc != G2S_OK) return rc; }  // (a list resident mode gave back: the kernel kept most node ids on the device)
  b->drop_tiers();
  b->force_host_d2 = GENV("G2S_HOST_D2") != nullptr;
  if (GENV("G2S_DEBUG")) fprintf(stderr, "[g2s] stage 1 begins\n");
  lap("flank upload");
  g2s_timing keep = b->timing;
  memset(&b->timing, 0, sizeof b->timing);
  b->timing.flank_bytes = keep.flank_bytes;
  b->timing.ms_prepare = keep.ms_prepare;
  b->timing.resident_fallbacks = keep.resident_fallbacks;

  const FillParams fp = fill_params_of(s);
  // device-budget analogue of -max-mem: states a gap may hold
  const uint64_t max_states = (uint64_t)std::max<int64_t>(s->params.max_mem, 1 << 16) / 64;

  std::vector<SubView>& views = b->views;
  std::vector<char>& mem_exceeded = b->mem_exceeded;
  if (views.capacity() < n && s->spare_views.capacity() >= n) views.swap(s->spare_views);
  if (b->prep.capacity() < n && s->spare_prep.capacity() >= n) b->prep.swap(s->spare_prep);
  views.assign(n, SubView());
  mem_exceeded.assign(n, 0);
  // analysis of finished gaps runs on the pool while the kernel is still busy with the rest
  std::vector<char> analyzed(n, 0);
  std::vector<uint32_t> fresh;
  double ms_stream = 0;
  if (analyze) {
    b->prep.resize(n);  // (every gap passes through analyze_gap once, which resets its element)
    b->info.assign(n, g2s_batch::GapInfo());
  }
  lap("per-gap arrays");
  TierData* td_live = nullptr;
  int seg_mode_live = 0;               // 1 / 2 while a launch of the segment tier (its large variant) is polled
  std::vector<char> seg_accounted(n, 0);  // 1: counted by on_done, 2: counted, over the -max-mem analogue
  std::vector<uint32_t> heavy_wait, heavy_job;
  std::function<void(size_t)> heavy_fn = [&](size_t t) { analyze_gap(b, heavy_job[t], fp, &results[heavy_job[t]]); };
  auto post_heavy = [&]() {  // the pool is free: what has piled up becomes its next job (largest closures first)
    s->pool->finish();
    heavy_job.swap(heavy_wait);
    heavy_wait.clear();
    std::sort(heavy_job.begin(), heavy_job.end(), [&](uint32_t a, uint32_t c) { return views[a].out->n_sub > views[c].out->n_sub; });
    s->pool->post(heavy_job.size(), heavy_fn);
  };
  auto flush_heavy = [&]() {  // before anything else uses the pool, and before the views of a launch are consumed
    while (true) {
      s->pool->finish();
      if (heavy_wait.empty()) break;
      post_heavy();
    }
  };
  struct DoneSums { uint64_t xA = 0, sA = 0, xB = 0, sB = 0, xD = 0, sD = 0, segs = 0; uint32_t seg_gaps = 0, segx_gaps = 0; };
  // a finished gap's view of the launch's buffers and its share of the launch's sums; 0: nothing to analyse (seen
  // before, or it runs again in a later pass), 1: analyse now, 2: a large closure for the host, analysed without waiting
  auto admit = [&](uint32_t i, DoneSums& acc) -> int {
    if (i >= n || analyzed[i]) return 0;
    const GapOut* outs = (const GapOut*)td_live->outs.p;
    const GapOut& go = outs[i];
    if (go.flags & (G2S_DEV_OVERFLOW_A | G2S_DEV_OVERFLOW_B)) return 0;  // runs again in a later pass
    if ((go.flags & G2S_DEV_COMPACT) && ((uint64_t)go.n_states > max_states || (uint64_t)go.n_right > max_states))
      mem_exceeded[i] = 1;  // segment tier: the -max-mem analogue is applied to the state count
    SubView& v = views[i];
    v.out = &go;
    if (go.flags & G2S_DEV_COMPACT) {
      v.segs = (const SegRec*)((const SubRec*)td_live->subs.p + go.sub_off);
      v.n_segs = go.n_xl;
      v.st = nullptr; v.n = 0; v.xp = nullptr; v.n_xp = 0;
    } else {
      v.st = (const SubRec*)td_live->subs.p + go.sub_off;
      v.n = go.n_sub;
      v.xp = (const uint64_t*)(v.st + go.n_sub);
      v.n_xp = go.n_xp;
    }
    analyzed[i] = 1;
    if (seg_mode_live) {  // the launch's bookkeeping while the gap's record is in this core's cache
      const bool mx = (uint64_t)go.n_states > max_states || (uint64_t)go.n_right > max_states;
      seg_accounted[i] = mx ? 2 : 1;
      if (!mx) {
        acc.xA += go.x_right; acc.sA += go.n_right;
        acc.xB += go.x_left; acc.sB += go.n_states;
        acc.xD += go.x_sub; acc.sD += go.n_sub;
        if (seg_mode_live == 2) acc.segx_gaps++; else acc.seg_gaps++;
        acc.segs += go.stat[3];
      }
    }
    // Closures of many thousand states that the host has to analyse (the large variant's gaps, a k-mer at two
    // depths: milliseconds each) go to the pool WITHOUT waiting for them: this thread keeps polling, and
    // whatever has arrived by the time the pool is free again forms the next job.
    const bool heavy = v.segs && !(go.dflags & G2S_DEVA_ANALYSED) && go.n_sub >= 4000;
    return heavy ? 2 : 1;
  };
  auto add_sums = [&](const DoneSums& acc) {
    b->timing.xA += acc.xA; b->timing.sA += acc.sA;
    b->timing.xB += acc.xB; b->timing.sB += acc.sB;
    b->timing.xD += acc.xD; b->timing.sD += acc.sD;
    b->timing.seg_tier_gaps += acc.seg_gaps; b->timing.segx_tier_gaps += acc.segx_gaps;
    b->timing.seg_segments += acc.segs;
  };
  std::mutex done_mu;
  const DoneFn on_done = [&](const uint32_t* done_ids, size_t cnt) {
    auto t0 = std::chrono::steady_clock::now();
    if (cnt >= 512) {
      // A hand-over of a long list: the finished gaps' records were written by the GPU and are in no core's cache,
      // so even the set-up above is a memory round trip or two per gap (10 000 gaps: 1 ms on this thread, twice the
      // kernel's time) — all of it goes to the pool, a few gaps per task.
      s->pool->finish();  // (a job of heavy closures may still be open)
      const size_t per = 16, nt = (cnt + per - 1) / per;
      s->pool->run(nt, [&](size_t t) {
        DoneSums acc;
        uint32_t heavy_here[16];
        size_t nh = 0;
        const size_t lo = t * per, hi = std::min(cnt, lo + per);
        const GapOut* outs = (const GapOut*)td_live->outs.p;
        for (size_t x = lo; x < hi; x++)
          if (done_ids[x] < n) __builtin_prefetch(&outs[done_ids[x]]);
        for (size_t x = lo; x < hi; x++) {
          const uint32_t i = done_ids[x];
          const int c = admit(i, acc);
          // (a closure of a few thousand states takes 20-40 us on the host: nothing to set aside when the whole
          // pool is at work on this hand-over; the closures of the large variant's gaps still are)
          if (c == 1 || (c == 2 && views[i].out->n_sub < 50000u)) analyze_gap(b, i, fp, &results[i]);
          else if (c == 2) heavy_here[nh++] = i;
        }
        std::lock_guard<std::mutex> lk(done_mu);
        add_sums(acc);
        for (size_t h = 0; h < nh; h++) heavy_wait.push_back(heavy_here[h]);
      });
      if (!heavy_wait.empty() && s->pool->idle()) post_heavy();
      ms_stream += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      return;
    }
    fresh.clear();
    {
      DoneSums acc;
      for (size_t x = 0; x < cnt; x++) {
        const uint32_t i = done_ids[x];
        const int c = admit(i, acc);
        if (c == 1) fresh.push_back(i);
        else if (c == 2) heavy_wait.push_back(i);
      }
      add_sums(acc);
    }
    size_t work = 0;
    for (uint32_t i : fresh)  // (closure states to look at; 2 for a gap analysed on the device)
      work += views[i].segs ? ((views[i].out->dflags & G2S_DEVA_ANALYSED) ? 2u : views[i].out->n_sub) : views[i].n;
    if (fresh.size() <= 1 || work < 1500) {  // not worth waking the pool (~20 ns per closure state)
      for (uint32_t i : fresh) analyze_gap(b, i, fp, &results[i]);
    } else {
      // the last gaps of a launch are few and large: one task per gap then
      s->pool->finish();  // (a job of heavy closures may still be open)
      const size_t per = fresh.size() >= 64 ? 8 : 1, nt = (fresh.size() + per - 1) / per;
      s->pool->run(nt, [&](size_t t) {
        for (size_t x = t * per; x < std::min(fresh.size(), (t + 1) * per); x++) analyze_gap(b, fresh[x], fp, &results[fresh[x]]);
      });
    }
    if (!heavy_wait.empty() && s->pool->idle()) post_heavy();
    ms_stream += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };

  // ---- GPU: phases A-D1, retrying gaps whose tables overflowed with 8x larger ones
  std::vector<uint32_t> todo, lds_ids, late1, late2;  // late1/late2: gaps that start in pass 1 / pass 2
  // (the LDS right set keeps 28-bit k-mer indices)
  const bool lds_ok = s->graph->g->dev.at(s->device).pred == nullptr && !s->no_lds_tier &&
                      s->graph->g->n < (1ull << 28) - 1;
  // ---- segment tier first (fill_seg.hip): every gap it can hold (odd k, -fuz <= 31, D < 2^15);
  // a gap that outgrows one of its capacities comes back flagged and takes the passes below
  std::vector<char> seg_done(n, 0);
  const bool seg_ok = lds_ok && s->graph->g->dev.at(s->device).rem != nullptr && !GENV("G2S_NO_SEG_TIER");
  if (seg_ok) {
    std::vector<uint32_t> seg_ids;
    for (size_t i = 0; i < n; i++) {
      const GapJob& j = b->jobs[i];
      if (j.bad_flank || j.rmf > 31 || j.lmf > 31 || j.lmf + j.rmf + j.g + fp.d_err >= 32767) continue;
      seg_ids.push_back((uint32_t)i);
    }
    // mode 1: the tier proper; mode 2: its large variant (g2s_fill_segx) for the gaps that outgrew a capacity of
    // mode 1 (-dist-error 2000: thousands of segments and right-set entries); what outgrows that too takes
    // the passes below
    // (tests: G2S_FORCE_SEGX=1 sends every gap to the large variant, G2S_NO_SEGX_TIER=1 none)
    for (int mode = GENV("G2S_FORCE_SEGX") ? 2 : 1; mode <= 2 && !seg_ids.empty(); mode++) {
      if (mode == 2 && GENV("G2S_NO_SEGX_TIER")) break;
      // longest searches first (see below): always for the large variant, whose workgroups take the list in order
      if ((seg_ids.size() > 1024 || mode == 2) && !GENV("G2S_NO_LPT")) {
        // (a stable counting sort by gap length, longest first: a comparison sort of 10 000 ids cost 0.3 ms)
        int gmax = 0;
        for (uint32_t i : seg_ids) gmax = std::max(gmax, b->jobs[i].g);
        if ((size_t)gmax <= 8 * seg_ids.size() + 65536) {
          std::vector<uint32_t> at((size_t)gmax + 2, 0), sorted(seg_ids.size());
          for (uint32_t i : seg_ids) at[(size_t)(gmax - b->jobs[i].g) + 1]++;
          for (size_t x = 1; x < at.size(); x++) at[x] += at[x - 1];
          for (uint32_t i : seg_ids) sorted[at[(size_t)(gmax - b->jobs[i].g)]++] = i;
          seg_ids.swap(sorted);
        } else {
          std::stable_sort(seg_ids.begin(), seg_ids.end(), [&](uint32_t a, uint32_t c) { return b->jobs[a].g > b->jobs[c].g; });
        }
      }
      TierData* td = take_tier(s, b->tiers.size());
      b->tiers.push_back(td);
      td_live = td;
      b->seg_td = td;
      {  // scratch of the per-gap analysis (24 bytes per closure segment; the rare closure with a k-mer at
         // two depths is expanded into per-state records here too): 1 KB per gap, per-gap buffers beyond
        // (the large variant's closures: a few thousand segments; a gap that finds the buffer used up allocates its own)
        const size_t want = std::min<size_t>(seg_ids.size() * (mode == 2 ? 2048u : 64u), (size_t)16 << 20) + 4096u;
        if (td->exp.size() < want) td->exp.resize(want);
        td->exp_cursor.store(0);
      }
      lap("segment pass set-up");
      seg_mode_live = mode;
      int rc = run_tier(b, seg_ids, 1, max_states, td, true, 0, false, 64u, analyze ? &on_done : nullptr, mode);
      seg_mode_live = 0;
      flush_heavy();
      if (rc != G2S_OK) return rc;
      lap("segment pass run_tier");
      const GapOut* outs = (const GapOut*)td->outs.p;
      std::vector<uint32_t> left;
      for (uint32_t i : seg_ids) {
        if (seg_accounted[i]) {  // (on_done saw the gap done and did the sums: its record is not read again here)
          seg_done[i] = 1;
          if (seg_accounted[i] == 2) mem_exceeded[i] = 1;
          seg_accounted[i] = 0;
          continue;
        }
        const GapOut& go = outs[i];
        if (go.flags & (G2S_DEV_OVERFLOW_A | G2S_DEV_OVERFLOW_B)) {
          if (GENV("G2S_DEBUG"))
            fprintf(stderr, "[g2s] gap %u left the segment tier (mode %d): flags 0x%x entries %u segments %u g %d\n", i, mode,
                    go.flags, go.stat[1], go.stat[3], b->jobs[i].g);
          if (go.flags & G2S_DEV_WATCHDOG) {  // a defect, never expected: say so, the gap is filled by the LDS tier
            b->timing.watchdog_gaps++;
            fprintf(stderr, "[g2s] segment tier (mode %d): a probe loop ran past its bound on gap %u (flags 0x%x); the gap runs in the LDS tier\n",
                    mode, i, go.flags);
          }
          left.push_back(i);
          continue;
        }
        seg_done[i] = 1;
        // device-budget analogue of -max-mem: the states a gap may hold
        if ((uint64_t)go.n_states > max_states || (uint64_t)go.n_right > max_states) { mem_exceeded[i] = 1; continue; }
        if (!analyzed[i]) {  // (gaps analysed while the kernel ran already have their view, expanded)
          SubView& v = views[i];
          v.out = &go;
          v.segs = (const SegRec*)((const SubRec*)td->subs.p + go.sub_off);
          v.n_segs = go.n_xl;
          v.st = nullptr; v.n = 0; v.xp = nullptr; v.n_xp = 0;
        }
        b->timing.xA += go.x_right; b->timing.sA += go.n_right;
        b->timing.xB += go.x_left; b->timing.sB += go.n_states;
        b->timing.xD += go.x_sub; b->timing.sD += go.n_sub;
        if (mode == 2) b->timing.segx_tier_gaps++; else b->timing.seg_tier_gaps++;
        b->timing.seg_segments += go.stat[3];
      }
      if (const char* dump = GENV("G2S_DUMP_STATS")) {
        if (FILE* f = fopen(dump, "a")) {
          fprintf(f, "# segment tier%s: gap g flags A_rounds A_entries B_rounds segments cycA cycB 0 0 cycD n_right x_right n_states x_left n_sub 0\n",
                  mode == 2 ? " (large variant)" : "");
          for (uint32_t i : seg_ids) {
            const GapOut& o = outs[i];
            fprintf(f, "%u %d %#x %u %u %u %u %llu %llu %u %u %llu %u %u %u %u %u %u\n", i, b->jobs[i].g, o.flags, o.stat[0], o.stat[1],
                    o.stat[2], o.stat[3], (unsigned long long)o.stat[4] << 8, (unsigned long long)o.stat[5] << 8, 0u, 0u,
                    (unsigned long long)o.stat[7] << 8, o.n_right, o.x_right, o.n_states, o.x_left, o.n_sub, 0u);
          }
          fclose(f);
        }
      }
      if (GENV("G2S_DEBUG")) {
        std::vector<uint32_t> ord(seg_ids);
        std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t c) {
          return outs[a].stat[4] + outs[a].stat[5] + outs[a].stat[7] > outs[c].stat[4] + outs[c].stat[5] + outs[c].stat[7]; });
        for (size_t q = 0; q < ord.size() && q < 4; q++) {
          const GapOut& o = outs[ord[q]];
          fprintf(stderr, "[g2s] segment tier (mode %d) slow gap %u: g %d | A rounds %u entries %u kcyc %u | B rounds %u segments %u kcyc %u | D1+emit kcyc %u | states %u closure %u flags %#x count %d\n",
                  mode, ord[q], b->jobs[ord[q]].g, o.stat[0], o.stat[1], o.stat[4] >> 2, o.stat[2], o.stat[3], o.stat[5] >> 2, o.stat[7] >> 2,
                  o.n_states, o.n_sub, o.flags, o.c_count);
        }
      }
      seg_ids.swap(left);
    }
  }
  {
    size_t nvalid = 0;
    for (size_t i = 0; i < n; i++) nvalid += !b->jobs[i].bad_flank && !seg_done[i];
    const uint32_t room = lds_room(std::max<size_t>(1, nvalid));
    for (size_t i = 0; i < n; i++) {
      if (b->jobs[i].bad_flank || seg_done[i]) continue;
      if (lds_ok && lds_rs_cap(b->jobs[i], fp.d_err, room) != 0) lds_ids.push_back((uint32_t)i);
      else if (lds_ok && lds_rs_cap(b->jobs[i], fp.d_err, 16384u) != 0) late1.push_back((uint32_t)i);   // pass 1's table
      else if (lds_ok && b->jobs[i].rmf <= (int)fill_lds_max_fuz()) late2.push_back((uint32_t)i);        // right set in HBM
      else todo.push_back((uint32_t)i);
    }
  }
  // ---- tier 0: LDS-resident kernels.  Pass 1 shares the CU's LDS among as many gaps as
  // the batch needs resident; gaps whose right set outgrows that share are run again with
  // the largest LDS tables (pass 2, few gaps per CU); what still does not fit (frontier
  // > 64, > 128 target hits, state log) falls through to the HBM tier.
  //   pass 0: LDS shared among all gaps of the batch, state log 8 (D+2)
  //   pass 1: largest LDS right set (16 K entries), state log 64 (D+2)
  //   pass 2: right set in HBM (any size up to -max-mem), the rest still in LDS
  std::vector<uint32_t> cand[3];
  cand[0] = lds_ids;
  cand[1] = late1;
  cand[2] = late2;
  const bool room0_is_max = lds_room(n) >= 16384u;
  for (int pass = 0; pass < 3; pass++) {
    if (cand[pass].empty()) continue;
    // Workgroups are dispatched in id order and a launch ends with its slowest gap: start the
    // gaps with the most DP levels first, so that the long ones are not the last to begin
    // (lists longer than the chip holds at once; G2S_NO_LPT=1 keeps the input order).
    if (cand[pass].size() > 1024 && !GENV("G2S_NO_LPT"))
      std::stable_sort(cand[pass].begin(), cand[pass].end(),
                       [&](uint32_t a, uint32_t c) { return b->jobs[a].g > b->jobs[c].g; });
    const std::vector<uint32_t>& ids = cand[pass];
    const uint32_t room = pass == 0 ? 0u : 16384u;
    TierData* td = take_tier(s, b->tiers.size());
    b->tiers.push_back(td);
    // pass 0 keeps the LDS footprint small (frontier 64); later passes run few gaps per CU and
    // take wide frontiers (1024 entries) so that repeat-rich gaps stay out of the HBM tier
    td_live = td;
    int rc = run_tier(b, ids, pass == 0 ? 1 : 8, max_states, td, true, room, pass == 2, pass == 0 ? 64u : 1024u,
                      analyze ? &on_done : nullptr);
    if (rc != G2S_OK) return rc;
    const GapOut* outs = (const GapOut*)td->outs.p;
    for (uint32_t i : ids) {
      const GapOut& go = outs[i];
      if (go.flags & (G2S_DEV_OVERFLOW_A | G2S_DEV_OVERFLOW_B)) {
        if (GENV("G2S_DEBUG"))
          fprintf(stderr, "[g2s] gap %u left LDS pass %d: flags 0x%x n_right %u x_right %u n_states %u x_left %u final_d %d g %d\n",
                  i, pass, go.flags, go.n_right, go.x_right, go.n_states, go.x_left, go.final_d, b->jobs[i].g);
        // a right-set overflow is cured by a larger right set (pass 1 unless pass 0 already
        // had the largest LDS table, else pass 2); state-log and frontier overflows by pass 1's
        // larger log and 1024-entry frontier; what still overflows goes to the HBM tier
        const bool only_a = (go.flags & G2S_DEV_OVERFLOW_A) && !(go.flags & G2S_DEV_OVERFLOW_B);
        int target = 3;
        // deep searches (-dist-error in the thousands) overflow the largest LDS right set too
        const bool deep = b->jobs[i].rmf + (b->jobs[i].g + fp.d_err + 1) / 2 > 2048;
        if (pass == 0) target = (only_a && (room0_is_max || deep)) ? 2 : 1;
        else if (pass == 1 && only_a) target = 2;
        if (target < 3) cand[target].push_back(i);
        else { todo.push_back(i); b->timing.retried_gaps++; }
        continue;
      }
      SubView& v = views[i];
      v.out = &go;
      v.st = (const SubRec*)td->subs.p + go.sub_off;
      v.n = go.n_sub;
      v.xp = (const uint64_t*)(v.st + go.n_sub);
      v.n_xp = go.n_xp;
      b->timing.xA += go.x_right; b->timing.sA += go.n_right;
      b->timing.xB += go.x_left; b->timing.sB += go.n_states;
      b->timing.xD += go.x_sub; b->timing.sD += go.n_sub;
      b->timing.x_fill_lds += (uint64_t)go.x_right + go.x_left;
      b->timing.s_fill_lds += (uint64_t)go.n_right + go.n_states;
      b->timing.lds_tier_gaps++;
      b->timing.log_pool_gaps += (go.flags & G2S_DEV_LOG_POOL) != 0;
      b->timing.rs_pool_gaps += (go.flags & G2S_DEV_RS_POOL) != 0;
    }
    if (const char* dump = GENV("G2S_DUMP_STATS")) {  // diagnostics: one line per gap of this pass (appended)
      if (FILE* f = fopen(dump, "a")) {
        fprintf(f, "# pass %d: gap g flags A_steps A_rounds B_slow B_bulk cycA cycB D_slow D_bulk cycD n_right x_right n_states x_left n_sub top_level\n", pass);
        for (uint32_t i : ids) {
          const GapOut& o = outs[i];
          fprintf(f, "%u %d %#x %u %u %u %u %llu %llu %u %u %llu %u %u %u %u %u %u\n", i, b->jobs[i].g, o.flags, o.stat[0], o.stat[1],
                  o.stat[2], o.stat[3], (unsigned long long)o.stat[4] << 8, (unsigned long long)o.stat[5] << 8, o.stat[6] & 0xFFFF,
                  o.stat[6] >> 16, (unsigned long long)o.stat[7] << 8, o.n_right, o.x_right, o.n_states, o.x_left, o.n_sub, o.top_level);
        }
        fclose(f);
      }
    }
    if (GENV("G2S_DEBUG")) {  // the slowest gaps of the LDS tier and why
      std::vector<uint32_t> ord(ids);
      std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t c) {
        return outs[a].stat[4] + outs[a].stat[5] + outs[a].stat[7] > outs[c].stat[4] + outs[c].stat[5] + outs[c].stat[7]; });
      for (size_t q = 0; q < ord.size() && q < 4; q++) {
        const GapOut& o = outs[ord[q]];
        fprintf(stderr, "[g2s] pass %d slow gap %u: g %d | A per-level %u bulk %u kcyc %u | B per-level %u bulk %u kcyc %u | D1 per-level %u bulk %u kcyc %u | xA %u xB %u xD %u | right set %u states %u final_d %d top level %u flags %#x count %d\n",
                pass, ord[q], b->jobs[ord[q]].g, o.stat[0], o.stat[1], o.stat[4] >> 2, o.stat[2], o.stat[3], o.stat[5] >> 2,
                o.stat[6] & 0xFFFF, o.stat[6] >> 16, o.stat[7] >> 2, o.x_right, o.x_left, o.x_sub, o.n_right, o.n_states,
                o.final_d, o.top_level, o.flags, o.c_count);
      }
    }
  }
  std::sort(todo.begin(), todo.end());
  // gaps that outgrew the LDS tier are known to branch: start them with 8x tables
  uint64_t scale = (lds_ids.empty() && late1.empty() && late2.empty()) ? 1 : 8;
  while (!todo.empty()) {
    std::vector<uint32_t> next_todo;
    size_t pos = 0;
    while (pos < todo.size()) {  // groups that fit the session's HBM budget
      std::vector<uint32_t> group;
      uint64_t bytes = 0;
      while (pos < todo.size()) {
        const Plan p = plan_gap(b->jobs[todo[pos]], fp.d_err, scale, max_states);
        if (!group.empty() && bytes + p.bytes > s->mem_budget) break;
        bytes += p.bytes;
        group.push_back(todo[pos++]);
      }
      TierData* td = take_tier(s, b->tiers.size());
      b->tiers.push_back(td);
      int rc = run_tier(b, group, scale, max_states, td, false);
      if (rc != G2S_OK) return rc;
      const GapOut* outs = (const GapOut*)td->outs.p;
      // this tier's kernel emits 32-byte records with the parents by GATB slot: convert them
      // to the host's 16-byte records + side lists (sizes first: the views point into the vectors)
      {
        size_t nrec = 0, nxp = 0;
        for (uint32_t i : group) {
          const GapOut& go = outs[i];
          if (go.flags & (G2S_DEV_OVERFLOW_A | G2S_DEV_OVERFLOW_B)) continue;
          const SubState* in = (const SubState*)td->subs.p + go.sub_off;
          nrec += go.n_sub;
          for (uint32_t q = 0; q < go.n_sub; q++) {
            int np = 0;
            for (int nt = 0; nt < 4; nt++) np += in[q].pred[nt] >= 0;
            if (np > 1) nxp += (size_t)np - 1;
          }
        }
        td->conv.clear(); td->conv.reserve(nrec);
        td->conv_xp.clear(); td->conv_xp.reserve(nxp);
      }
      for (size_t x = 0; x < group.size(); x++) {
        const uint32_t i = group[x];
        const GapOut& go = outs[i];
        if (go.flags & (G2S_DEV_OVERFLOW_A | G2S_DEV_OVERFLOW_B)) {
          const Plan p = plan_gap(b->jobs[i], fp.d_err, scale, max_states);
          const Plan p8 = plan_gap(b->jobs[i], fp.d_err, scale * 8, max_states);
          if (p8.bytes > s->mem_budget || (p8.rlog_cap == p.rlog_cap && p8.slog_cap == p.slog_cap))
            mem_exceeded[i] = 1;  // cannot grow further: -max-mem verdict
          else
            next_todo.push_back(i);
          continue;
        }
        SubView& v = views[i];
        v.out = &go;
        {
          const size_t r0 = td->conv.size(), x0 = td->conv_xp.size();
          sub_convert((const SubState*)td->subs.p + go.sub_off, go.n_sub, &td->conv, &td->conv_xp);
          v.st = td->conv.data() + r0;
          v.n = go.n_sub;
          v.xp = td->conv_xp.data() + x0;
          v.n_xp = (uint32_t)(td->conv_xp.size() - x0);
        }
        b->timing.xA += go.x_right; b->timing.sA += go.n_right;
        b->timing.xB += go.x_left; b->timing.sB += go.n_states;
        b->timing.xD += go.x_sub; b->timing.sD += go.n_sub;
      }
    }
    b->timing.retried_gaps += (uint32_t)next_todo.size();
    todo.swap(next_todo);
    scale *= 8;
  }

  b->timing.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  if (analyze) {
    // ---- host: D2 + stop-depth analysis per gap, thread pool (teams: per group, so that it
    // overlaps the other sessions' kernels)
    lap("passes, bookkeeping");
    // what was not analysed while the kernels ran: bad flanks, verdicts, gaps of the HBM tier
    auto t_post = std::chrono::steady_clock::now();
    fresh.clear();
    for (size_t i = 0; i < n; i++) if (!analyzed[i]) fresh.push_back((uint32_t)i);
    const size_t per = fresh.size() >= 64 ? 8 : 1, nt = (fresh.size() + per - 1) / per;
    s->pool->run(nt, [&](size_t t) {
      for (size_t x = t * per; x < std::min(fresh.size(), (t + 1) * per); x++) analyze_gap(b, fresh[x], fp, &results[fresh[x]]);
    });
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_post).count();
    b->timing.ms_host_post = ms + ms_stream;  // ms_stream overlapped the kernels
    b->timing.ms_total += ms;
    if (GENV("G2S_DEBUG")) {
      fprintf(stderr, "[g2s] analysis: %.3f ms while the kernels ran, %.3f ms after (%zu gaps)\n", ms_stream, ms, fresh.size());
      fprintf(stderr, "[g2s] analysis on the host: %llu closures on segments %.3f ms; %llu on per-state records (a k-mer at two depths) %.3f ms, %llu states\n",
              (unsigned long long)dbg_n_seg.exchange(0), dbg_ns_seg.exchange(0) / 1e6, (unsigned long long)dbg_n_state.exchange(0),
              dbg_ns_state.exchange(0) / 1e6, (unsigned long long)dbg_states.exchange(0));
    }
  }
  (void)g;
  (void)views;
  return G2S_OK;
}

// Stage 2 over batches in gap order: assign rand() stream offsets (:178,1440,1513) and run
// the tracebacks.  results/arena are laid out batch after batch.
//   1. (only when `analyze`) per-gap analysis on the pool, chunks of gaps claimed dynamically;
//   2. the in-order pass on the calling thread: O(1) per gap from the 16-byte GapInfo records
//      when the number of draws does not depend on the draws (the others are traced inline),
//      honouring skip_if_prev_right_fuz_gt;
//   3. tracebacks + fill offsets on the pool.
int batches_stage2(const std::vector<g2s_batch*>& bs, g2s_session* lead, g2s_result* results, char* arena,
                   g2s_timing* timing, bool analyze) {
  const Graph& g = *lead->graph->g;
  const FillParams fp = fill_params_of(lead);
  auto t_begin = std::chrono::steady_clock::now();
  size_t n = 0;
  for (g2s_batch* b : bs) n += b->jobs.size();
  if (n == 0) return G2S_OK;
  std::vector<size_t> arena_off(n), rand_off(n, 0);
  std::vector<char> todo_tb(n, 0);
  std::vector<int> expect_draws(n, 0);
  std::vector<g2s_batch*> owner(n);
  std::vector<uint32_t> local(n);
  {
    size_t apos = 0, gi = 0;
    for (g2s_batch* b : bs) {
      if (analyze) { b->prep.resize(b->jobs.size()); b->info.assign(b->jobs.size(), g2s_batch::GapInfo()); }
      for (size_t i = 0; i < b->jobs.size(); i++, gi++) {
        arena_off[gi] = apos;
        apos += b->jobs[i].buf_bytes(g.k, fp.d_err);
        owner[gi] = b;
        local[gi] = (uint32_t)i;
      }
    }
  }
  const size_t per = 8, nchunks = (n + per - 1) / per;
  std::atomic<uint64_t> fill_bytes(0);
  double ms_order = 0, ms_order_loop = 0, ms_ana = 0, ms_walks = 0;
  size_t n_inline = 0, n_two = 0, n_rest = 0;
  bool serial = false;
  size_t draws_used = 0;

  size_t draws_total = 0;
  bool prev_filled = false;
  int prev_right_fuz = 0;
  bool jobs_in_flight = false;  // tracebacks of an earlier segment are reading the rand() buffer
  auto grow_rands = [&](size_t upto) {  // materialise more values; the buffer may move, so nobody may be reading it
    if (!lead->rcache.would_grow(upto)) return;
    if (jobs_in_flight) { lead->pool->finish(); jobs_in_flight = false; }
    lead->rcache.ensure(upto + (1u << 16));
  };
  std::vector<uint32_t> walk_ids;  // gaps of the current segment whose draw count needs a walk
  auto in_order_pass = [&](size_t g_lo, size_t g_hi) {
    auto t0 = std::chrono::steady_clock::now();
    // The walks are the serial part of a run (one depends on the other through the stream offset), and the
    // closure records they chase were written by the GPU: nobody on the host has read them yet.  The records
    // of the gaps a few walks ahead are fetched while the current walk runs.
    walk_ids.clear();
    for (size_t gi = g_lo; gi < g_hi; gi++) {
      const g2s_batch::GapInfo& in = owner[gi]->info[local[gi]];
      if (in.kind == 0 && in.n_len > 0 && (in.fixed[0] < 0 || (in.n_len > 1 && in.fixed[1] < 0))) walk_ids.push_back((uint32_t)(gi - g_lo));
    }
    size_t walk_next = 0;
    auto fetch_ahead = [&](size_t gi) {
      while (walk_next < walk_ids.size() && g_lo + walk_ids[walk_next] <= gi) walk_next++;
      const size_t k = walk_next + 5;
      if (k >= walk_ids.size()) return;
      const size_t gj = g_lo + walk_ids[k];
      const SubView& pv = owner[gj]->views[local[gj]];
      if (!pv.segs) return;
      const char* p0 = (const char*)pv.segs;
      const size_t bytes = std::min<size_t>((size_t)pv.n_segs * sizeof(SegRec), 4096);
      for (size_t o = 0; o < bytes; o += 64) __builtin_prefetch(p0 + o);
    };
    for (size_t gi = g_lo; gi < g_hi; gi++) {
      g2s_batch* b = owner[gi];
      const size_t i = local[gi];
      const g2s_batch::GapInfo& in = b->info[i];
      if (i + 32 < b->info.size()) __builtin_prefetch(&b->info[i + 32]);  // written by other cores
      const int skip_thr = in.skip_thr;
      if (skip_thr >= 0 && prev_filled && prev_right_fuz > skip_thr) {
        g2s_result& r = results[gi];  // the gap is not attempted at all (:369)
        memset(&r, 0, sizeof r);
        r.flags = G2S_GAP_SKIPPED;
        if (in.filled & 2) arena[arena_off[gi] + (size_t)b->jobs[i].lmf] = '\0';
        prev_filled = false;
        continue;
      }
      if (in.kind != 0) { prev_filled = false; continue; }
      int right_fuz = 0;
      if (in.n_len > 0) {
        rand_off[gi] = draws_total;
        if (in.n_len > 1) grow_rands(draws_total + 1);
        const int pick = in.n_len > 1 ? (int)(lead->rcache.at_const(draws_total) % in.n_len) : 0;
        int draws = in.fixed[pick];
        n_two += in.n_len > 1;
        if (draws >= 0 && !(in.filled & 2) && serial) draws = -1;  // few gaps left to trace: no pool round for them
        if (draws >= 0) {
          todo_tb[gi] = !(in.filled & 2);
          right_fuz = in.reached_j;
        } else if (serial) {
          n_inline++;
          g2s_result& r = results[gi];
          const SubView& v = b->views[i];
          grow_rands(draws_total + (size_t)v.out->len[pick] + 2);
          if (b->prep[i].seg_mode) seg_traceback(g, fp, b->jobs[i], v, b->prep[i], lead->rcache.ptr(draws_total), arena + arena_off[gi], &r);
          else sub_traceback(g, fp, b->jobs[i], v, b->prep[i], lead->rcache.ptr(draws_total), arena + arena_off[gi], &r);
          draws = r.draws;
          right_fuz = r.right_fuz;
        } else {
          // the draw count depends on the draws: walk the parent links once for the count,
          // the fill itself is written by the pool with the others
          n_inline++;
          const auto tw0 = std::chrono::steady_clock::now();
          fetch_ahead(gi);
          const SubView& v = b->views[i];
          grow_rands(draws_total + (size_t)v.out->len[pick] + 2);
          {  // the values this walk will draw were written by other threads a moment ago: all their lines at once
            const char* r0 = (const char*)lead->rcache.ptr(draws_total);
            for (size_t o = 0; o < ((size_t)v.out->len[pick] + 2) * 4; o += 64) __builtin_prefetch(r0 + o);
          }
          draws = b->prep[i].seg_mode ? seg_count_draws(g, v, b->prep[i], lead->rcache.ptr(draws_total))
                                      : sub_count_draws(g, v, b->prep[i], lead->rcache.ptr(draws_total));
          ms_walks += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
          todo_tb[gi] = 1;
          right_fuz = in.reached_j;
        }
        expect_draws[gi] = draws;
        draws_total += (size_t)draws;
      }
      prev_filled = (in.filled & 1) != 0;
      prev_right_fuz = right_fuz;
    }
    grow_rands(draws_total + 1);
    draws_used = draws_total;
    ms_order += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ms_order_loop = ms_order;
  };

  // Threads never busy-wait for one another: GPU nodes are usually shared and the process
  // may run under a CPU quota, where spinning workers get the whole process throttled.
  if (analyze)
    lead->pool->run(nchunks, [&](size_t c) {
      for (size_t gi = c * per; gi < std::min(n, (c + 1) * per); gi++) analyze_gap(owner[gi], local[gi], fp, &results[gi]);
    });
  ms_ana = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  // The in-order pass and the tracebacks are pipelined over segments of the gap list: while
  // the pool traces segment s, this thread assigns the offsets of segment s+1.  When the
  // analysis has already written most fills, the few that are left are traced by the
  // in-order pass itself: waking the pool costs more than they do.
  // Long lists: what does not depend on the stream offset is done per block of the list on the pool — which gaps
  // are skipped (:369; that rule reads the previous gap of the record only), the draws of the gaps whose count is
  // fixed, their running sums inside the block — so that the sequential part only visits the gaps whose draw
  // count depends on the draws themselves (~4 %).  Offsets are then sums of three terms, put together where
  // they are needed: fixed draws of earlier blocks + of earlier gaps of the block + draws of the earlier
  // variable gaps.
  // (tests: G2S_STAGE2_BLOCK=<gaps per block> makes short lists take this path too, two blocks and up)
  static const size_t fb_env = getenv("G2S_STAGE2_BLOCK") ? (size_t)std::max(1, atoi(getenv("G2S_STAGE2_BLOCK"))) : 0;
  const size_t FB = fb_env ? fb_env : 1024;
  const bool blocks = n >= 2 * FB;
  const size_t nblk = blocks ? (n + FB - 1) / FB : 0;
  std::vector<size_t> blk_fix(nblk, 0), blk_fixbase(nblk + 1, 0), blk_rest(nblk, 0);
  std::vector<uint32_t> blk_varbase(nblk + 1, 0), vrank(blocks ? n : 0), vars_flat;
  std::vector<std::vector<uint32_t>> blk_vars(nblk);
  std::vector<size_t> var_cum;
  double ms_blocks = 0;
  if (blocks) {
    const auto tb0 = std::chrono::steady_clock::now();
    lead->pool->run(nblk, [&](size_t t) {
      const size_t s0 = t * FB, e0 = std::min(n, s0 + FB);
      auto info_of = [&](size_t gi) -> const g2s_batch::GapInfo& { return owner[gi]->info[local[gi]]; };
      // what the gap in front of the block left behind: walk back to a gap the rule cannot skip, then forward
      bool pf = false;
      int prf = 0;
      auto step = [&](size_t gi, bool act) {  // the rule for one gap; act: this block's own gaps
        const g2s_batch::GapInfo& in = info_of(gi);
        if (in.skip_thr >= 0 && pf && prf > in.skip_thr) {  // not attempted at all (:369)
          if (act) {
            g2s_result& r = results[gi];
            memset(&r, 0, sizeof r);
            r.flags = G2S_GAP_SKIPPED;
            if (in.filled & 2) arena[arena_off[gi] + (size_t)owner[gi]->jobs[local[gi]].lmf] = '\0';
          }
          pf = false;
          return 0;
        }
        if (in.kind != 0) { pf = false; return 0; }
        pf = (in.filled & 1) != 0;
        prf = in.n_len > 0 ? in.reached_j : 0;
        return in.n_len > 0 ? (in.n_len == 1 && in.fixed[0] >= 0 ? 1 : 2) : 0;  // 1: fixed draw count, 2: depends on the stream
      };
      if (s0 > 0) {
        size_t c = s0 - 1;
        while (c > 0 && info_of(c).skip_thr >= 0) c--;
        for (size_t gi = c; gi < s0; gi++) (void)step(gi, false);
      }
      size_t fsum = 0, rest = 0;
      uint32_t nv = 0;
      for (size_t gi = s0; gi < e0; gi++) {
        const g2s_batch::GapInfo& in = info_of(gi);
        rand_off[gi] = fsum;  // (inside the block, fixed draws only)
        vrank[gi] = nv;
        const int cls = step(gi, true);
        if (cls == 1) {
          fsum += (size_t)in.fixed[0];
          todo_tb[gi] = !(in.filled & 2);
          expect_draws[gi] = in.fixed[0];
        } else if (cls == 2) {
          blk_vars[t].push_back((uint32_t)gi);
          nv++;
        }
        rest += in.kind == 0 && in.n_len > 0 && !(in.filled & 2);
      }
      blk_fix[t] = fsum;
      blk_rest[t] = rest;
    });
    for (size_t t = 0; t < nblk; t++) {
      blk_fixbase[t + 1] = blk_fixbase[t] + blk_fix[t];
      blk_varbase[t + 1] = blk_varbase[t] + (uint32_t)blk_vars[t].size();
      vars_flat.insert(vars_flat.end(), blk_vars[t].begin(), blk_vars[t].end());
      n_rest += blk_rest[t];
    }
    var_cum.assign(vars_flat.size() + 1, 0);
    ms_blocks = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count();
  } else {
    for (size_t gi = 0; gi < n; gi++) {
      const g2s_batch::GapInfo& in = owner[gi]->info[local[gi]];
      n_rest += in.kind == 0 && in.n_len > 0 && !(in.filled & 2);
    }
  }
  serial = n_rest <= 64;
  const bool fast = blocks && !serial;
  // where gap gi's draws start in the stream
  auto offset_of = [&](size_t gi) -> size_t {
    return fast ? blk_fixbase[gi / FB] + rand_off[gi] + var_cum[(size_t)blk_varbase[gi / FB] + vrank[gi]] : rand_off[gi];
  };
  size_t vk = 0;  // variable gaps done so far
  auto in_order_fast = [&](size_t g_lo, size_t g_hi) {
    (void)g_lo;
    auto t0 = std::chrono::steady_clock::now();
    while (vk < vars_flat.size() && vars_flat[vk] < g_hi) {
      const size_t gi = vars_flat[vk];
      g2s_batch* b = owner[gi];
      const size_t i = local[gi];
      const g2s_batch::GapInfo& in = b->info[i];
      const size_t off = blk_fixbase[gi / FB] + rand_off[gi] + var_cum[vk];
      if (in.n_len > 1) grow_rands(off + 1);
      const int pick = in.n_len > 1 ? (int)(lead->rcache.at_const(off) % in.n_len) : 0;
      n_two += in.n_len > 1;
      int draws = in.fixed[pick];
      if (draws >= 0) {
        todo_tb[gi] = !(in.filled & 2);
      } else {  // the draw count depends on the draws: walk the parent links once for the count
        n_inline++;
        const auto tw0 = std::chrono::steady_clock::now();
        if (vk + 5 < vars_flat.size()) {  // (the closure records of a walk a few walks ahead: written by the GPU, not read yet)
          const size_t gj = vars_flat[vk + 5];
          const SubView& pv = owner[gj]->views[local[gj]];
          if (pv.segs)
            for (size_t o = 0; o < std::min<size_t>((size_t)pv.n_segs * sizeof(SegRec), 4096); o += 64) __builtin_prefetch((const char*)pv.segs + o);
        }
        const SubView& v = b->views[i];
        grow_rands(off + (size_t)v.out->len[pick] + 2);
        {
          const char* r0 = (const char*)lead->rcache.ptr(off);
          for (size_t o = 0; o < ((size_t)v.out->len[pick] + 2) * 4; o += 64) __builtin_prefetch(r0 + o);
        }
        draws = b->prep[i].seg_mode ? seg_count_draws(g, v, b->prep[i], lead->rcache.ptr(off))
                                    : sub_count_draws(g, v, b->prep[i], lead->rcache.ptr(off));
        ms_walks += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
        todo_tb[gi] = 1;
      }
      expect_draws[gi] = draws;
      var_cum[vk + 1] = var_cum[vk] + (size_t)draws;
      vk++;
    }
    if (g_hi >= n) {  // the whole list is done: what the run consumed
      draws_total = blk_fixbase[nblk] + var_cum[vars_flat.size()];
      grow_rands(draws_total + 1);
      draws_used = draws_total;
    }
    ms_order += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  const size_t nseg = serial ? 1 : (n >= 256 ? std::min<size_t>(8, n / 128) : 1);
  std::vector<std::function<void(size_t)>> jobs(nseg);
  for (size_t sg = 0; sg < nseg; sg++) {
    const size_t g_lo = n * sg / nseg, g_hi = n * (sg + 1) / nseg;
    if (fast) in_order_fast(g_lo, g_hi); else in_order_pass(g_lo, g_hi);
    if (jobs_in_flight) { lead->pool->finish(); jobs_in_flight = false; }
    jobs[sg] = [&, g_lo, g_hi](size_t c) {
      uint64_t bytes = 0;
      for (size_t gi = g_lo + c * per; gi < std::min(g_hi, g_lo + (c + 1) * per); gi++) {
        g2s_result& r = results[gi];
        const GapJob& j = owner[gi]->jobs[local[gi]];
        if ((owner[gi]->info[local[gi]].filled & 2) && (r.flags & G2S_GAP_PHASE_D)) {  // written by the analysis
          bytes += (uint64_t)r.fill_len;
          continue;
        }
        r.fill_off = (uint64_t)arena_off[gi] + (uint64_t)j.lmf;
        if (todo_tb[gi]) {
          const g2s_batch* b = owner[gi];
          const size_t i = local[gi];
          const uint32_t* rp = lead->rcache.ptr(offset_of(gi));
          if (b->prep[i].seg_mode) seg_traceback(g, fp, j, b->views[i], b->prep[i], rp, arena + arena_off[gi], &r);
          else sub_traceback(g, fp, j, b->views[i], b->prep[i], rp, arena + arena_off[gi], &r);
          // cannot happen: the draw count was proven fixed, or counted over the same draws
          if (r.draws != expect_draws[gi]) {
            r.flags |= G2S_GAP_BACKTRACE_FAIL;
            if (GENV("G2S_DEBUG_DRAWS")) {
              const SubView& vw = b->views[i];
              const GapOut& go = *vw.out;
              fprintf(stderr, "[g2s] gap %zu: traceback drew %d values, %d expected; seg mode %d, n_len %d lens %d %d start_seg %#x start_t %#x fixed %d %d stop %#x %#x n_segs %u n_xl %u flags %#x dflags %#x rand %u %u offset %llu\n",
                      gi, r.draws, (int)expect_draws[gi], (int)b->prep[i].seg_mode, go.n_len, go.len[0], go.len[1], go.start_seg, go.start_t, go.fixed_draws[0], go.fixed_draws[1],
                      go.stop[0], go.stop[1], vw.n_segs, go.n_xl, go.flags, go.dflags, rp[0] >> 1, rp[1] >> 1, (unsigned long long)offset_of(gi));
              for (uint32_t q = 0; q < std::min(vw.n_segs, 3u); q++) {
                const uint32_t sq = q == 0 ? (go.start_seg & 0xFFFFu) : q == 1 ? (go.start_seg >> 16) : 0u;
                if (sq < vw.n_segs) fprintf(stderr, "[g2s]    segment %u: node %#x depth_len %#x cnt %u ts_tt %#x par %#x %#x flags %#x\n", sq, vw.segs[sq].node, vw.segs[sq].depth_len, vw.segs[sq].cnt, vw.segs[sq].ts_tt, vw.segs[sq].par01, vw.segs[sq].par23, vw.segs[sq].flags);
              }
            }
          }
        }
        if (r.flags & G2S_GAP_PHASE_D) {
          r.fill_off = (uint64_t)arena_off[gi] + (uint64_t)(j.lmf - r.left_fuz);
          r.fill_len = (int32_t)strlen(arena + r.fill_off);
          bytes += (uint64_t)r.fill_len;
        }
      }
      if (bytes) fill_bytes.fetch_add(bytes);
    };
    if (serial) {
      for (size_t c = 0; c < (g_hi - g_lo + per - 1) / per; c++) jobs[sg](c);
    } else {
      lead->pool->post((g_hi - g_lo + per - 1) / per, jobs[sg]);
      jobs_in_flight = true;
    }
  }
  if (jobs_in_flight) lead->pool->finish();
  lead->rcache.consume(draws_used);
  auto t_end = std::chrono::steady_clock::now();
  if (GENV("G2S_DEBUG"))
    fprintf(stderr, "[g2s] host stage 2 (%s analysis): %.3f ms = setup+analysis %.3f + block pass %.3f + in-order pass %.3f (draw-count walks %.3f; %zu gaps traced inline, %zu with two lengths, %zu not traced by the analysis) + tracebacks\n",
            analyze ? "with" : "after", std::chrono::duration<double, std::milli>(t_end - t_begin).count(), ms_ana, ms_blocks, ms_order,
            ms_walks, n_inline, n_two, n_rest);
  if (timing) {
    timing->fill_bytes += fill_bytes.load();
    timing->ms_host_post += std::chrono::duration<double, std::milli>(t_end - t_begin).count();
    timing->ms_total += std::chrono::duration<double, std::milli>(t_end - t_begin).count();
  }
  return G2S_OK;
}

size_t rand_need_of(const g2s_gap* gaps, size_t n, int k) {
  size_t need = 0;
  for (size_t i = 0; i < n; i++) need += (size_t)(std::max(0, gaps[i].gap_len) + k + std::max(0, gaps[i].lmf) + std::max(0, gaps[i].rmf) + 2);
  return need;
}


// ---------------------------------------------------------------------------------------------------------
// RESIDENT MODE: the whole of fill_gap on the device.  The segment tier's kernel leaves closures and per-gap
// records in device memory, phase D3 (rand() stream, offsets, tracebacks: d3_device.hip) follows on the same
// stream, and the kernels write the result records and the fill text where the caller wants them — directly
// when those buffers are pinned (g2s_host_alloc), through pinned staging otherwise.  The host prepares the
// descriptors, launches, waits once and reads one summary.  Lists this cannot finish (a gap that outgrows the
// segment tier or whose closure needs the host's analysis, draw-count tables beyond the budget, anything a
// walk did not expect) are counted by the kernels and run again through the host path below, which stays
// the authority; three such lists in a row switch the mode off for the session.
// Returns G2S_OK (done), 1 (not applicable / fall back to the host path), or an error.
// ---------------------------------------------------------------------------------------------------------
static bool device_pointer_of(void* host, void** dev) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, host) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (attr.type != hipMemoryTypeHost || !attr.devicePointer) return false;
  *dev = attr.devicePointer;
  return true;
}

// One gap the device left to the host (D3HostItem): phase D2 on its closure segments (post.cpp), then its traceback
// over the rand() values the device copied out for it.  False when the traceback did not draw what the device's
// walk over the same closure counted (never expected: the list then takes the host path).
// (pre: the closure was analysed when it arrived — SegEarly, over the same segments: only the traceback is left)
static bool finish_gap_on_host(const Graph& g, const FillParams& fp, const GapJob& j, const GapOut& go, const SegRec* segs,
                               uint32_t n_segs, const uint32_t* rands, uint32_t expect_draws, uint64_t arena_off, char* arena,
                               g2s_result* r, const SubPrep* pre = nullptr, bool rands_packed12 = false) {
  memset(r, 0, sizeof *r);
  const auto t_fin0 = std::chrono::steady_clock::now();
  SubView v;
  v.out = &go; v.segs = segs; v.n_segs = n_segs;
  static thread_local SubPrep pp_store;  // (its vectors keep their storage from gap to gap)
  static thread_local std::vector<uint64_t> scratch;
  SubPrep& pp = pre ? const_cast<SubPrep&>(*pre) : pp_store;
  if (!pre) pp.reset();
  if (!pre && scratch.size() < 3 * (size_t)n_segs + 1) scratch.resize(3 * (size_t)n_segs + 1);
  std::vector<SubRec> own;
  if (!pre && !seg_analyze(fp, j, v, &pp, scratch.data())) {  // (G2S_STATE_D2: per-state records)
    pp.reset();
    own.resize((size_t)go.n_sub + ((size_t)go.n_xp + 1) / 2 + 1);
    uint64_t* xp = (uint64_t*)(own.data() + go.n_sub);
    seg_expand(fp, j, go, segs, n_segs, own.data(), xp);
    v.st = own.data(); v.n = go.n_sub; v.xp = xp; v.n_xp = go.n_xp; v.segs = nullptr;
    if (v.n_xp > 1) std::sort(xp, xp + v.n_xp);
    sub_analyze(fp, j, v, &pp);
  }
  r->phaseC_count = go.c_count;
  r->n_lengths = go.n_len;
  r->lengths[0] = go.len[0];
  r->lengths[1] = go.len[1];
  if (go.flags & (G2S_DEV_Q7_A | G2S_DEV_Q7_B | G2S_DEV_Q7_D)) r->flags |= G2S_GAP_Q7;
  r->flags |= pp.flags;
  r->count = pp.count;
  r->fill_off = arena_off + (uint64_t)j.lmf;
  if (!pp.phase_d) return false;
  r->vertices = pp.sub[0]; r->edges = pp.sub[1]; r->nontrivial_components = pp.sub[2];
  r->size_nontrivial_components = pp.sub[3]; r->vertices_final = pp.sub[4]; r->edges_final = pp.sub[5];
  const auto t_an = std::chrono::steady_clock::now();
  if (pp.seg_mode) seg_traceback(g, fp, j, v, pp, rands, arena + arena_off, r, rands_packed12);
  else if (!rands_packed12) sub_traceback(g, fp, j, v, pp, rands, arena + arena_off, r);
  else {  // (G2S_STATE_D2, tests: the per-state traceback reads raw words)
    std::vector<uint32_t> raw((size_t)expect_draws + 2);
    for (size_t x = 0; x < raw.size(); x++) raw[x] = ((rands[x >> 3] >> (4 * (x & 7))) & 15u) << 1;
    sub_traceback(g, fp, j, v, pp, raw.data(), arena + arena_off, r);
  }
  if (dbg_analysis_stats && n_segs >= 2000 && pp.run_mode) {
    const double* l = g2s_post_laps;
    fprintf(stderr, "[g2s] run analysis laps (us): front %.0f | collect %.0f merge+sort %.0f runs %.0f edges %.0f csr %.0f tarjan %.0f rest %.0f\n",
            l[0] - std::chrono::duration<double, std::micro>(t_fin0.time_since_epoch()).count(), l[1] - l[0], l[2] - l[1], l[3] - l[2], l[4] - l[3], l[5] - l[4], l[6] - l[5], l[7] - l[6]);
  }
  if (dbg_analysis_stats && n_segs >= 2000)
    fprintf(stderr, "[g2s] host-finished gap of %u segments: analysis %.3f ms (%s, %zu runs), traceback %.3f ms\n", n_segs,
            std::chrono::duration<double, std::milli>(t_an - t_fin0).count(), pp.run_mode ? "runs" : pp.seg_mode ? "segments" : "states", pp.runs.size(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_an).count());
  if ((uint32_t)r->draws != expect_draws || (r->flags & G2S_GAP_BACKTRACE_FAIL)) return false;
  r->fill_off = arena_off + (uint64_t)(j.lmf - r->left_fuz);
  r->fill_len = (int32_t)strlen(arena + r->fill_off);
  return true;
}

// ---- resident mode, first half: one batch's fill kernel on its session's stream, records and closures into the
// session's own device buffers (d_outs, d_sub).  *units: 16-byte units of closure records the launch may write.
// Returns G2S_OK, 1 (the batch is not one for this mode) or an error.
struct ResidentLaunch {
  uint64_t units = 0;
  bool two_waves = false;
  bool timed = false;  // HIP events around the fill kernel
  bool segw = false;   // the large variant was launched behind it
  size_t launched = 0;
};
// Resident mode brackets its kernels with HIP events on one launch in eight (the session's first included): an
// event between two kernels costs the stream 4-5 us, three of them 4 % of a 500-gap list's step, and the product
// has no use for the durations — bench.py and the tests read them.  G2S_KERNEL_TIMING=all|off|sample:N overrides.
static bool kernel_events_on(g2s_session* s) {
  const char* m = GENV("G2S_KERNEL_TIMING");
  const uint32_t seq = s->timed_seq++;
  if (m && !strcmp(m, "all")) return true;
  if (m && !strcmp(m, "off")) return false;
  uint32_t period = 8u;
  if (m && !strncmp(m, "sample:", 7)) period = (uint32_t)std::max(1, atoi(m + 7));  // (one launch in N)
  return seq % period == 0u;
}
// Where a resident list's kernels write results and fill text: the caller's buffers when the device can (page-locked:
// g2s_host_alloc), else the session's staging buffers (copied to the caller's when the list has ended).  The fill launch
// (tracebacks in the kernel) and phase D3's launch both ask; the same answer both times.
static int resident_targets(g2s_session* s, g2s_result* results, char* arena, size_t n, size_t arena_bytes, void** res_dev, void** arena_dev,
                            bool* res_direct, bool* arena_direct) {
  *res_dev = nullptr; *arena_dev = nullptr;
  *res_direct = device_pointer_of(results, res_dev);
  *arena_direct = arena_bytes == 0 || device_pointer_of(arena, arena_dev);
  if (!*res_direct) {
    HIP_TRY_S(s->h_res.ensure(n * sizeof(g2s_result)));
    HIP_TRY_S(hipHostGetDevicePointer(res_dev, s->h_res.p, 0));
  }
  if (!*arena_direct) {
    HIP_TRY_S(s->h_text.ensure(arena_bytes + 16));
    HIP_TRY_S(hipHostGetDevicePointer(arena_dev, s->h_text.p, 0));
  }
  return G2S_OK;
}
// (results / arena: where the list's results go — null: not known here, a team's group: no tracebacks in the fill kernel)
static int resident_launch_fill(g2s_batch* b, ResidentLaunch* rl, g2s_result* results = nullptr, char* arena = nullptr) {
  g2s_session* s = b->s;
  const size_t n = b->jobs.size();
  static const bool dbg_laps = getenv("G2S_DEBUG") != nullptr;
  auto t_lap = std::chrono::steady_clock::now();
  double laps_us[6] = {0, 0, 0, 0, 0, 0};
  auto lap = [&](int q) { if (!dbg_laps) return; const auto now = std::chrono::steady_clock::now(); laps_us[q] += std::chrono::duration<double, std::micro>(now - t_lap).count(); t_lap = now; };
  if (!resident_applicable(s, n) || !b->seg_tier_all || b->host_lookup) return 1;
  const Graph& g = *s->graph->g;
  const DeviceGraph& dg = g.dev.at(s->device);
  const int d_err = s->params.d_err;
  if (hipSetDevice(s->device) != hipSuccess) return fail(G2S_ERR_NO_DEVICE, "cannot select device");
  if (b->rnd_cap >= (1ull << 31) || b->dmax > 12000) return 1;  // (the trace kernel maps a whole fill in LDS)
  { const int rc = b->upload_flanks(true); if (rc != G2S_OK) return rc; }
  const int gmax = b->gmax;
  // ---- the launch order: longest gaps first (a stable counting sort; the session keeps the vectors)
  std::vector<uint32_t>& ids = s->res_ids;
  ids.clear();
  ids.reserve(n);
  bool ids_identity = false;  // the launch takes the gaps in list order, all of them
  if (b->n_valid > 1024 && !GENV("G2S_NO_LPT") && (size_t)gmax <= 8 * b->n_valid + 65536) {
    std::vector<uint32_t>& at = s->res_at;
    at.assign((size_t)gmax + 2, 0);
    for (size_t i = 0; i < n; i++) if (!b->jobs[i].bad_flank) at[(size_t)(gmax - b->jobs[i].g) + 1]++;
    for (size_t x = 1; x < at.size(); x++) at[x] += at[x - 1];
    ids.resize(b->n_valid);
    for (size_t i = 0; i < n; i++) if (!b->jobs[i].bad_flank) ids[at[(size_t)(gmax - b->jobs[i].g)]++] = (uint32_t)i;
  } else {
    for (size_t i = 0; i < n; i++) if (!b->jobs[i].bad_flank) ids.push_back((uint32_t)i);
    ids_identity = ids.size() == n;
  }
  // A deep list (-dist-error in the thousands): the launch of the large variant is as long as its slowest gap, and
  // that gap is one of the list's longest — which would only enter the large variant when the regular tier's whole
  // launch is through (0.54 of config 5's 4.1 ms).  The gaps in the upper part of the list's length range go to
  // the large variant at once, on a stream of its own beside the regular tier's kernel (half of them would have
  // outgrown the regular tier anyway; the others cost the large variant's idle workgroups some work); the rest as
  // before.  ids: [the regular tier's gaps][the early launch's gaps].
  size_t n_early = 0;
  const bool deep_list = b->dmax >= 2500 && !s->in_team_list && !GENV("G2S_NO_SEGX_TIER");
  // (not beside other lists in flight: there the device is full, and the gaps the regular tier would have finished
  // cost the large variant more than an earlier start gains — 485 k against 343 k gaps/s with three lists in flight)
  if (deep_list && ids.size() >= 256 && !b->others_in_flight && !GENV("G2S_NO_EARLY_SEGW")) {
    int gmin = gmax;
    for (uint32_t i : ids) gmin = std::min(gmin, b->jobs[i].g);
    const int gcut = gmin + (int)(0.55 * (double)(gmax - gmin));
    std::vector<uint32_t>& tmp = s->res_at;
    tmp.clear();
    size_t w = 0;
    for (size_t x = 0; x < ids.size(); x++) {
      if (b->jobs[ids[x]].g >= gcut && gmax > gmin) tmp.push_back(ids[x]);
      else ids[w++] = ids[x];
    }
    n_early = tmp.size();
    for (size_t x = 0; x < n_early; x++) ids[w + x] = tmp[x];
    // (longest first among the early ones: the slowest start first)
    std::stable_sort(ids.begin() + (ptrdiff_t)w, ids.end(), [&](uint32_t a2, uint32_t b2) { return b->jobs[a2].g > b->jobs[b2].g; });
  }
  const size_t n_reg = ids.size() - n_early;
  lap(0);
  HIP_TRY_S(s->h_gaps.ensure(n * sizeof(GapDev) + n * 4 + 16));
  HIP_TRY_S(s->h_d3.ensure(n * sizeof(D3Gap) + 2048 + 64 * 128 + G2S_RAND_WINDOW * 4));
  GapLite* gd = (GapLite*)s->h_gaps.p;  // (the short records: fill_device.h — the kernels expand them with the list's constants)
  uint32_t* ids_pinned = (uint32_t*)(gd + n);
  if (!ids.empty()) memcpy(ids_pinned, ids.data(), ids.size() * 4);
  D3Gap* dgaps = (D3Gap*)s->h_d3.p;
  if (!(b->fast_desc && s->desc_owner == b)) {
    // ---- descriptors (the preparation fills them itself when it expects this mode; a batch that is run again
    // after another one was prepared on the session finds them overwritten): GapDev for the fill kernel, D3Gap
    // for phase D3
    auto fill_range = [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; i++) {
        const GapJob& j = b->jobs[i];
        GapLite& d = gd[i];
        D3Gap& q = dgaps[i];
        q.arena_off = (uint64_t)b->arena_off[i];
        q.skip_thr = std::max(-1, std::min(j.skip_if_prev_right_fuz_gt, 32767));
        q.lmf = (uint16_t)j.lmf;
        q.kind = j.bad_flank ? 1 : 0;
        q.pad = 0;
        d.g = j.g; d.lmf = (uint16_t)j.lmf; d.rmf = (uint16_t)j.rmf;
        d.flank_off = b->flank_off[i];
        d.text_off = j.bad_flank ? 0u : j.text_off;
        d.arena_off = (uint64_t)b->arena_off[i];
        d.has_skip = j.skip_if_prev_right_fuz_gt >= 0 ? 1u : 0u;
        d.pad = 0u;
      }
    };
    const size_t per_task = std::max<size_t>(512, (n + 15) / 16), ntasks = (n + per_task - 1) / per_task;
    if (ntasks > 4) s->pool->run(ntasks, [&](size_t t) { fill_range(t * per_task, std::min(n, (t + 1) * per_task)); });
    else fill_range(0, n);
    s->desc_owner = b;
    b->fast_desc = true;
  }
  // ---- device buffers
  // A gap that outgrows the regular tier's capacities runs again in the large variant, behind the fill kernel on
  // the stream, and stays on the device like the others (one such gap used to send the whole list to the host
  // path).  Not in a team's groups yet: their closure records are copied to the lead's device by size.
  const bool rerun = (!s->in_team_list || s->team_sharded) && !GENV("G2S_NO_SEGX_TIER") && (s->segw_quiet < 8 || b->dmax >= 2500);
  rl->segw = rerun;
  // 16-byte units: two per closure segment (the large variant's closures: thousands of segments)
  const uint64_t out_states = (uint64_t)ids.size() * 128u + 2u * G2S_SEG_CAP + (rerun ? std::min<uint64_t>((uint64_t)ids.size() * 8192u, 4ull << 20) + 2u * G2S_SEGX_CAP : 0u);
  HIP_TRY_S(s->d_gaps.ensure(n * sizeof(GapDev)));
  HIP_TRY_S(s->d_ids.ensure(std::max<size_t>(ids.size() * 4, 16)));
  HIP_TRY_S(s->d_outs.ensure(n * sizeof(GapOut)));
  // ([0] closure cursor, [1] overflow list, [2] [3] work counters of the large variant's two launches; d2_device.hip:
  // [4] gaps listed, [5] small instantiation's work counter, [6] gaps passed on, [7] large one's work counter, [8] runs)
  HIP_TRY_S(s->d_counter.ensure(1536));  // ([32 .. 95]: gaps through, for a polling g2s_d2_small; [128 .. 191]: G2S_D2_PROF — neither zeroed by the kernels)
  if (s->d_counter.p != s->d2_ctr_seen) {  // (a fresh buffer: the counters that only grow start at zero)
    HIP_TRY_S(hipMemset((char*)s->d_counter.p + 128, 0, 1536 - 128));
    s->d2_ctr_seen = s->d_counter.p; s->d2_done_total = 0; s->d2_prof_on = false;
  }
  HIP_TRY_S(s->d_sub.ensure(out_states * sizeof(SubRec)));
  const uint32_t segw_wgs = (uint32_t)std::min<size_t>(ids.size(), (size_t)std::max(1, s->num_cus));
  // Phase D2 on the device for the closures the fill kernels do not analyse themselves (d2_device.hip: more than 192
  // segments, a k-mer at several depths): g2s_d2_small / g2s_d2_big behind the fill kernels, in front of phase D3.
  // Not for a team's groups that are gathered on the lead's device (their results would have to travel too), not with
  // -all-upper (no phase D2 at all).  G2S_DEVICE_D2=0: those closures are the host's, as until round 4 (post.cpp).
  // By default for the lists that fill the chip (from 3 072 gaps) when they are not deep: g2s_d2_* runs on the third
  // stream beside phase D3's kernels, and the trace waves of its gaps wait for their gap's verdict word (config 3's list:
  // no gap left to the host's threads, the step 7 % longer — the kernel works a wave per closure, 0.2-0.3 ms for the 70
  // closures against 0.1 ms of cover).  A short list keeps handing its two or three such closures to the host's threads,
  // which finish them under the trace kernel: there the launch is the step's critical path and the threads are idle.
  // A deep list (-dist-error in the thousands: closures of thousands of segments, 412 of config 5's 1 000) keeps the
  // host's threads with the early hand-over too: one wave per closure takes 4 ms where the host's pool takes 0.6 under
  // the launch.  G2S_DEVICE_D2=1 / 0: always / never.
  bool dev_d2 = (!s->in_team_list || s->team_sharded) && !s->params.skip_confident && !ids.empty();
  if (const char* env = GENV("G2S_DEVICE_D2")) dev_d2 = dev_d2 && atoi(env) != 0;
  else {
    // (a deep list all the same when the host cannot give this session the threads: the processes of a launcher — one
    // rank per GPU — and the sessions of a team share the host's CPUs.  Config 5, profiles/r05_c5_host_threads.txt: the
    // device's 4.87 ms per step whatever the threads; the host's pool 14.1 / 6.1 / 4.4 / 3.8 ms with 2 / 4 / 8 / 16)
    static const int cpus = usable_cpus();
    static const int ranks = getenv("LOCAL_WORLD_SIZE") ? std::max(1, atoi(getenv("LOCAL_WORLD_SIZE"))) : 1;
    const int threads = std::min(s->pool->size() + 1, std::max(1, cpus / (ranks * std::max(1, s->team_sessions))));
    const bool starved = threads < 6;
    // (a list of a stream with lists in flight — g2s_fill_begin — keeps the host's threads too: its closures are analysed
    // under the other lists' kernels, where g2s_d2_* would compete with them: twelve config-3 lists, three in flight,
    // 21.2 M gaps/s against 19.7 M)
    dev_d2 = dev_d2 && ((ids.size() >= 3072 && b->dmax < 2500 && (!b->through_begin || starved)) || (b->dmax >= 2500 && ids.size() >= 256 && starved));
  }
  const bool d2_deep = dev_d2 && b->dmax >= 2500;
  // (the large instantiation follows on that stream on deep lists, where the small one passes a quarter of the closures on.
  // On other lists what the small one cannot take — none of config 3's 70 — is left to the host's threads: the large one's
  // workgroups want a whole compute unit's LDS each, find it only when the trace kernel's 10 000 workgroups have all but
  // left, and that kernel's last wave waits for them: 205 instead of 196 us.  G2S_D2_BIG=1 / 2: always / every closure
  // through it (tests); =0: never.)
  const bool d2_big = dev_d2 && (GENV("G2S_D2_BIG") ? atoi(GENV("G2S_D2_BIG")) != 0 : d2_deep);
  // G2S_D2_POLL=1 (measurements): a few workgroups of the small instantiation run BESIDE the fill kernel and take the
  // closures as their gaps end (the fill launch ends with its slowest gaps: most of its wave slots are empty for its
  // last third) — what is listed late is taken by the launch behind the fill kernels.  Not on deep lists (the large
  // variant holds the compute units' LDS).  It relies on the third stream's kernels not sharing a hardware queue with
  // the fill kernel's stream (a polling kernel in front of the fill kernel in one queue would wait out its bound).
  // (measured, config 3: 0.93-0.96 ms per step against 0.89-0.92 with everything behind the fill kernel — the polling waves
  // cost the fill kernel 5 % — so: only with G2S_D2_POLL=1)
  const bool d2_poll = dev_d2 && !d2_deep && GENV("G2S_D2_POLL") && atoi(GENV("G2S_D2_POLL")) == 1;
  const uint32_t d2_poll_wgs = d2_poll ? (uint32_t)std::min<size_t>(ids.size(), (size_t)std::max(1, atoi(GENV("G2S_D2_POLL_WGS") ? GENV("G2S_D2_POLL_WGS") : "16"))) : 0u;
  const uint32_t d2_tag = d2_poll ? (0x80000000u | ((++s->d2_lists & 0x7Fu) << 24)) : 0u;
  // (how many workgroups: the trace kernel's last wave waits until every one of them has left, and they are dispatched
  // beside that kernel's 10 000 waves — config 3's list, 70 closures: 0.89 ms per step with 128 workgroups, 1.2 with 512,
  // 1.6 with 1 024, profiles/r05_d2_workgroups_c3.txt; a deep list has hundreds of closures and waits for the kernel anyway)
  const uint32_t d2_small_wgs = (uint32_t)std::min<size_t>(ids.size(), GENV("G2S_D2_SMALL_WGS") ? (size_t)std::max(1, atoi(GENV("G2S_D2_SMALL_WGS")))
                                                                                   : (b->dmax >= 2500 ? (size_t)std::max(1, s->num_cus) * 4u : (size_t)128));
  // (the large instantiation's workgroups need a whole compute unit's LDS each: on a list that is not deep only a few
  // are launched — what the small one passes on there is rare —, so that they find their units beside the trace kernel)
  const uint32_t d2_big_wgs = (uint32_t)std::min<size_t>(ids.size(), b->dmax >= 2500 ? (size_t)std::max(1, s->num_cus) : (size_t)8);
  const uint64_t d2_run_cap = out_states + 65536u;
  if (dev_d2) {
    HIP_TRY_S(s->d_d2list.ensure(std::max<size_t>(3 * n * 4, 16)));  // (list, the large instantiation's list, G2S_D2_LOG: listing times)
    HIP_TRY_S(s->d_d2out.ensure(std::max<size_t>(n * sizeof(D2Out), 32)));
    HIP_TRY_S(s->d_d2runs.ensure((size_t)d2_run_cap * 8));
    HIP_TRY_S(s->d_d2scr_small.ensure(d2_scratch_bytes(false, d2_small_wgs)));
    if (d2_big) HIP_TRY_S(s->d_d2scr_big.ensure(d2_scratch_bytes(true, d2_big_wgs)));
    HIP_TRY_S(s->d_hops.ensure((size_t)(out_states / 2 + 64) * 8));
  }
  s->d2_launched = dev_d2;
  s->d2_wait = d2_deep;  // (the trace waves of a deep list hold so much LDS that the large instantiation might find no unit)
  s->d2_wgs = dev_d2 ? d2_poll_wgs + d2_small_wgs + (d2_big ? d2_big_wgs : 0u) : 0u;
  if (rerun) {
    HIP_TRY_S(s->d_ovf.ensure(std::max<size_t>(ids.size() * 4, 16)));
    HIP_TRY_S(s->d_segx.ensure(fill_segw_scratch_bytes(segw_wgs)));
  }
  lap(1);
  hipStream_t st = s->stream;
  void* d_gaps_host = nullptr;
  HIP_TRY_S(hipHostGetDevicePointer(&d_gaps_host, s->h_gaps.p, 0));
  const GapLite* gaps_dev = (const GapLite*)d_gaps_host;
  const uint32_t* ids_dev = (const uint32_t*)(gaps_dev + n);
  const int lite_e = d_err, lite_ap = s->params.all_paths ? 1 : 0;
  // (long lists: descriptors and launch order go to device memory in front of the kernel — read over the link by
  // 10 000 starting waves they cost config 3's launch 0.04 ms: 0.365 against 0.324 ms; short lists read them over the link)
  if (ids.size() > 2048) {
    // (on the third stream, beside the look-up kernel the stream still holds — behind it the two copies were 0.04 ms
    // between that kernel and the fill kernel; the fill kernel waits for the event.  The stream's earlier work — the
    // previous list's phase D2 — read the same buffers: stream order keeps the copies behind it)
    // (not for a stream of lists in flight: nine streams of three sessions share the hardware queues, and the copies
    // on one more of them cost twelve config-3 lists 17.5 instead of 21.2 M gaps/s)
    static const bool beside_ok = !getenv("G2S_DESC_ON_STREAM");
    const bool beside = beside_ok && !b->through_begin;
    hipStream_t cs = beside ? s->stream3 : st;
    // (one copy: the short records and, right behind them, the launch order)
    HIP_TRY_S(hipMemcpyAsync(s->d_gaps.p, gd, n * sizeof(GapLite) + ids.size() * 4, hipMemcpyHostToDevice, cs));
    if (beside) { HIP_TRY_S(hipEventRecord(s->ev_desc, cs)); HIP_TRY_S(hipStreamWaitEvent(st, s->ev_desc, 0)); }
    gaps_dev = (const GapLite*)s->d_gaps.p;
    ids_dev = (const uint32_t*)(gaps_dev + n);
  }
  if (s->d_outs.clean < n * sizeof(GapOut)) HIP_TRY_S(hipMemsetAsync(s->d_outs.p, 0, n * sizeof(GapOut), st));
  if (s->d_counter.clean < 128) HIP_TRY_S(hipMemsetAsync(s->d_counter.p, 0, 128, st));
  s->d_outs.clean = 0;
  s->d_counter.clean = 0;
  // (a deep list: the closures the host will analyse leave the large variant's gaps one by one, into pinned memory)
  SegEarly early_dev;
  s->early_host = SegEarly();
  // (with phase D2 on the device — below — there is nothing to hand over early)
  if (rerun && b->dmax >= 2500 && !s->in_team_list && !GENV("G2S_NO_EARLY_HANDOVER") && !dev_d2) {
    const size_t cap_items = n, cap_segs = (size_t)std::min<uint64_t>((uint64_t)n * 1024u, 2ull << 20) + 65536u;
    const size_t b_items = (cap_items * 32 + 63) & ~(size_t)63, b_outs = (cap_items * sizeof(GapOut) + 63) & ~(size_t)63;
    HIP_TRY_S(s->h_early.ensure(b_items + b_outs + cap_segs * sizeof(SegRec)));
    HIP_TRY_S(s->d_early_ctr.ensure(16));
    char* hp = (char*)s->h_early.p;
    memset(hp, 0, b_items);  // (the ready words)
    void* dp = nullptr;
    HIP_TRY_S(hipHostGetDevicePointer(&dp, hp, 0));
    SegEarly& eh = s->early_host;
    eh.items = (uint32_t*)hp; eh.outs = (GapOut*)(hp + b_items); eh.segs = (SegRec*)(hp + b_items + b_outs);
    eh.cap_items = (uint32_t)cap_items; eh.cap_segs = (uint32_t)cap_segs;
    early_dev = eh;
    early_dev.items = (uint32_t*)dp; early_dev.outs = (GapOut*)((char*)dp + b_items); early_dev.segs = (SegRec*)((char*)dp + b_items + b_outs);
    early_dev.ctr = (unsigned long long*)s->d_early_ctr.p;
    HIP_TRY_S(hipMemsetAsync(s->d_early_ctr.p, 0, 16, st));
  }
  // (a list that is not deep, its closures left to the host — the short lists: the few closures the regular tier does not
  // analyse itself leave their gaps' waves the same way, and the thread that waits for the hand-over analyses them
  // meanwhile: resident_d3_wait.  The two counters live behind the fill kernel's cursors, zeroed with them.)
  const bool early_reg = b->dmax < 2500 && !s->in_team_list && !dev_d2 && !s->params.skip_confident && !ids.empty() &&
                         !GENV("G2S_NO_EARLY_HANDOVER");
  if (early_reg) {
    const size_t cap_items = n, cap_segs = std::max<size_t>(n * 16, 65536);
    const size_t b_items = (cap_items * 32 + 63) & ~(size_t)63, b_outs = (cap_items * sizeof(GapOut) + 63) & ~(size_t)63;
    HIP_TRY_S(s->h_early.ensure(b_items + b_outs + cap_segs * sizeof(SegRec)));
    char* hp = (char*)s->h_early.p;
    memset(hp, 0, b_items);  // (the ready words)
    void* dp = nullptr;
    HIP_TRY_S(hipHostGetDevicePointer(&dp, hp, 0));
    SegEarly& eh = s->early_host;
    eh.items = (uint32_t*)hp; eh.outs = (GapOut*)(hp + b_items); eh.segs = (SegRec*)(hp + b_items + b_outs);
    eh.cap_items = (uint32_t)cap_items; eh.cap_segs = (uint32_t)cap_segs;
    early_dev = eh;
    early_dev.items = (uint32_t*)dp; early_dev.outs = (GapOut*)((char*)dp + b_items); early_dev.segs = (SegRec*)((char*)dp + b_items + b_outs);
    early_dev.ctr = (unsigned long long*)s->d_counter.p + 10;
  }
  // (two waves per gap when the launch is short enough to end with its slowest gap — unless the sessions of a team
  // share this device: the chip is then as full as one long launch makes it)
  const bool two_waves = GENV("G2S_SEG_WAVES") ? atoi(GENV("G2S_SEG_WAVES")) == 2 : (n_reg <= 2048 && !s->team_shares_device);
  // (the early launch of the large variant — see n_early above: behind everything the stream has prepared, beside the
  // regular tier's kernel; its gaps' records and closures go where the others' do, through the same cursors)
  if (n_early && rerun) HIP_TRY_S(hipEventRecord(s->ev_pre, st));  // (what the early launch waits for: not the regular tier's kernel)
  D2Args DA;
  memset(&DA, 0, sizeof DA);
  if (dev_d2) {
    unsigned long long* ctr = (unsigned long long*)s->d_counter.p;
    DA.gaps = nullptr; DA.lite = gaps_dev; DA.lite_e = lite_e; DA.flank_nodes = (const uint32_t*)s->d_flank.p; DA.outs = (GapOut*)s->d_outs.p; DA.sub = (SubRec*)s->d_sub.p;
    DA.list = (uint32_t*)s->d_d2list.p; DA.count = ctr + 4; DA.next = ctr + 5;
    DA.d2out = (D2Out*)s->d_d2out.p; DA.runs = (uint32_t*)s->d_d2runs.p; DA.run_cursor = ctr + 8; DA.run_cap = d2_run_cap;
    DA.all_paths = s->params.all_paths ? 1 : 0; DA.list_cap = (uint32_t)n;
    DA.wgs_done = ctr + 9;
    DA.behind = (d2_deep && !GENV("G2S_D2_RELEASE")) ? 1u : 0u;  // (a deep list's phase D3 waits for the launch: s->d2_wait)
    DA.tag = d2_tag;
    static const bool d2_prof = getenv("G2S_D2_PROF") != nullptr;
    if (d2_prof) {
      s->d2_prof_on = true; DA.prof = ctr + 128;
      if (GENV("G2S_D2_LOG")) {
        if (!s->d_d2log.p) { HIP_TRY_S(s->d_d2log.ensure(16u * 8192u * 8u)); HIP_TRY_S(hipMemset(s->d_d2log.p, 0, 16u * 8192u * 8u)); }
        DA.log = (unsigned long long*)s->d_d2log.p; DA.log_cap = 8192u;
        g2s::d2_ticks_offset = (uint32_t)(2 * n); s->d2_ticks_n = n;
        HIP_TRY_S(hipMemsetAsync((uint32_t*)s->d_d2list.p + 2 * n, 0, n * 4, st));
      }
    }
    DA.pass_all = (GENV("G2S_D2_BIG") && atoi(GENV("G2S_D2_BIG")) == 2) ? 1u : 0u;  // (tests: every closure through the large instantiation)
    if (GENV("G2S_D2_NO_CHAINS")) DA.pass_all |= 2u;  // (tests: no node counts as pass-through — the whole graph of runs goes through the component search)
  }
  if (d2_poll) {  // (beside the fill kernel: behind everything the stream has prepared for it)
    s->d2_done_total += n_reg;
    D2Args DP = DA;
    DP.poll = 1u; DP.done = (const unsigned long long*)s->d_counter.p + 32; DP.expected = s->d2_done_total;
    if (!(n_early && rerun)) HIP_TRY_S(hipEventRecord(s->ev_pre, st));
    HIP_TRY_S(hipStreamWaitEvent(s->stream3, s->ev_pre, 0));
    HIP_TRY_S(launch_d2_poll(s->stream3, DP, d2_poll_wgs, (uint32_t*)s->d_d2scr_small.p, (uint32_t*)s->d_d2list.p + n,
                             (unsigned long long*)s->d_counter.p + 6));
  }
  lap(2);
  rl->timed = kernel_events_on(s);
  // (a bracketed launch: the kernel's own start and stop times go into ev[1] / ev[2] with the dispatch — two
  // hipEventRecord calls around it put packets of their own into the stream, ~10 us of such a step.  G2S_EVENT_RECORD=1:
  // as until round 6.)
  const bool ext_events = rl->timed && !GENV("G2S_EVENT_RECORD");
  if (rl->timed && !ext_events) HIP_TRY_S(hipEventRecord(s->ev[1], st));
  // (the flank look-ups in this kernel's waves: every valid gap of such a list is in this launch — inline_ok excludes
  // the deep lists, whose longest gaps start in the large variant)
  SegInline inl;
  const bool use_inl = b->inline_pending && n_early == 0;
  if (b->inline_pending && !use_inl) { const int rc = b->upload_flanks(false); if (rc != G2S_OK) return rc; }
  if (use_inl) {
    void* d_nodes = nullptr;
    HIP_TRY_S(hipHostGetDevicePointer(&d_nodes, b->nodes, 0));
    inl.lk = s->lookup; inl.text = b->inline_text_dev; inl.nodes_dev = (uint32_t*)s->d_flank.p; inl.nodes_host = (uint32_t*)d_nodes;
    inl.text_stride = b->text_stride;  // (by gap, whatever the launch order)
  }
  // (in list order: no launch order to read — on a short list one round trip of the link less at the head of every gap)
  const uint32_t* ids_fill = (ids_identity && n_reg == n) ? nullptr : ids_dev;
  // (tracebacks that have no choice to make, by the gaps' own waves: fill_seg.hip.  G2S_TRACE_IN_FILL=0: all by phase D3.)
  SegTrace tr;
  const bool use_tr = results != nullptr && !s->in_team_list && !GENV("G2S_D3_STAGE") &&
                      !(GENV("G2S_TRACE_IN_FILL") && atoi(GENV("G2S_TRACE_IN_FILL")) == 0);
  if (use_tr) {
    void *res_dev = nullptr, *arena_dev = nullptr;
    bool rd = false, ad = false;
    { const int rc = resident_targets(s, results, arena, n, b->arena_base + b->arena_bytes, &res_dev, &arena_dev, &rd, &ad); if (rc != G2S_OK) return rc; }
    tr.results = (uint32_t*)res_dev; tr.arena = (char*)arena_dev; tr.arena_base = (unsigned long long)b->arena_base;
    tr.chu = (const char*)s->d_lastch.p; tr.chd = (const char*)s->d_lastch.p + g.n;
    tr.max_states = (uint64_t)std::max<int64_t>(s->params.max_mem, 1 << 16) / 64;
    tr.k = g.k;
    // (tracebacks WITH choices as guesses, their text and records also in device memory for the trace kernel to compare
    // with — G2S_TRACE_GUESS=0: only the tracebacks that have no choice to make)
    // (lists that fill the chip only — one wave a gap: their trace kernel is the link's 8 MB; a short list's launch ends
    // with its slowest gap, whose own guess would add its 15-20 us to the step for a trace kernel that is not link-bound:
    // config 2 0.179 -> 0.194 ms with guesses.  G2S_TRACE_GUESS=1: short lists too.)
    // Round 6, later: short lists guess too, but only the gaps whose searches end while most others still run — the
    // first seven in eight (G2S_GUESS_PERCENT) in the order they end: the trace kernel then has a fifth of the text left to
    // send.)
    const bool guess = GENV("G2S_TRACE_GUESS") ? atoi(GENV("G2S_TRACE_GUESS")) != 0 : true;
    if (guess) {
      if (two_waves && !GENV("G2S_TRACE_GUESS")) {
        const int pct = GENV("G2S_GUESS_PERCENT") ? std::min(100, std::max(1, atoi(GENV("G2S_GUESS_PERCENT")))) : 87;
        tr.guess_until = pct >= 100 ? 0u : (uint32_t)std::max<size_t>(1, n_reg * (size_t)pct / 100);
      }
      HIP_TRY_S(s->d_textout.ensure(b->arena_base + b->arena_bytes + 16));
      HIP_TRY_S(s->d_resout.ensure(n * sizeof(g2s_result)));
      tr.spec_text = (char*)s->d_textout.p; tr.spec_res = (uint32_t*)s->d_resout.p;
    }
  }
  s->spec_on = use_tr && tr.spec_text != nullptr;
  HIP_TRY_S(launch_fill_seg(st, (uint32_t)n_reg, dg.succ, dg.urec, nullptr, ids_fill, (const uint32_t*)s->d_flank.p,
                          (SubRec*)s->d_sub.p, (unsigned long long)out_states, (unsigned long long*)s->d_counter.p,
                          (GapOut*)s->d_outs.p, nullptr, nullptr, s->params.skip_confident ? 1 : 0, nullptr, two_waves,
                          nullptr, nullptr, 0u, 1u, true, rerun ? (uint32_t*)s->d_ovf.p : nullptr,
                          dev_d2 ? (uint32_t*)s->d_d2list.p : nullptr, d2_tag, use_inl ? &inl : nullptr,
                          early_reg ? &early_dev : nullptr, (use_tr && tr.chu) ? &tr : nullptr, gaps_dev, lite_e, lite_ap,
                          ext_events ? s->ev[1] : nullptr, ext_events ? s->ev[2] : nullptr));
  if (use_inl) { b->inline_pending = false; b->nodes_dev_only = true; }  // (behind this kernel d_flank holds the ids; the pinned copy those of the host's gaps)
  s->lap_fill_queued = std::chrono::steady_clock::now();
  lap(3);
  if (rl->timed && !ext_events) HIP_TRY_S(hipEventRecord(s->ev[2], st));
  // (queued BEHIND the regular tier's kernel, which takes the compute units first — a workgroup of the large variant
  // needs a whole unit's LDS and stays for the launch: started first, 256 of them leave the regular tier no unit until
  // the early list runs dry (4.4 instead of 3.65 ms, when the hardware happened to order them so); its workgroups
  // begin as units run out of regular-tier gaps, a tenth of a millisecond in)
  if (n_early && rerun) {
    const uint32_t wgs1 = (uint32_t)std::min<size_t>(n_early, (size_t)std::max(1, s->num_cus));
    HIP_TRY_S(s->d_segx1.ensure(fill_segw_scratch_bytes(wgs1)));
    HIP_TRY_S(hipStreamWaitEvent(s->stream3, s->ev_pre, 0));
    HIP_TRY_S(launch_fill_segw(s->stream3, (uint32_t)n_early, wgs1, dg.succ, dg.urec, nullptr, ids_dev + n_reg,
                               (const uint32_t*)s->d_flank.p, (SubRec*)s->d_sub.p, (unsigned long long)out_states,
                               (unsigned long long*)s->d_counter.p, (GapOut*)s->d_outs.p, nullptr, nullptr,
                               s->params.skip_confident ? 1 : 0, nullptr, (uint32_t*)s->d_segx1.p,
                               (unsigned long long*)s->d_counter.p + 3, true, nullptr, (early_dev.items && !early_reg) ? &early_dev : nullptr,
                               dev_d2 ? (uint32_t*)s->d_d2list.p : nullptr, d2_tag, gaps_dev, lite_e, lite_ap));
    HIP_TRY_S(hipEventRecord(s->ev_early, s->stream3));
  }
  // (the large variant for what the launch above listed: its workgroups read the list's length from device memory and
  // leave at once when it is empty — the usual case)
  if (rerun)
    HIP_TRY_S(launch_fill_segw(st, (uint32_t)std::max<size_t>(n_reg, 1), segw_wgs, dg.succ, dg.urec, nullptr, (const uint32_t*)s->d_ovf.p,
                             (const uint32_t*)s->d_flank.p, (SubRec*)s->d_sub.p, (unsigned long long)out_states,
                             (unsigned long long*)s->d_counter.p, (GapOut*)s->d_outs.p, nullptr, nullptr,
                             s->params.skip_confident ? 1 : 0, nullptr, (uint32_t*)s->d_segx.p,
                             (unsigned long long*)s->d_counter.p + 2, true, (const unsigned long long*)s->d_counter.p + 1,
                             (early_dev.items && !early_reg) ? &early_dev : nullptr, dev_d2 ? (uint32_t*)s->d_d2list.p : nullptr, d2_tag,
                             gaps_dev, lite_e, lite_ap));
  if (n_early && rerun) HIP_TRY_S(hipStreamWaitEvent(st, s->ev_early, 0));  // (phase D3 follows on this stream: behind both)
  if (dev_d2) {  // (its workgroups read the list's length from device memory and leave at once when it is empty)
    // (on the third stream, behind the fill kernels: phase D3's first kernels do not wait for it — launch_d3)
    HIP_TRY_S(hipEventRecord(s->ev_fill, st));
    HIP_TRY_S(hipStreamWaitEvent(s->stream3, s->ev_fill, 0));
    HIP_TRY_S(launch_d2(s->stream3, DA, d2_small_wgs, d2_big ? d2_big_wgs : 0u, (uint32_t*)s->d_d2scr_small.p, (uint32_t*)s->d_d2scr_big.p,
                        (uint32_t*)s->d_d2list.p + n, (unsigned long long*)s->d_counter.p + 6, (unsigned long long*)s->d_counter.p + 7));
    HIP_TRY_S(hipEventRecord(s->ev_d2, s->stream3));
  }
  if (rerun && rl->timed) HIP_TRY_S(hipEventRecord(s->ev_segw, st));
  rl->units = out_states;
  rl->two_waves = two_waves;
  rl->launched = ids.size();
  lap(4);
  if (dbg_laps && n >= 1024)
    fprintf(stderr, "[g2s] fill launch of %zu gaps (us): launch order %.1f, descriptors + buffers %.1f, copies + events %.1f, fill kernel queued %.1f, the kernels behind it %.1f\n",
            n, laps_us[0], laps_us[1], laps_us[2], laps_us[3], laps_us[4]);
  return G2S_OK;
}
// the buffers a fill launch used are reset for the next one, off its critical path
static int resident_reset_fill(g2s_session* s, size_t n) {
  HIP_TRY(hipMemsetAsync(s->d_outs.p, 0, n * sizeof(GapOut), s->stream));
  s->d_outs.clean = n * sizeof(GapOut);
  HIP_TRY(hipMemsetAsync(s->d_counter.p, 0, 128, s->stream));
  s->d_counter.clean = 128;
  return G2S_OK;
}

// ---- resident mode, second half: phase D3 for a whole list on the lead session's device.  The list is one batch
// (run_resident) or the groups of a team (team_resident), whose records and closures have been gathered on the lead's
// device: outs_dev [n], the closure records of group q at sub_dev + q * sub_region.
struct ResidentList {
  std::vector<g2s_batch*> groups;      // in list order
  size_t n = 0, group_size = 0;        // gaps of the list; gaps per group (all but the last)
  std::vector<size_t> group_arena;     // where each group's share of the arena begins
  size_t arena_bytes = 0;
  const GapOut* outs_dev = nullptr;
  const SubRec* sub_dev = nullptr;
  uint64_t sub_region = 0;
  PinBuf* pin = nullptr;               // [D3Gap x n | summary | fill-byte counters | stream window], D3Gap filled by the caller
  size_t rnd_cap = 0;
  int dmax = 0;
  bool has_skip = false;
  const GapDev* gaps_dev = nullptr;
  hipEvent_t ready = nullptr;          // (optional) the lead's stream waits for it in front of phase D3
};
// the rand() values a list can draw, generated on a stream of their own: window of the generator's state from the
// host, then g2s_rand_fill.  Called in front of the fill kernel's launch when the list is one batch (the stream then
// fills while the look-ups run and the host prepares the launch: beside the fill kernel it cost that kernel 4 %).
static size_t rand_capacity(size_t list_cap) { return (list_cap + 2 * G2S_RAND_BLOCK) & ~(size_t)(G2S_RAND_BLOCK - 1); }
// (chain_from: the session on which the list in front of this one has its phase D3 queued — the stream continues
// from the state that list's kernels leave in device memory, behind the event that says it is there; null: from
// the host's generator, which stands at this list's first draw when every list in front of it has ended)
static int resident_rand(g2s_session* s, PinBuf* pin, size_t n, size_t list_cap, g2s_session* chain_from = nullptr) {
  if (hipSetDevice(s->device) != hipSuccess) return fail(G2S_ERR_NO_DEVICE, "cannot select device");
  const size_t rnd_cap = rand_capacity(list_cap);
  char* hsum = (char*)pin->p + n * sizeof(D3Gap) + 16 - (n * sizeof(D3Gap)) % 16;
  uint32_t* hwin = (uint32_t*)(hsum + 1024 + 64 * 128);  // (summary, the trace kernel's 64 fill-byte counters, the window)
  HIP_TRY(s->d_rnd.ensure((31 + rnd_cap + 64) * 4));
  if (chain_from) {
    HIP_TRY(hipStreamWaitEvent(s->stream2, chain_from->ev_chain, 0));
    HIP_TRY(launch_rand_window(s->stream2, (uint32_t*)s->d_rnd.p, (const uint32_t*)chain_from->d_link.p));
    HIP_TRY(launch_rand_fill(s->stream2, (uint32_t*)s->d_rnd.p, s->rtab, nullptr, (uint64_t)rnd_cap));
    return G2S_OK;
  }
  memcpy(hwin, s->rcache.window(G2S_RAND_WINDOW), G2S_RAND_WINDOW * 4);
  HIP_TRY(hipMemcpyAsync(s->d_rnd.p, hwin, G2S_RAND_WINDOW * 4, hipMemcpyHostToDevice, s->stream2));
  HIP_TRY(launch_rand_fill(s->stream2, (uint32_t*)s->d_rnd.p, s->rtab, nullptr, (uint64_t)rnd_cap));
  return G2S_OK;
}

// (what the second half — waiting, the host-finished gaps, the summary — needs of the first: queueing the kernels)
struct D3Pending {
  ResidentList L;
  bool timed = false, res_direct = false, arena_direct = false, stage_dev = false, self_clean = false;
  bool discard = false;  // (its stream continued a list that did not end on the device: nothing of it counts)
  D3Side side_h;
  D3Work W;
  D3Summary* hsum = nullptr;
  hipEvent_t d3_begin = nullptr;
  g2s_result* results = nullptr;
  char* arena = nullptr;
  std::chrono::steady_clock::time_point t_enter, t_launched;
  // a group of a sharded list (team_resident_sharded): what the second and third step launch with
  bool sharded = false;
  D3Params P;
  D3Side side_dev;
  void* res_dev = nullptr;
  char* arena_dev = nullptr;
  const D3Gap* dgaps_dev = nullptr;
  void* summary_dev = nullptr;   // the pinned summary slot as the device sees it
  uint32_t* group_fn = nullptr;  // pinned: the group function (host pointer; device pointer below)
  uint32_t* group_fn_dev = nullptr;
  size_t rnd_cap = 0;
};
// first half: everything of phase D3 queued on the session's stream(s), nothing waited for (no_spin: not even the
// few microseconds for the rand() stream's kernel — the main stream waits for its event instead)
static int resident_d3_launch(g2s_session* s, const ResidentList& L, bool timed, bool rand_launched, g2s_result* results, char* arena,
                              bool no_spin, bool sharded = false /* the first step only: classes and the group's totals */,
                              g2s_session* chain_from = nullptr /* resident_rand */) {
  const size_t n = L.n;
  const Graph& g = *s->graph->g;
  const FillParams fp = fill_params_of(s);
  const auto t_enter = std::chrono::steady_clock::now();
  // What phase D3's kernels are built for, checked for the LIST (a team's list is the sum of its groups, and a slice
  // grows beyond its size while a skip rule chains its gaps): 32-bit prefix sums over at most 20 480 gaps of at most
  // 12 000 draws each, the stream below 2^31 values.  Anything else is the host path's.
  if (n > 20480 || L.dmax > 12000 || rand_capacity(L.rnd_cap) >= (1ull << 31)) return 1;
  if (hipSetDevice(s->device) != hipSuccess) return fail(G2S_ERR_NO_DEVICE, "cannot select device");
  D3Summary* hsum = (D3Summary*)((char*)L.pin->p + n * sizeof(D3Gap) + 16 - (n * sizeof(D3Gap)) % 16);
  const size_t rnd_cap = rand_capacity(L.rnd_cap);
  HIP_TRY_S(s->d_d3.ensure(d3_work_bytes((uint32_t)n)));
  // what the hand-off kernel gives the host for the gaps whose closure the host analyses (a fraction of a per cent of a list)
  D3Side side, side_h;
  {
    side_h.cap_items = n;
    // (deep searches — -dist-error in the thousands — leave closures of thousands of segments to the host's analysis)
    const bool deep = L.dmax >= 2500;
    side_h.cap_segs = deep ? std::min<uint64_t>((uint64_t)n * 4096u, 4ull << 20) + 65536u : std::max<uint64_t>((uint64_t)n * 16u, 65536u);
    side_h.cap_rnd = deep ? rnd_cap + 65536u : rnd_cap / 8 + 65536u;
    const size_t b_items = (n * sizeof(D3HostItem) + 63) & ~(size_t)63, b_outs = (n * sizeof(GapOut) + 63) & ~(size_t)63;
    const size_t b_segs = side_h.cap_segs * sizeof(SegRec);
    const void* side_was = s->h_side.p;
    HIP_TRY_S(s->h_side.ensure(b_items + b_outs + b_segs + side_h.cap_rnd * 4 + 128));
    if (s->h_side.p != side_was) s->side_dirty = SIZE_MAX;  // (fresh memory: every ready word is to be zeroed)
    char* hp = (char*)s->h_side.p;
    side_h.items = (D3HostItem*)hp; side_h.outs = (GapOut*)(hp + b_items); side_h.segs = (SegRec*)(hp + b_items + b_outs);
    side_h.rnd = (uint32_t*)(hp + b_items + b_outs + b_segs);
    side_h.count = (unsigned long long*)(hp + b_items + b_outs + b_segs + ((side_h.cap_rnd * 4 + 63) & ~(size_t)63));
    *(volatile unsigned long long*)side_h.count = ~0ull;  // (until the number of items is known)
    // the items' ready words.  The arrays behind the items begin where a list of n gaps puts them: a list of another
    // length wrote its records, segments and values over what is item space now — every word is zeroed then; lists of
    // one length in a row (the usual case) only left the words of their own items set.
    {
      const size_t z = (s->side_layout_n == n) ? std::min(s->side_dirty, n) : n;
      for (size_t x = 0; x < z; x++) side_h.items[x].pad = 0;
    }
    s->side_layout_n = n;
    s->side_dirty = n;  // (until this list is through: any of them may be written)
    void* dp = nullptr;
    HIP_TRY_S(hipHostGetDevicePointer(&dp, s->h_side.p, 0));
    side = side_h;
    side.items = (D3HostItem*)dp; side.outs = (GapOut*)((char*)dp + b_items); side.segs = (SegRec*)((char*)dp + b_items + b_outs);
    side.rnd = (uint32_t*)((char*)dp + b_items + b_outs + b_segs);
    side.count = (unsigned long long*)((char*)dp + ((char*)side_h.count - hp));
  }
  D3Work W;
  d3_work_carve(s->d_d3.p, (uint32_t)n, &W);
  W.link = sharded ? nullptr : (uint32_t*)s->d_link.p;
  if (s->spec_on && !sharded && L.groups.size() == 1 && L.outs_dev == (const GapOut*)s->d_outs.p) {  // (this session's own fill launch left guesses)
    W.spec_text = (const char*)s->d_textout.p; W.spec_res = (const uint32_t*)s->d_resout.p;
  }
  if (s->d2_launched && L.outs_dev == (const GapOut*)s->d_outs.p) {  // (this session's own fill launch: g2s_d2_* ran behind it)
    W.d2out = (const D2Out*)s->d_d2out.p; W.d2runs = (const uint32_t*)s->d_d2runs.p; W.hops = (uint64_t*)s->d_hops.p;
  }
  // where the kernels write results and text: the caller's buffers when those are pinned, staging otherwise
  void *res_dev = nullptr, *arena_dev = nullptr;
  // (G2S_D3_STAGE=device, measurements only: the kernels write device memory, two copies bring it to the caller)
  const bool stage_dev = GENV("G2S_D3_STAGE") && !strcmp(GENV("G2S_D3_STAGE"), "device");
  bool res_direct = false, arena_direct = false;
  if (!stage_dev) { const int rc = resident_targets(s, results, arena, n, L.arena_bytes, &res_dev, &arena_dev, &res_direct, &arena_direct); if (rc != G2S_OK) return rc; }
  if (stage_dev) {
    HIP_TRY_S(s->d_resout.ensure(n * sizeof(g2s_result)));
    HIP_TRY_S(s->d_textout.ensure(L.arena_bytes + 16));
    res_dev = s->d_resout.p;
    arena_dev = s->d_textout.p;
    res_direct = arena_direct = true;
  }
  hipStream_t st = s->stream;
  void* d_dgaps = nullptr;
  HIP_TRY_S(hipHostGetDevicePointer(&d_dgaps, L.pin->p, 0));
  // the rand() values the list can draw (a team's list: generated here, beside the copies of the groups' records;
  // a group of a sharded list: in its second step, when its place in the stream is known)
  if (!rand_launched && !sharded) { const int rc = resident_rand(s, L.pin, n, L.rnd_cap, chain_from); if (rc != G2S_OK) return rc; }
  // (the per-gap descriptors of phase D3 of a short list go to device memory behind it: read over the link by the
  // list's single classify workgroup they were 3 us of its 8; a long list's are read by 40 workgroups at once, and a
  // 160 KB copy beside the fill kernel cost that kernel 4 %)
  const bool dgap_on_device = n <= 3072 && !sharded;
  if (dgap_on_device) {
    HIP_TRY_S(s->d_dgap.ensure(n * sizeof(D3Gap) + 16));
    HIP_TRY_S(hipMemcpyAsync(s->d_dgap.p, L.pin->p, n * sizeof(D3Gap), hipMemcpyHostToDevice, s->stream2));
  }
  if (!sharded) HIP_TRY_S(hipEventRecord(s->ev_rand, s->stream2));
  // (The fill kernel runs for hundreds of microseconds, the stream's work for tens: this thread waits for the
  // latter here instead of putting a wait for it into the main stream — that packet, between the fill kernel and
  // the first kernel of phase D3, cost the list 5 us.  Only if the other stream is late does the main one wait.)
  {
    const auto t_w = std::chrono::steady_clock::now();
    hipError_t q = hipErrorNotReady;
    while (!no_spin && !sharded && (q = hipEventQuery(s->ev_rand)) == hipErrorNotReady &&
           std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_w).count() < 200.0)
      cpu_relax();
    if (q != hipSuccess && !sharded) HIP_TRY_S(hipStreamWaitEvent(st, s->ev_rand, 0));
  }
  // (the time of phase D3's kernels: from the end of this session's fill kernel, or — a team's list — from here)
  // (one batch on this session: the event behind its fill kernel; a team's list: an event of its own — the lead
  // may not have launched a fill kernel at all)
  hipEvent_t d3_begin = s->ev[2];
  if (L.ready) HIP_TRY_S(hipStreamWaitEvent(st, L.ready, 0));
  if (timed && !(L.groups.size() == 1 && L.outs_dev == (const GapOut*)s->d_outs.p)) {
    HIP_TRY_S(hipEventRecord(s->ev[0], st));
    d3_begin = s->ev[0];
  }
  D3Params P;
  memset(&P, 0, sizeof P);
  P.k = fp.k; P.skip_confident = fp.skip_confident ? 1 : 0; P.all_paths = fp.all_paths ? 1 : 0; P.unique_paths = fp.unique_paths ? 1 : 0;
  P.max_states = (uint64_t)std::max<int64_t>(s->params.max_mem, 1 << 16) / 64;
  P.n = (uint32_t)n;
  P.has_skip = L.has_skip ? 1u : 0u;
  P.arena_base = 0;
  P.group_size = (uint32_t)std::max<size_t>(L.group_size, 1);
  P.sub_region = L.sub_region;
  P.laps = GENV("G2S_DEBUG") ? (1u | (GENV("G2S_DEBUG_GAP") ? ((uint32_t)atoi(GENV("G2S_DEBUG_GAP")) + 1u) << 4 : 0u)) : 0u;
  // (one batch on this session: the kernels read this session's own records and cursors, and clean up behind
  // themselves; not while the lap stamps are wanted — they live in the summary's slot)
  const bool self_clean = L.groups.size() == 1 && L.outs_dev == (const GapOut*)s->d_outs.p && !P.laps && !sharded;
  P.self_clean = self_clean ? 1u : 0u;
  if (W.d2out) { P.d2_done = (const unsigned long long*)s->d_counter.p + 9; P.d2_wgs = s->d2_wgs; }
  P.seg_cap = fp.skip_confident ? G2S_SEG_CAP : 192u;
  P.map_cap = ((uint32_t)L.dmax + 2u + 3u) & ~3u;
  if (sharded) {
    HIP_TRY_S(launch_d3_sharded_classes(st, P, W, L.outs_dev, (const D3Gap*)d_dgaps, s->d_d3.clean >= 1024 + 64 * 128));
    HIP_TRY_S(hipMemcpyAsync(hsum, W.sum, sizeof(D3Summary), hipMemcpyDeviceToHost, st));  // (the group's totals)
  } else
  HIP_TRY_S(launch_d3(st, P, W, L.gaps_dev, L.outs_dev, dgap_on_device ? (const D3Gap*)s->d_dgap.p : (const D3Gap*)d_dgaps, L.sub_dev,
                    (const char*)s->d_lastch.p, (const char*)s->d_lastch.p + g.n, s->rtab, (uint32_t*)s->d_rnd.p,
                    (uint64_t)rnd_cap, res_dev, (char*)arena_dev, side, (char*)d_dgaps + ((char*)hsum - (char*)L.pin->p),
                    s->d_d3.clean >= 1024 + 64 * 128, self_clean ? (uint32_t*)s->d_counter.p : nullptr,
                    no_spin ? s->ev_chain : nullptr /* (lists in flight: the next one's stream may wait for it) */,
                    (W.d2out && s->d2_wait) ? s->ev_d2 : nullptr,
                    (timed && !GENV("G2S_EVENT_RECORD")) ? s->ev[3] : nullptr /* (the trace kernel's own stop time) */));
  s->d_d3.clean = 0;
  if (timed && !sharded && GENV("G2S_EVENT_RECORD")) HIP_TRY_S(hipEventRecord(s->ev[3], st));
  if (stage_dev && !sharded) {
    HIP_TRY_S(hipMemcpyAsync(results, s->d_resout.p, n * sizeof(g2s_result), hipMemcpyDeviceToHost, st));
    HIP_TRY_S(hipMemcpyAsync(arena, s->d_textout.p, L.arena_bytes, hipMemcpyDeviceToHost, st));
  }
  {
    D3Pending* dp = new D3Pending;
    dp->L = L; dp->timed = timed; dp->res_direct = res_direct; dp->arena_direct = arena_direct; dp->stage_dev = stage_dev;
    dp->self_clean = self_clean; dp->side_h = side_h; dp->W = W; dp->hsum = hsum; dp->d3_begin = d3_begin; dp->results = results;
    dp->arena = arena; dp->t_enter = t_enter; dp->t_launched = std::chrono::steady_clock::now();
    dp->sharded = sharded; dp->P = P; dp->side_dev = side; dp->res_dev = res_dev; dp->arena_dev = (char*)arena_dev;
    dp->dgaps_dev = (const D3Gap*)d_dgaps; dp->summary_dev = (char*)d_dgaps + ((char*)hsum - (char*)L.pin->p);
    delete (D3Pending*)s->d3_pending;
    s->d3_pending = dp;
  }
  return G2S_OK;
}
// A group of a sharded list, second step: its place in the list's rand() stream is known — (base0, R0): the stream
// from the list's first value as far as this group can reach (win: the generator's window at the list's start), the
// layout again, the tables, the blocks, the group function into pinned memory.
static int resident_d3_sharded_tables(g2s_session* s, uint32_t base0, uint32_t R0, const uint32_t* win) {
  D3Pending* dp = (D3Pending*)s->d3_pending;
  if (!dp || !dp->sharded) return fail(G2S_ERR_ARG, "sharded list: no group pending");
  if (hipSetDevice(s->device) != hipSuccess) return fail(G2S_ERR_NO_DEVICE, "cannot select device");
  const size_t n = dp->L.n;
  const size_t rnd_cap = rand_capacity((size_t)base0 + (size_t)R0 + dp->L.rnd_cap);
  dp->rnd_cap = rnd_cap;
  {  // the stream: window from the lead's generator, then g2s_rand_fill on the second stream; the main one waits for it
    char* hsum_c = (char*)dp->hsum;
    uint32_t* hwin = (uint32_t*)(hsum_c + 1024 + 64 * 128);
    HIP_TRY_S(s->d_rnd.ensure((31 + rnd_cap + 64) * 4));
    memcpy(hwin, win, G2S_RAND_WINDOW * 4);
    HIP_TRY_S(hipMemcpyAsync(s->d_rnd.p, hwin, G2S_RAND_WINDOW * 4, hipMemcpyHostToDevice, s->stream2));
    HIP_TRY_S(launch_rand_fill(s->stream2, (uint32_t*)s->d_rnd.p, s->rtab, nullptr, (uint64_t)rnd_cap, (uint64_t)base0));
    HIP_TRY_S(hipEventRecord(s->ev_rand, s->stream2));
    HIP_TRY_S(hipStreamWaitEvent(s->stream, s->ev_rand, 0));
  }
  HIP_TRY_S(s->h_gfn.ensure(((size_t)R0 + 64) * 4));
  dp->group_fn = (uint32_t*)s->h_gfn.p;
  void* gd = nullptr;
  HIP_TRY_S(hipHostGetDevicePointer(&gd, s->h_gfn.p, 0));
  dp->group_fn_dev = (uint32_t*)gd;
  dp->P.base0 = base0; dp->P.R0 = R0;
  HIP_TRY_S(launch_d3_sharded_tables(s->stream, dp->P, dp->W, dp->L.outs_dev, dp->L.sub_dev, (uint32_t*)s->d_rnd.p, (uint64_t)rnd_cap,
                                   dp->group_fn_dev));
  HIP_TRY_S(hipMemcpyAsync(dp->hsum, dp->W.sum, sizeof(D3Summary), hipMemcpyDeviceToHost, s->stream));  // (the status behind the offsets)
  (void)n;
  return G2S_OK;
}
// third step: the deviation the group starts with is known: chain, hand-off, trace kernel (resident_d3_wait follows)
static int resident_d3_sharded_trace(g2s_session* s, uint32_t d_in) {
  D3Pending* dp = (D3Pending*)s->d3_pending;
  if (!dp || !dp->sharded) return fail(G2S_ERR_ARG, "sharded list: no group pending");
  if (hipSetDevice(s->device) != hipSuccess) return fail(G2S_ERR_NO_DEVICE, "cannot select device");
  const Graph& g = *s->graph->g;
  dp->P.d_in = d_in;
  *(volatile unsigned long long*)dp->side_h.count = ~0ull;
  HIP_TRY_S(launch_d3_sharded_trace(s->stream, dp->P, dp->W, dp->L.outs_dev, dp->L.sub_dev, (const char*)s->d_lastch.p,
                                  (const char*)s->d_lastch.p + g.n, (uint32_t*)s->d_rnd.p, (uint64_t)dp->rnd_cap, dp->res_dev,
                                  dp->arena_dev, dp->side_dev, dp->summary_dev, (dp->W.d2out && s->d2_wait) ? s->ev_d2 : nullptr));
  if (dp->timed) HIP_TRY_S(hipEventRecord(s->ev[3], s->stream));
  dp->t_launched = std::chrono::steady_clock::now();
  return G2S_OK;
}
// second half: the hand-over, the gaps the host finishes, the end of the stream's work, the summary
static int resident_d3_wait(g2s_session* s, g2s_timing* tm_out, double* ms_d3_out, bool* fell_back) {
  *fell_back = false;
  std::unique_ptr<D3Pending> dpp((D3Pending*)s->d3_pending);
  s->d3_pending = nullptr;
  if (!dpp) return fail(G2S_ERR_ARG, "resident mode: no phase D3 queued");
  const ResidentList& L = dpp->L;
  const size_t n = L.n;
  const Graph& g = *s->graph->g;
  const FillParams fp = fill_params_of(s);
  const bool timed = dpp->timed, res_direct = dpp->res_direct, arena_direct = dpp->arena_direct, stage_dev = dpp->stage_dev;
  const bool self_clean = dpp->self_clean;
  const D3Side& side_h = dpp->side_h;
  const D3Work& W = dpp->W;
  D3Summary* hsum = dpp->hsum;
  hipEvent_t d3_begin = dpp->d3_begin;
  g2s_result* results = dpp->results;
  char* arena = dpp->arena;
  const auto t_enter = dpp->t_enter, t_launched = dpp->t_launched;
  if (hipSetDevice(s->device) != hipSuccess) return fail(G2S_ERR_NO_DEVICE, "cannot select device");
  hipStream_t st = s->stream;
  // ---- gaps the device leaves to the host (their closure holds a k-mer at two depths: post.cpp analyses those):
  // handed over in front of the trace kernel, finished here while it runs — analysis of the closure, traceback,
  // record, written where the kernels write the others' (the caller's buffers or the staging)
  // (the hand-off kernel says so in pinned memory; an event between it and the trace kernel cost the stream 9 us.
  // Should the kernels end without saying so — never expected — the stream's end is noticed instead.)
  unsigned long long handed = ~0ull;
  if (dpp->discard) {  // (nothing of this attempt counts: its kernels drew from a stream that was not the list's)
    HIP_TRY_S(hipStreamSynchronize(st));
    if (!self_clean) HIP_TRY_S(hipMemsetAsync(W.sum, 0, 1024 + 64 * 128, st));
    s->d_d3.clean = 1024 + 64 * 128;
    s->self_cleaned = self_clean && hsum->status == 0;
    s->side_dirty = SIZE_MAX;
    *ms_d3_out = 0;
    *fell_back = true;
    return G2S_OK;
  }
  // ---- (a deep list) the closures that leave the large variant's gaps one by one (SegEarly): analysed as they come,
  // by the pool, while this thread waits for the hand-over — the list's deepest closure, four thousand segments, is
  // 0.4 ms of analysis that used to begin behind phase D3's front kernels
  const SegEarly eh = s->early_host;
  std::atomic<uint32_t> early_next(0);
  std::atomic<int> early_done(0);
  std::function<void(size_t)> early_worker;
  bool early_posted = false;
  // (not when the hand-over is there already — a list in flight that is ended late: every closure has arrived, and
  // the whole pool takes them largest first below)
  // (a list that is not deep has a handful of such closures, a few hundred segments each: this thread takes them itself
  // while it waits — waking the pool costs more than they do)
  // (a long list whose closures are the host's — G2S_DEVICE_D2=0 — has dozens: the pool's, as on deep lists)
  const bool early_inline = eh.cap_items && !stage_dev && L.groups.size() == 1 && L.dmax < 2500 && n < 3072;
  std::vector<std::pair<uint32_t, uint32_t>>& inline_done = s->early_inline_done;  // (gap, item) analysed here
  inline_done.clear();
  uint32_t inline_next = 0;
  auto early_take_ready = [&]() {
    while (inline_next < eh.cap_items && __atomic_load_n(&eh.items[8 * (size_t)inline_next + 4], __ATOMIC_ACQUIRE) != 0u) {
      const uint32_t idx = inline_next++;
      const uint32_t gap = eh.items[8 * (size_t)idx], ns = eh.items[8 * (size_t)idx + 1], so = eh.items[8 * (size_t)idx + 2];
      if (ns == 0u || gap >= n) continue;  // (no room for its segments: the hand-over brings them)
      if (s->early_prep.size() <= idx) { s->early_prep.resize((size_t)idx + 8); s->early_scratch.resize((size_t)idx + 8); }
      SubPrep& pp = s->early_prep[idx];
      pp.reset();
      std::vector<uint64_t>& sc = s->early_scratch[idx];
      if (sc.size() < 3 * (size_t)ns + 1) sc.resize(3 * (size_t)ns + 1);
      SubView v;
      v.out = &eh.outs[idx]; v.segs = eh.segs + so; v.n_segs = ns;
      if (seg_analyze(fp, L.groups[0]->jobs[gap], v, &pp, sc.data())) inline_done.emplace_back(gap, idx);
    }
  };
  if (!early_inline && eh.cap_items && !stage_dev && L.groups.size() == 1 && s->pool->size() > 0 &&
      __atomic_load_n(side_h.count, __ATOMIC_ACQUIRE) == ~0ull) {
    s->early_of_gap.assign(n, -1);
    if (s->early_prep.size() < eh.cap_items) { s->early_prep.resize(eh.cap_items); s->early_scratch.resize(eh.cap_items); }
    early_worker = [&](size_t) {
      for (;;) {
        uint32_t idx = early_next.load(std::memory_order_acquire);
        const bool ready = idx < eh.cap_items && __atomic_load_n(&eh.items[8 * (size_t)idx + 4], __ATOMIC_ACQUIRE) != 0u;
        if (ready) {
          if (!early_next.compare_exchange_weak(idx, idx + 1u)) continue;
          const uint32_t gap = eh.items[8 * (size_t)idx], ns = eh.items[8 * (size_t)idx + 1], so = eh.items[8 * (size_t)idx + 2];
          if (ns == 0u || gap >= n) continue;  // (no room for its segments: the hand-over brings them)
          SubPrep& pp = s->early_prep[idx];
          pp.reset();
          std::vector<uint64_t>& sc = s->early_scratch[idx];
          if (sc.size() < 3 * (size_t)ns + 1) sc.resize(3 * (size_t)ns + 1);
          SubView v;
          v.out = &eh.outs[idx]; v.segs = eh.segs + so; v.n_segs = ns;
          if (seg_analyze(fp, L.groups[0]->jobs[gap], v, &pp, sc.data())) s->early_of_gap[gap] = (int32_t)idx;
          continue;
        }
        if (early_done.load(std::memory_order_acquire)) {
          // (the kernels are through: an item that is not complete now never will be)
          idx = early_next.load(std::memory_order_acquire);
          if (idx < eh.cap_items && __atomic_load_n(&eh.items[8 * (size_t)idx + 4], __ATOMIC_ACQUIRE) != 0u) continue;
          return;
        }
        cpu_relax();
      }
    };
    s->pool->post((size_t)std::min(s->pool->size(), 15), early_worker);
    early_posted = true;
  }
  for (unsigned spins = 0;; spins++) {
    handed = __atomic_load_n(side_h.count, __ATOMIC_ACQUIRE);
    if (handed != ~0ull) break;
    if (early_inline) early_take_ready();
    if ((spins & 1023u) == 1023u && hipStreamQuery(st) != hipErrorNotReady) {
      handed = __atomic_load_n(side_h.count, __ATOMIC_ACQUIRE);
      if (handed == ~0ull) handed = 1ull << 63;
      break;
    }
    cpu_relax();
  }
  if (early_inline && handed != ~0ull && !(handed >> 63) && (handed & 0x7FFFFFFFFFFFFFFFull)) early_take_ready();  // (the fill kernels have ended: every item is there)
  if (early_posted) { early_done.store(1, std::memory_order_release); s->pool->finish(); }
  const auto t_handed = std::chrono::steady_clock::now();
  std::atomic<int> host_bad(0);
  const size_t ni = (size_t)(handed & 0x7FFFFFFFFFFFFFFFull);
  g2s_result* rs_host = res_direct && !stage_dev ? results : (g2s_result*)s->h_res.p;
  uint64_t host_fill_bytes = 0;
  if (ni && !(handed >> 63) && !stage_dev) {
    char* text = arena_direct ? arena : (char*)s->h_text.p;
    auto one = [&](size_t x) {
      // (the gap's wave of the trace kernel says when the item is complete)
      for (unsigned spins = 0; __atomic_load_n(&side_h.items[x].pad, __ATOMIC_ACQUIRE) == 0u; spins++) {
        if ((spins & 4095u) == 4095u && hipStreamQuery(st) != hipErrorNotReady &&
            __atomic_load_n(&side_h.items[x].pad, __ATOMIC_ACQUIRE) == 0u) { host_bad.fetch_add(1); return; }
        cpu_relax();
      }
      const D3HostItem& h = side_h.items[x];
      const size_t q = h.gap / L.group_size, loc = h.gap % L.group_size;
      const g2s_batch* gb = L.groups[q];
      const auto t_one = std::chrono::steady_clock::now();
      struct Lap { const std::chrono::steady_clock::time_point t0, th; const D3HostItem& h; size_t x; bool on;
                   ~Lap() { if (on && h.n_segs >= 2000) fprintf(stderr, "[g2s] host-finished item %zu (gap %u): %u segments, %u draws: picked up %.3f ms after the hand-over, %.3f ms\n", x, h.gap, h.n_segs, h.draws,
                                                              std::chrono::duration<double, std::milli>(t0 - th).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); } }
          lap{t_one, t_handed, h, x, dbg_analysis_stats};
      int32_t ei = early_posted ? s->early_of_gap[h.gap] : -1;  // (analysed when it arrived: the traceback is left)
      if (early_inline) for (const auto& pr : inline_done) if (pr.first == h.gap) { ei = (int32_t)pr.second; break; }
      const bool ok = ei >= 0 && eh.items[8 * (size_t)ei + 1] == h.n_segs
          ? finish_gap_on_host(g, fp, gb->jobs[loc], eh.outs[ei], eh.segs + eh.items[8 * (size_t)ei + 2], h.n_segs, side_h.rnd + h.rnd_off,
                               h.draws, (uint64_t)(L.group_arena[q] + gb->arena_off[loc]), text, &rs_host[h.gap], &s->early_prep[(size_t)ei], true)
          : finish_gap_on_host(g, fp, gb->jobs[loc], side_h.outs[x], side_h.segs + h.seg_off, h.n_segs, side_h.rnd + h.rnd_off, h.draws,
                               (uint64_t)(L.group_arena[q] + gb->arena_off[loc]), text, &rs_host[h.gap], nullptr, true);
      if (!ok) host_bad.fetch_add(1);
    };
    // (largest closures first: the hand-off kernel wrote every item's size before it said how many there are;
    // the last of a -dist-error 2000 list's 400 items to be picked up was its largest, a third of the wait)
    std::vector<uint32_t>& lpt = s->host_order;
    lpt.resize(ni);
    for (size_t x = 0; x < ni; x++) lpt[x] = (uint32_t)x;
    if (ni > 2) {
      std::sort(lpt.begin(), lpt.end(), [&](uint32_t a, uint32_t b) {
        const uint32_t na = side_h.items[a].n_segs, nb = side_h.items[b].n_segs;
        return na != nb ? na > nb : a < b;
      });
      s->pool->run(ni, [&](size_t x) { one((size_t)lpt[x]); });
    } else for (size_t x = 0; x < ni; x++) one(x);
    for (size_t x = 0; x < ni; x++) host_fill_bytes += (uint64_t)rs_host[side_h.items[x].gap].fill_len;
    if (const char* path = GENV("G2S_HOST_ITEMS_DUMP")) {  // (tools/host_items_replay.py: the host's share of a list, replayed without a GPU)
      if (FILE* f = fopen(path, "wb")) {
        for (size_t x = 0; x < ni; x++) {
          const D3HostItem& h = side_h.items[x];
          const GapOut& go = side_h.outs[x];
          const int32_t hd[8] = {(int32_t)h.gap, (int32_t)h.n_segs, go.c_count, go.n_len, go.len[0], go.len[1], go.reached_j, go.final_d};
          fwrite(hd, sizeof hd, 1, f);
          fwrite(side_h.segs + h.seg_off, sizeof(SegRec), h.n_segs, f);
        }
        fclose(f);
      }
    }
  }
  const auto t_finished = std::chrono::steady_clock::now();
  HIP_TRY_S(hipStreamSynchronize(st));
  const auto t_synced = std::chrono::steady_clock::now();
  float ms_d3 = 0;
  if (timed) HIP_TRY_S(hipEventElapsedTime(&ms_d3, d3_begin, s->ev[3]));
  *ms_d3_out = ms_d3;
  if (GENV("G2S_DEBUG")) {  // the kernels' lap stamps (d3_device.hip: stamp), 100 MHz
    unsigned long long lp[34];
    HIP_TRY_S(hipMemcpy(lp, (char*)W.sum + 512, sizeof lp, hipMemcpyDeviceToHost));
    auto us = [&](int a, int b) { return lp[a] && lp[b] ? ((double)lp[b] - (double)lp[a]) / 100.0 : -1.0; };
    fprintf(stderr, "[g2s] phase D3 laps (us): front classify %.1f scan %.1f | to tables %.1f: status %.1f records %.1f closure %.1f walks %.1f | to back %.1f: tables into LDS %.1f chain %.1f hand-off %.1f fence %.1f | to trace %.1f, longest wave: to closure %.1f walk %.1f bases %.1f all %.1f, first entry to last end %.1f\n",
            us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(6, 7), us(7, 8), us(8, 9), us(9, 10), us(10, 11), us(11, 12), us(12, 13),
            lp[14] / 100.0, lp[15] / 100.0, lp[16] / 100.0, (double)(lp[17] >> 24) / 100.0, us(13, 18));
    fprintf(stderr, "[g2s] the longest wave of the trace kernel: gap %llu%s\n", lp[17] & 0x7FFFFFull, (lp[17] >> 23) & 1ull ? " (handed to the host)" : "");
    fprintf(stderr, "[g2s] scan (us): sums %.1f, base + layout pass %.1f, table offsets + records %.1f\n", us(20, 21), us(21, 22), us(22, 23));
    fprintf(stderr, "[g2s] the last wave of the trace kernel: the summary's copy to the host begins %.1f us after the kernel's first entry and takes %.1f us\n", us(13, 24), us(24, 25));
    if (GENV("G2S_DEBUG_GAP") && lp[26])
      fprintf(stderr, "[g2s] gap %s in the trace kernel (us): enters %.1f behind the kernel's first wave | to closure in LDS %.1f | chain %.1f | bases classified %.1f | text %.1f | record %.1f\n",
              GENV("G2S_DEBUG_GAP"), us(13, 26), us(26, 27), us(27, 28), us(28, 29), us(29, 30), us(30, 31));
    fprintf(stderr, "[g2s] g2s_d3_tables: the last workgroup with a tile ends %.1f us after the first workgroup's entry, the last workgroup of all %.1f us\n", us(3, 32), us(3, 33));
    fprintf(stderr, "[g2s] the wave with the longest walk: %.1f us, %llu segments entered in %.1f us\n", (double)(lp[19] >> 32) / 100.0, (lp[19] >> 16) & 0xFFFF, (double)(lp[19] & 0xFFFF) / 100.0);
  }
  // (the summary is zero again for the next list: the trace kernel did it, or a memset now, off the critical path)
  if (!self_clean) HIP_TRY_S(hipMemsetAsync(W.sum, 0, 1024 + 64 * 128, st));
  s->d_d3.clean = 1024 + 64 * 128;
  s->self_cleaned = self_clean && hsum->status == 0;  // (a list the kernels gave up on: its waves left early)
  uint64_t spec_groups = 0, spec_sent = 0;
  for (int q = 0; q < 64; q++) {  // (bits 40 and up count the trace kernel's waves: d3_device.hip)
    hsum->fill_bytes += ((const unsigned long long*)((const char*)hsum + 1024))[q * 16] & ((1ull << 40) - 1);
    const unsigned long long gw = ((const unsigned long long*)((const char*)hsum + 1024))[q * 16 + 1];  // (guessed gaps: groups compared | sent << 32)
    spec_groups += gw & 0xFFFFFFFFull; spec_sent += gw >> 32;
  }
  hsum->fill_bytes += host_fill_bytes;
  // (tests: the attempt is discarded — every one, or with "rel:K" only the K-th wait since that value was first seen)
  bool test_fallback = false;
  if (const char* tf = GENV("G2S_RESIDENT_TEST_FALLBACK")) {
    static std::mutex mu;
    static std::string seen;
    static int waits = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (seen != tf) { seen = tf; waits = 0; }
    const int mine = waits++;
    test_fallback = strncmp(tf, "rel:", 4) != 0 || mine == atoi(tf + 4);
  }
  if (stage_dev && hsum->host_items) hsum->anomalies++;  // (the measurement switch has no path for host-finished gaps)
  if (hsum->status != 0 || hsum->anomalies != 0 || test_fallback || host_bad.load() || hsum->host_items != ni) {
    if (GENV("G2S_DEBUG"))
      fprintf(stderr, "[g2s] resident mode: list of %zu gaps goes to the host path (status %#x, %u gaps not finished on the device, %u anomalies, %llu table entries, %d host-finished gaps disagree)\n",
              n, hsum->status, hsum->unhandled, hsum->anomalies, (unsigned long long)hsum->table_entries, host_bad.load());
    // (only what points at a defect counts towards switching the mode off for the session: a walk that met something
    // unexpected, a host-finished gap that disagrees.  A gap beyond every tier's capacity, tables beyond the budget,
    // a side buffer that was too small are properties of that list.)
    const bool defect = hsum->anomalies != 0 || host_bad.load();
    if (hsum->unhandled) s->segw_quiet = 0;  // (a gap outgrew the regular tier: the large variant follows the next lists' kernels again)
    if (!test_fallback && defect && ++s->resident_strikes >= 3) s->resident_off = true;
    *fell_back = true;
    return G2S_OK;
  }
  s->resident_strikes = 0;
  s->segw_quiet = hsum->big_gaps ? 0 : std::min(s->segw_quiet + 1, 1 << 20);
  s->side_dirty = ni;  // (the ready words this list set)
  // ---- results that went through staging
  if (!res_direct || !arena_direct) {
    const g2s_result* rs = res_direct ? results : (const g2s_result*)s->h_res.p;
    const char* text = (const char*)s->h_text.p;
    const size_t per_task = 256, ntasks = (n + per_task - 1) / per_task;
    auto copy_range = [&](size_t t) {
      const size_t lo = t * per_task, hi = std::min(n, lo + per_task);
      if (!res_direct) memcpy(results + lo, rs + lo, (hi - lo) * sizeof(g2s_result));
      if (!arena_direct)
        for (size_t i = lo; i < hi; i++) memcpy(arena + rs[i].fill_off, text + rs[i].fill_off, (size_t)rs[i].fill_len + 1);
    };
    if (ntasks > 2) s->pool->run(ntasks, copy_range);
    else for (size_t t = 0; t < ntasks; t++) copy_range(t);
  }
  if (!dpp->sharded) s->rcache.jump((size_t)hsum->draws_total, hsum->rand_state);  // (a sharded list: the team's lead, once)
  g2s_timing& tm = *tm_out;
  tm.xA += hsum->xA; tm.sA += hsum->sA; tm.xB += hsum->xB; tm.sB += hsum->sB; tm.xD += hsum->xD; tm.sD += hsum->sD;
  tm.seg_segments += hsum->segs;
  tm.seg_tier_gaps += hsum->seg_gaps - hsum->big_gaps;
  tm.segx_tier_gaps += hsum->big_gaps;
  tm.fill_bytes += hsum->fill_bytes;
  tm.ms_d3 += ms_d3;
  tm.resident_launches++;
  tm.draw_dependent_gaps += hsum->n_var;
  tm.host_finished_gaps += (uint32_t)hsum->host_items;
  tm.traced_in_fill_gaps += hsum->traced_gaps;
  tm.guessed_in_fill_gaps += hsum->spec_gaps;
  tm.guessed_groups += (uint32_t)spec_groups; tm.guessed_groups_resent += (uint32_t)spec_sent;
  tm.d3_table_entries += hsum->table_entries;
  const auto t_end = std::chrono::steady_clock::now();
  s->lap_d3_queued = t_launched; s->lap_handed = t_handed; s->lap_finished = t_finished; s->lap_synced = t_synced; s->lap_end = t_end;
  s->laps_valid = true;
  if (GENV("G2S_DEBUG"))
    fprintf(stderr, "[g2s] resident mode, phase D3 of %zu gaps: set-up + launches %.3f ms, wait for the hand-over %.3f ms, %zu gaps finished by the host in %.3f ms, wait for the trace kernel %.3f ms, results %.3f ms (kernels %.3f ms); %u draw-dependent gaps, %llu table entries, %llu draws; results %s, text %s\n",
            n, std::chrono::duration<double, std::milli>(t_launched - t_enter).count(), std::chrono::duration<double, std::milli>(t_handed - t_launched).count(),
            ni, std::chrono::duration<double, std::milli>(t_finished - t_handed).count(), std::chrono::duration<double, std::milli>(t_synced - t_finished).count(),
            std::chrono::duration<double, std::milli>(t_end - t_synced).count(), ms_d3, hsum->n_var,
            (unsigned long long)hsum->table_entries, (unsigned long long)hsum->draws_total, res_direct ? "direct" : "staged",
            arena_direct ? "direct" : "staged");
  return G2S_OK;
}

static int resident_d3(g2s_session* s, const ResidentList& L, bool timed, bool rand_launched, g2s_result* results, char* arena,
                       g2s_timing* tm_out, double* ms_d3_out, bool* fell_back) {
  *fell_back = false;
  const int rc = resident_d3_launch(s, L, timed, rand_launched, results, arena, false);
  if (rc != G2S_OK) return rc;
  return resident_d3_wait(s, tm_out, ms_d3_out, fell_back);
}

// One batch on one session.  Returns G2S_OK (done), 1 (not applicable / fall back to the host path), or an error.
// (in two halves for lists in flight — g2s_fill_begin / g2s_fill_end: `queue` puts the list's kernels on the stream,
// phase D3 included, without waiting for anything; `finish` waits and reads the summary)
static int run_resident_queue(g2s_batch* b, g2s_result* results, char* arena, bool no_spin, ResidentLaunch* rl_out,
                              g2s_session* chain_from = nullptr) {
  g2s_session* s = b->s;
  const size_t n = b->jobs.size();
  ResidentLaunch& rl = *rl_out;
  // (the stream of rand() values first: it does not depend on the list's kernels — if the list turns out not to be
  // for this mode, a few microseconds of one kernel were for nothing)
  bool rand_launched = false;
  // (a long list only: its launch preparation leaves the stream 50 us to fill in; on a short list the two launches would
  // delay the fill kernel's by 10 us, and its stream is short enough to fill beside it.  Round 6, measured: queued BEHIND
  // the fill kernel's launch instead, the host reaches that launch 8 us earlier and the kernel, with the stream's 2 741
  // waves beside its first generation, takes 6 us longer — 0.739-0.829 against 0.764-0.820 ms alternating: kept in front)
  if (!b->pre_launched && n > 3072 && resident_applicable(s, n) && b->seg_tier_all && !b->host_lookup && b->rnd_cap < (1ull << 31) &&
      s->h_d3.cap >= n * sizeof(D3Gap) + 2048 + 64 * 128 + G2S_RAND_WINDOW * 4) {  // (pinned window in place: the launch below will not move it)
    const int rc = resident_rand(s, &s->h_d3, n, b->rnd_cap);
    if (rc != G2S_OK) return rc;
    rand_launched = true;
  }
  if (b->pre_launched) {  // (g2s_fill_begin queued the fill kernel when the list was handed over)
    rl.units = b->pre_units; rl.two_waves = b->pre_two; rl.timed = b->pre_timed; rl.segw = b->pre_segw; rl.launched = b->n_valid;
    b->pre_launched = false;
  } else {
    const int rc = resident_launch_fill(b, &rl, results, arena);
    if (rc != G2S_OK) {
      if (rand_launched) (void)hipStreamSynchronize(s->stream2);  // (its copy reads the pinned window)
      return rc;
    }
  }
  ResidentList L;
  L.groups.push_back(b);
  L.n = n; L.group_size = std::max<size_t>(n, 1);
  L.group_arena.push_back(b->arena_base);
  L.arena_bytes = b->arena_base + b->arena_bytes;
  L.outs_dev = (const GapOut*)s->d_outs.p;
  L.sub_dev = (const SubRec*)s->d_sub.p;
  L.sub_region = 0;
  L.pin = &s->h_d3;
  L.rnd_cap = b->rnd_cap; L.dmax = b->dmax; L.has_skip = b->has_skip;
  L.gaps_dev = (const GapDev*)s->d_gaps.p;
  if (b->arena_base) {  // (the D3Gap offsets are within the batch's share: make them offsets into the arena)
    D3Gap* dq = (D3Gap*)s->h_d3.p;
    for (size_t i = 0; i < n; i++) dq[i].arena_off += (uint64_t)b->arena_base;
    s->desc_owner = nullptr;
  }
  return resident_d3_launch(s, L, rl.timed, rand_launched, results, arena, no_spin, false, chain_from);
}
static int run_resident_finish(g2s_batch* b, const ResidentLaunch& rl, std::chrono::steady_clock::time_point t_enter) {
  g2s_session* s = b->s;
  const size_t n = b->jobs.size();
  double ms_d3 = 0;
  bool fell_back = false;
  const int rc = resident_d3_wait(s, &b->timing, &ms_d3, &fell_back);
  if (rc != G2S_OK) return rc;
  if (s->self_cleaned) { s->d_outs.clean = n * sizeof(GapOut); s->d_counter.clean = 128; }
  else { const int r2 = resident_reset_fill(s, n); if (r2 != G2S_OK) return r2; }
  s->self_cleaned = false;
  if (fell_back) { b->timing.resident_fallbacks++; return 1; }
  float ms_fill = 0, ms_segw = 0;
  if (rl.timed) HIP_TRY(hipEventElapsedTime(&ms_fill, s->ev[1], s->ev[2]));
  if (rl.timed && rl.segw) HIP_TRY(hipEventElapsedTime(&ms_segw, s->ev[2], s->ev_segw));
  g2s_timing& tm = b->timing;
  if (rl.segw) { tm.ms_fill_segx += ms_segw; tm.segx_launches++; tm.ms_d3 -= std::min<double>(tm.ms_d3, ms_segw); }
  tm.ms_fill_seg += ms_fill;
  tm.seg_launches++;
  if (rl.timed) tm.seg_timed_launches++;
  if (rl.two_waves) tm.seg2_launches++;
  tm.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
  if (GENV("G2S_DEBUG")) fprintf(stderr, "[g2s] resident mode: %zu gaps in %.3f ms (fill kernel %.3f ms)\n", n, tm.ms_total, ms_fill);
  return G2S_OK;
}
int run_resident(g2s_batch* b, g2s_result* results, char* arena) {
  const auto t_enter = std::chrono::steady_clock::now();
  ResidentLaunch rl;
  if (b->d3_queued) {  // (g2s_fill_begin queued everything already)
    b->d3_queued = false;
    rl.units = b->pre_units; rl.two_waves = b->pre_two; rl.timed = b->pre_timed; rl.segw = b->pre_segw; rl.launched = b->n_valid;
    if (!b->chain_broken) return run_resident_finish(b, rl, t_enter);
    // Its rand() stream was to continue, on the device, that of a list which then did not end there: the kernels
    // are waited for, what they wrote is dropped, and the list runs again from the host's generator.
    b->chain_broken = false;
    if (b->s->d3_pending) ((D3Pending*)b->s->d3_pending)->discard = true;
    const int rc = run_resident_finish(b, rl, t_enter);
    if (rc < 0) return rc;
    b->timing.resident_fallbacks = 0;  // (not a list the mode gave up on)
  }
  const int rc = run_resident_queue(b, results, arena, false, &rl);
  if (rc != G2S_OK) return rc;
  return run_resident_finish(b, rl, t_enter);
}

}  // namespace

static void d3_pending_drop(g2s_session* s) { delete (D3Pending*)s->d3_pending; s->d3_pending = nullptr; }

extern "C" int g2s_batch_run(g2s_batch* b, g2s_result* results, char* arena, size_t arena_cap) {
  g2s_env_sync();
  if (!b || !results || (!arena && b->arena_bytes)) return fail(G2S_ERR_ARG, "g2s_batch_run: bad argument");
  if (arena_cap < b->arena_bytes) return fail(G2S_ERR_ARG, "g2s_batch_run: fill arena too small");
  g2s_session* s = b->s;
  REFUSE_IN_FLIGHT(s, "g2s_batch_run");
  const size_t n = b->jobs.size();
  const auto t_run0 = std::chrono::steady_clock::now();
  {  // the whole list on the device when that applies (run_resident); otherwise, or when it gives up, the host path
    b->arena = arena;
    b->arena_base = 0;
    const int rr = run_resident(b, results, arena);
    if (rr == G2S_OK) { s->last_timing = b->timing; return G2S_OK; }
    if (rr < 0) return rr;
  }
  memset(results, 0, n * sizeof(g2s_result));
  // an unfilled gap reads as the empty string; filled ones are written in full by the traceback
  for (size_t i = 0; i < n; i++) arena[b->arena_off[i] + (size_t)b->jobs[i].lmf] = '\0';
  // rand() values are input independent: materialise what this batch will need while the
  // GPU runs (one draw per traced base plus one per gap, Gap2Seq.cpp:1440,1513)
  size_t rand_need = 0;
  for (size_t i = 0; i < n; i++)
    rand_need += (size_t)(b->jobs[i].g + s->graph->g->k + b->jobs[i].lmf + b->jobs[i].rmf + 2);
  s->bg.submit([s, rand_need]() { s->rcache.ensure(rand_need); });
  s->tier_cursor = 0;
  b->arena = arena;
  b->arena_base = 0;
  const auto t_run1 = std::chrono::steady_clock::now();
  int rc = batch_stage1(b, true, results);
  const auto t_join = std::chrono::steady_clock::now();
  s->bg.wait();
  const auto t_run2 = std::chrono::steady_clock::now();
  if (rc == G2S_OK) rc = batches_stage2(std::vector<g2s_batch*>{b}, s, results, arena, &b->timing, false);
  if (GENV("G2S_DEBUG")) {
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
    fprintf(stderr, "[g2s] batch_run: init %.3f ms, stage 1 %.3f ms, wait for rand() values %.3f ms, stage 2 %.3f ms\n",
            ms(t_run0, t_run1), ms(t_run1, t_join), ms(t_join, t_run2), ms(t_run2, std::chrono::steady_clock::now()));
  }
  s->last_timing = b->timing;
  return rc;
}
extern "C" int g2s_session_last_timing(const g2s_session* s, g2s_timing* out) {
  if (!s || !out) return fail(G2S_ERR_ARG, "g2s_session_last_timing: bad argument");
  *out = s->last_timing;
  return G2S_OK;
}

// The dispatcher's shared iterator (Gap2Seq.cpp:313-323: threads pull the next scaffold under
// a lock): the list is cut into contiguous groups and every session's host thread pulls the
// next group from one counter — a static start (session t begins with group t) with work
// stealing by construction (whoever is free takes what is left).
namespace {
struct GroupQueue {
  size_t n = 0, group = 1, ngroups = 0;
  std::atomic<size_t> next{0};
  GroupQueue(size_t n_, size_t group_) : n(n_), group(std::max<size_t>(1, group_)) { ngroups = (n + group - 1) / group; }
  // the next group [off, off+cnt) and its index, or false when the list is used up (or aborted)
  bool pull(size_t* gi, size_t* off, size_t* cnt) {
    const size_t g = next.fetch_add(1);
    if (g >= ngroups) return false;
    *gi = g;
    *off = g * group;
    *cnt = std::min(group, n - *off);
    return true;
  }
  void abort() { next.store(ngroups); }
};
}  // namespace


// A team of sessions finishes one list on the devices: every session's thread pulls groups from the shared counter
// (GroupQueue: whoever is free takes the next one), runs the look-ups and the fill kernel of its group on its GPU and
// sends the group's records and closures to the lead session's device (a peer copy over xGMI; a device-to-device
// copy when both sit on one GPU); phase D3 then runs once, for the whole list in gap order, on the lead's device
// (resident_d3) — the stream offsets chain through all groups, and the kernels write results and text straight into
// the caller's buffers.  No collective; the graph is replicated.  Returns G2S_OK (done), 1 (the host path takes the
// list) or an error.
// ---- A team's list with phase D3 SHARDED: one group per session, and every group stays on the GPU that filled it —
// that GPU classifies, builds the draw-count tables of, traces and writes the results of its own gaps, through its
// own PCIe link (gathering all groups on the lead's device made the lead trace 10 000 gaps and write 7.7 MB while
// seven GPUs waited: DESIGN §7).  What couples the groups is the one rand() stream (:178): gap i's first draw is the
// sum of the draws of all gaps before it.  Three steps, the sessions' threads meeting twice:
//   1. every GPU: fill kernel, classes, totals of its group (fewest draws, spread)        -> prefix sums over the groups
//   2. every GPU: the stream as far as its group reaches, its tables for every deviation its gaps can start with, and
//      the GROUP FUNCTION (deviation behind the group by deviation in front)               -> composed in group order
//   3. every GPU: chain from its deviation, hand-off, trace kernel; the gaps the host finishes, per session
// Results are those of one session filling the whole list (asserted by bench.py and tests/test_gpu_resident.py).
// Returns 1 when the list is not one for this form (the gather form or the host path take it).
namespace {
struct TeamBarrier {
  std::mutex mu;
  std::condition_variable cv;
  int n, waiting = 0;
  uint64_t gen = 0;
  explicit TeamBarrier(int n_) : n(n_) {}
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t g = gen;
    if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); return; }
    cv.wait(lk, [&] { return gen != g; });
  }
};
}  // namespace
static int team_resident_sharded(g2s_session* const* sessions, int nsessions, const g2s_gap* gaps, size_t n, size_t group_size,
                                 g2s_result* results, char* arena, g2s_timing* timing_out) {
  g2s_session* lead = sessions[0];
  if (GENV("G2S_TEAM_GATHER")) return 1;  // (measurements: the gather form)
  const size_t ngroups = (n + group_size - 1) / group_size;
  if (nsessions < 2 || ngroups != (size_t)nsessions || nsessions > 16) return 1;
  for (int t = 0; t < nsessions; t++) if (!resident_applicable(sessions[t], n) || sessions[t]->d3_pending) return 1;
  for (size_t q = 1; q < ngroups; q++) if (gaps[q * group_size].skip_if_prev_right_fuz_gt >= 0) return 1;  // a record cut by a group boundary
  {  // the kernels write the caller's buffers from every device: page-locked through the ABI (g2s_host_alloc), or not this form
    void* d = nullptr;
    if (!device_pointer_of(results, &d) || !device_pointer_of(arena, &d)) return 1;
  }
  struct InTeam {
    g2s_session* const* ss; int ns;
    InTeam(g2s_session* const* a, int b_) : ss(a), ns(b_) {
      for (int t = 0; t < ns; t++) {
        ss[t]->in_team_list = true; ss[t]->team_sharded = true; ss[t]->team_sessions = ns;
        for (int u = 0; u < ns; u++) if (u != t && ss[u]->device == ss[t]->device) ss[t]->team_shares_device = true;
      }
    }
    ~InTeam() { for (int t = 0; t < ns; t++) { ss[t]->in_team_list = false; ss[t]->team_sharded = false; ss[t]->team_shares_device = false; ss[t]->team_sessions = 1; } }
  } in_team(sessions, nsessions);
  const auto t_begin = std::chrono::steady_clock::now();
  std::vector<size_t> group_arena(ngroups + 1, 0);
  for (size_t gi = 0; gi < ngroups; gi++) {
    const size_t off = gi * group_size, cnt = std::min(group_size, n - off);
    group_arena[gi + 1] = group_arena[gi] + g2s_team_arena_bytes(lead, gaps + off, cnt);
  }
  std::vector<g2s_batch*> subs(ngroups, nullptr);
  std::vector<ResidentLaunch> rls(ngroups);
  std::vector<int> rcs((size_t)nsessions, G2S_OK);
  std::vector<std::string> errs((size_t)nsessions);
  std::vector<uint64_t> tot_min(ngroups, 0), tot_spread(ngroups, 0);
  std::vector<uint32_t> base0(ngroups, 0), R0(ngroups, 0), d_in(ngroups, 0);
  std::vector<const uint32_t*> gfn(ngroups, nullptr);  // the groups' functions (pinned memory of their sessions)
  std::vector<char> fell(ngroups, 0);
  std::vector<double> ms_d3(ngroups, 0.0), ms_wall(ngroups, 0.0);
  std::atomic<int> give_up(0);  // 1: not a list for this form (decided in front of the results: the caller goes on), 2: error
  uint32_t win[G2S_RAND_WINDOW];
  memcpy(win, lead->rcache.window(G2S_RAND_WINDOW), sizeof win);  // the generator where the list starts
  TeamBarrier bar(nsessions);
  auto worker = [&](int t) {
    g2s_session* s = sessions[t];
    const size_t off = (size_t)t * group_size, cnt = std::min(group_size, n - off);
    auto failed = [&](int rc) { rcs[(size_t)t] = rc; errs[(size_t)t] = tl_error; give_up.store(2); };
    // ---- step 1: fill kernel, classes, totals
    bool mine_ok = false;
    {
      g2s_batch* b = nullptr;
      const auto t0 = std::chrono::steady_clock::now();
      int rc = g2s_batch_prepare(s, gaps + off, cnt, &b);
      if (rc == G2S_OK) {
        subs[(size_t)t] = b;
        b->timing.ms_prepare = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        b->arena = arena + group_arena[(size_t)t];
        b->arena_base = group_arena[(size_t)t];
        rc = resident_launch_fill(b, &rls[(size_t)t]);
        if (rc == 1) give_up.store(1);
        else if (rc == G2S_OK) {
          ResidentList L;
          L.groups.push_back(b);
          L.n = cnt; L.group_size = std::max<size_t>(cnt, 1);
          L.group_arena.push_back(group_arena[(size_t)t]);
          L.arena_bytes = group_arena[ngroups];
          L.outs_dev = (const GapOut*)s->d_outs.p;
          L.sub_dev = (const SubRec*)s->d_sub.p;
          L.sub_region = 0;
          L.pin = &s->h_d3;
          L.rnd_cap = b->rnd_cap; L.dmax = b->dmax; L.has_skip = b->has_skip;
          L.gaps_dev = (const GapDev*)s->d_gaps.p;
          {  // (the D3Gap offsets are within the group's share: make them offsets into the arena)
            D3Gap* dq = (D3Gap*)s->h_d3.p;
            for (size_t i = 0; i < cnt; i++) dq[i].arena_off += (uint64_t)group_arena[(size_t)t];
            s->desc_owner = nullptr;
          }
          rc = resident_d3_launch(s, L, rls[(size_t)t].timed, false, results + off, arena, true, true);
          if (rc == G2S_OK && hipStreamSynchronize(s->stream) != hipSuccess) rc = fail(G2S_ERR_HIP, "sharded list, step 1");
          if (rc == G2S_OK) {
            const D3Summary* hs = ((D3Pending*)s->d3_pending)->hsum;
            if (hs->status != 0) give_up.store(1);  // (a gap beyond every tier, tables beyond the budget: the host path's business)
            tot_min[(size_t)t] = hs->draws_min; tot_spread[(size_t)t] = hs->draws_spread;
            mine_ok = true;
          }
        }
      }
      if (rc < 0) failed(rc);
    }
    bar.wait();
    // ---- the groups' places in the stream (every thread computes the same few sums)
    uint64_t b0 = 0, r0 = 0;
    for (int q = 0; q < t; q++) { b0 += tot_min[(size_t)q]; r0 += tot_spread[(size_t)q]; }
    if (b0 + r0 >= 0xF0000000ull) give_up.store(1);
    const bool go2 = give_up.load() == 0 && mine_ok;
    // ---- step 2: the stream, the tables, the group function
    if (go2) {
      base0[(size_t)t] = (uint32_t)b0; R0[(size_t)t] = (uint32_t)r0;
      int rc = resident_d3_sharded_tables(s, (uint32_t)b0, (uint32_t)r0, win);
      if (rc == G2S_OK && hipStreamSynchronize(s->stream) != hipSuccess) rc = fail(G2S_ERR_HIP, "sharded list, step 2");
      if (rc == G2S_OK && ((D3Pending*)s->d3_pending)->hsum->status != 0) give_up.store(1);
      if (rc == G2S_OK) gfn[(size_t)t] = ((D3Pending*)s->d3_pending)->group_fn;
      if (rc < 0) failed(rc);
    }
    bar.wait();
    // ---- the deviation every group starts with: the group functions composed in group order
    const bool go3 = give_up.load() == 0 && mine_ok;
    if (go3) {
      uint32_t d = 0;
      for (int q = 0; q < t; q++) d = gfn[(size_t)q][std::min(d, R0[(size_t)q])];
      d_in[(size_t)t] = d;
      // ---- step 3: chain, hand-off, trace; the gaps the host finishes; the summary
      int rc = resident_d3_sharded_trace(s, d);
      bool fb = false;
      if (rc == G2S_OK) rc = resident_d3_wait(s, &subs[(size_t)t]->timing, &ms_d3[(size_t)t], &fb);
      if (rc == G2S_OK) rc = resident_reset_fill(s, cnt);
      fell[(size_t)t] = fb ? 1 : 0;
      ms_wall[(size_t)t] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
      if (rc < 0) failed(rc);
    } else if (s->d3_pending) {  // (the list goes another way: what this session queued is waited for and dropped)
      (void)hipSetDevice(s->device);
      (void)hipStreamSynchronize(s->stream);
      (void)hipStreamSynchronize(s->stream2);
      delete (D3Pending*)s->d3_pending;
      s->d3_pending = nullptr;
      s->d_d3.clean = 0;
      (void)resident_reset_fill(s, cnt);
    }
  };
  {
    if (!lead->team_pool || lead->team_pool->size() < nsessions - 1) {
      delete lead->team_pool;
      lead->team_pool = new WorkerPool(nsessions - 1);
    }
    const std::function<void(size_t)> job = [&](size_t t) { worker((int)t); };
    lead->team_pool->run((size_t)nsessions, job);
  }
  int rc = G2S_OK;
  for (int t = 0; t < nsessions; t++) if (rcs[(size_t)t] != G2S_OK) { rc = rcs[(size_t)t]; tl_error = errs[(size_t)t]; }
  bool fell_back = give_up.load() == 1;
  for (size_t q = 0; q < ngroups; q++) fell_back = fell_back || fell[q] != 0 || !subs[q];
  g2s_timing total;
  memset(&total, 0, sizeof total);
  if (rc == G2S_OK && !fell_back) {
    // the one stream goes on behind the list: the last group's summary knows where and in which state
    const D3Summary* last = (const D3Summary*)((char*)sessions[nsessions - 1]->h_d3.p +
        (n - (size_t)(nsessions - 1) * group_size) * sizeof(D3Gap) + 16 - ((n - (size_t)(nsessions - 1) * group_size) * sizeof(D3Gap)) % 16);
    lead->rcache.jump((size_t)last->draws_total, last->rand_state);
    total.team_groups = (uint32_t)ngroups;
    total.team_sessions = (uint32_t)nsessions;
    for (size_t gi = 0; gi < ngroups; gi++) {
      const g2s_timing& t = subs[gi]->timing;
      g2s_session* s = sessions[gi];
      total.flank_bytes += t.flank_bytes; total.ms_prepare += t.ms_prepare;
      total.xA += t.xA; total.sA += t.sA; total.xB += t.xB; total.sB += t.sB; total.xD += t.xD; total.sD += t.sD;
      total.seg_segments += t.seg_segments; total.seg_tier_gaps += t.seg_tier_gaps; total.segx_tier_gaps += t.segx_tier_gaps;
      total.fill_bytes += t.fill_bytes; total.ms_d3 += t.ms_d3;
      total.draw_dependent_gaps += t.draw_dependent_gaps; total.host_finished_gaps += t.host_finished_gaps;
      total.d3_table_entries += t.d3_table_entries;
      float ms_fill = 0;
      if (rls[gi].timed && hipSetDevice(s->device) == hipSuccess && hipEventElapsedTime(&ms_fill, s->ev[1], s->ev[2]) == hipSuccess) {
        total.ms_fill_seg += ms_fill; total.seg_timed_launches++;
        if (gi < 16) total.team_ms_fill[gi] = ms_fill;
      }
      if (gi < 16) { total.team_ms_d3[gi] = ms_d3[gi]; total.team_ms_wall[gi] = ms_wall[gi]; }
      (void)hipGetLastError();
      total.seg_launches++;
      total.seg2_launches += rls[gi].two_waves ? 1u : 0u;
      if (gi < 16) total.team_groups_by_session[gi]++;
    }
    total.team_d3_sharded = 1;
    total.resident_launches = 1;  // (one list, finished on the devices)
  }
  for (g2s_batch* b : subs) if (b) g2s_batch_free(b);
  if (rc != G2S_OK) return rc;
  if (fell_back) return 1;
  total.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  if (timing_out) *timing_out = total;
  lead->last_timing = total;
  return G2S_OK;
}

static int team_resident(g2s_session* const* sessions, int nsessions, const g2s_gap* gaps, size_t n, size_t group_size,
                         g2s_result* results, char* arena, g2s_timing* timing_out) {
  {  // one group per session: the groups stay where they were filled (phase D3 sharded)
    const int rs = team_resident_sharded(sessions, nsessions, gaps, n, group_size, results, arena, timing_out);
    if (rs != 1) return rs;
  }
  g2s_session* lead = sessions[0];
  if (!resident_applicable(lead, n)) return 1;
  for (int t = 1; t < nsessions; t++) if (!resident_applicable(sessions[t], n)) return 1;
  struct InTeam {  // (the size threshold of resident mode is the list's, not the groups')
    g2s_session* const* ss; int ns;
    InTeam(g2s_session* const* a, int b_) : ss(a), ns(b_) {
      for (int t = 0; t < ns; t++) {
        ss[t]->in_team_list = true;
        for (int u = 0; u < ns; u++) if (u != t && ss[u]->device == ss[t]->device) ss[t]->team_shares_device = true;
      }
    }
    ~InTeam() { for (int t = 0; t < ns; t++) { ss[t]->in_team_list = false; ss[t]->team_shares_device = false; } }
  } in_team(sessions, nsessions);
  const auto t_begin = std::chrono::steady_clock::now();
  GroupQueue queue(n, group_size);
  const size_t ngroups = queue.ngroups;
  std::vector<g2s_batch*> subs(ngroups, nullptr);
  std::vector<int> owner(ngroups, -1);
  std::vector<float> fill_ms(ngroups, 0.f);
  std::vector<char> two(ngroups, 0), timed_g(ngroups, 0);
  std::vector<size_t> group_arena(ngroups + 1, 0);
  for (size_t gi = 0; gi < ngroups; gi++) {
    const size_t off = gi * group_size, cnt = std::min(group_size, n - off);
    group_arena[gi + 1] = group_arena[gi] + g2s_team_arena_bytes(lead, gaps + off, cnt);
  }
  // the lead's device gathers every group's records and closure records (a region per group)
  const uint64_t region = (uint64_t)group_size * 128u + 2u * G2S_SEG_CAP;  // 16-byte units, what one launch may write
  if (hipSetDevice(lead->device) != hipSuccess) return fail(G2S_ERR_NO_DEVICE, "cannot select device");
  HIP_TRY(lead->d_outs_all.ensure(n * sizeof(GapOut)));
  HIP_TRY(lead->d_sub_all.ensure(ngroups * region * sizeof(SubRec)));
  HIP_TRY(lead->h_d3all.ensure(n * sizeof(D3Gap) + 2048 + 64 * 128 + G2S_RAND_WINDOW * 4));
  D3Gap* dg_all = (D3Gap*)lead->h_d3all.p;
  // sessions on other devices push their groups' records into the lead's memory: direct access over xGMI where the
  // devices allow it (asked for once per session; without it the runtime stages the copy through the host)
  for (int t = 0; t < nsessions; t++) {
    g2s_session* o = sessions[t];
    if (o->device == lead->device || o->peer_asked_for == lead->device) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, o->device, lead->device) == hipSuccess && can && hipSetDevice(o->device) == hipSuccess) {
      const hipError_t pe = hipDeviceEnablePeerAccess(lead->device, 0);
      if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled && GENV("G2S_DEBUG"))
        fprintf(stderr, "[g2s] team: device %d cannot open device %d's memory (%s): copies go through the host\n", o->device, lead->device, hipGetErrorString(pe));
    }
    (void)hipGetLastError();
    o->peer_asked_for = lead->device;
  }
  if (hipSetDevice(lead->device) != hipSuccess) return fail(G2S_ERR_NO_DEVICE, "cannot select device");
  std::vector<int> rcs((size_t)nsessions, G2S_OK);
  std::vector<std::string> errs((size_t)nsessions);
  std::atomic<int> not_for_us(0);
  auto worker = [&](int t) {
    g2s_session* s = sessions[t];
    size_t gi = 0, off = 0, cnt = 0;
    while (queue.pull(&gi, &off, &cnt)) {
      owner[gi] = t;
      g2s_batch* b = nullptr;
      auto t0 = std::chrono::steady_clock::now();
      int rc = g2s_batch_prepare(s, gaps + off, cnt, &b);
      if (rc == G2S_OK) {
        subs[gi] = b;
        b->timing.ms_prepare = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        b->arena = arena + group_arena[gi];
        b->arena_base = group_arena[gi];
        ResidentLaunch rl;
        rc = resident_launch_fill(b, &rl);
        if (rc == 1) { not_for_us.fetch_add(1); queue.abort(); break; }
        if (rc == G2S_OK) {
          two[gi] = rl.two_waves ? 1 : 0;
          timed_g[gi] = rl.timed ? 1 : 0;
          // the group's share of the list's D3Gap array (offsets into the whole arena)
          const D3Gap* mine = (const D3Gap*)s->h_d3.p;
          for (size_t i = 0; i < cnt; i++) { dg_all[off + i] = mine[i]; dg_all[off + i].arena_off += (uint64_t)group_arena[gi]; }
          // records and closure records to the lead's device
          hipError_t e;
          char* dst_o = (char*)lead->d_outs_all.p + off * sizeof(GapOut);
          char* dst_s = (char*)lead->d_sub_all.p + gi * region * sizeof(SubRec);
          const size_t sub_bytes = (size_t)std::min<uint64_t>(rl.units, region) * sizeof(SubRec);
          if (s->device == lead->device) {
            e = hipMemcpyAsync(dst_o, s->d_outs.p, cnt * sizeof(GapOut), hipMemcpyDeviceToDevice, s->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(dst_s, s->d_sub.p, sub_bytes, hipMemcpyDeviceToDevice, s->stream);
          } else {
            e = hipMemcpyPeerAsync(dst_o, lead->device, s->d_outs.p, s->device, cnt * sizeof(GapOut), s->stream);
            if (e == hipSuccess) e = hipMemcpyPeerAsync(dst_s, lead->device, s->d_sub.p, s->device, sub_bytes, s->stream);
          }
          if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
          if (e == hipSuccess && rl.timed) e = hipEventElapsedTime(&fill_ms[gi], s->ev[1], s->ev[2]);
          if (e != hipSuccess) rc = fail(G2S_ERR_HIP, std::string("team, group to the lead's device: ") + hipGetErrorString(e));
          else rc = resident_reset_fill(s, cnt);
          s->desc_owner = nullptr;  // (h_d3 of this session is rewritten by its next group)
        }
      }
      if (rc != G2S_OK) { rcs[(size_t)t] = rc; errs[(size_t)t] = tl_error; queue.abort(); break; }
    }
  };
  {
    // one host thread per session, from the lead's team pool (persistent: creating and joining a thread per
    // session and list cost a 10 000-gap list on eight sessions 0.2 ms)
    if (nsessions > 1 && (!lead->team_pool || lead->team_pool->size() < nsessions - 1)) {
      delete lead->team_pool;
      lead->team_pool = new WorkerPool(nsessions - 1);
    }
    const std::function<void(size_t)> job = [&](size_t t) { worker((int)t); };
    if (nsessions > 1) lead->team_pool->run((size_t)nsessions, job);
    else worker(0);
  }
  int rc = G2S_OK;
  for (int t = 0; t < nsessions; t++) if (rcs[(size_t)t] != G2S_OK) { rc = rcs[(size_t)t]; tl_error = errs[(size_t)t]; }
  bool fell_back = not_for_us.load() != 0;
  for (g2s_batch* b : subs) if (!b) fell_back = fell_back || rc == G2S_OK;  // (aborted before every group ran)
  g2s_timing total;
  memset(&total, 0, sizeof total);
  if (rc == G2S_OK && !fell_back) {
    ResidentList L;
    L.groups = subs;
    L.n = n; L.group_size = group_size;
    L.group_arena = group_arena;
    L.arena_bytes = group_arena[ngroups];
    L.outs_dev = (const GapOut*)lead->d_outs_all.p;
    L.sub_dev = (const SubRec*)lead->d_sub_all.p;
    L.sub_region = region;
    L.pin = &lead->h_d3all;
    L.gaps_dev = nullptr;
    for (g2s_batch* b : subs) { L.rnd_cap += b->rnd_cap; L.dmax = std::max(L.dmax, b->dmax); L.has_skip = L.has_skip || b->has_skip; }
    double ms_d3 = 0;
    rc = resident_d3(lead, L, timed_g[0] != 0, false, results, arena, &total, &ms_d3, &fell_back);
  }
  if (rc == G2S_OK && !fell_back) {
    total.team_groups = (uint32_t)ngroups;
    total.team_sessions = (uint32_t)nsessions;
    for (size_t gi = 0; gi < ngroups; gi++) {
      const g2s_timing& t = subs[gi]->timing;
      total.flank_bytes += t.flank_bytes;
      total.ms_prepare += t.ms_prepare;
      total.ms_fill_seg += fill_ms[gi];
      total.seg_launches++;
      total.seg_timed_launches += timed_g[gi];
      total.seg2_launches += two[gi];
      if (owner[gi] >= 0 && owner[gi] < 16) total.team_groups_by_session[owner[gi]]++;
    }
  }
  for (g2s_batch* b : subs) g2s_batch_free(b);
  if (rc != G2S_OK) return rc;
  if (fell_back) return 1;
  total.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  if (timing_out) *timing_out = total;
  lead->last_timing = total;
  return G2S_OK;
}

// A team of sessions (any mix of devices, several per device allowed) fills one gap list:
// the list is cut into groups, every session's host thread pulls the next group from a
// shared counter (static start, work stealing by construction), runs stage 1 on its GPU,
// and stage 2 runs once over all groups in gap order on the first session's rand() stream.
// Two sessions on one device overlap one group's host analysis with the next group's
// kernels.  No collective: the graph is replicated, gaps are independent.
extern "C" int g2s_team_fill(g2s_session* const* sessions, int nsessions, const g2s_gap* gaps, size_t n,
                             size_t group_size, g2s_result* results, char* arena, size_t arena_cap,
                             g2s_timing* timing_out) {
  g2s_env_sync();
  if (!sessions || nsessions < 1 || (!gaps && n) || !results) return fail(G2S_ERR_ARG, "g2s_team_fill: bad argument");
  for (int t = 0; t < nsessions; t++)
    if (!sessions[t] || sessions[t]->graph != sessions[0]->graph)
      return fail(G2S_ERR_ARG, "g2s_team_fill: sessions must share one graph");
  for (int t = 0; t < nsessions; t++) REFUSE_IN_FLIGHT(sessions[t], "g2s_team_fill");
  if (group_size == 0) group_size = 2048;
  {
    const size_t need = g2s_team_arena_bytes(sessions[0], gaps, n);
    if (arena_cap < need || (!arena && need)) return fail(G2S_ERR_ARG, "g2s_team_fill: fill arena too small");
  }
  // Long lists go slice by slice.  The draw-count tables of phase D3 grow with the SQUARE of the draw-dependent gaps
  // they chain through (d3_device.hip): 16 384 gaps need about a million entries, 100 000 would need forty.  A slice
  // ends where a record ends (no skip rule reaches across), is spread over all the sessions, and the lead's rand()
  // stream runs on from slice to slice.
  constexpr size_t kSlice = 16384;
  auto slice_end = [&](size_t lo) {
    size_t hi = std::min(n, lo + kSlice);
    if (n - hi < kSlice / 4) hi = n;                                   // (no sliver at the end)
    while (hi < n && gaps[hi].skip_if_prev_right_fuz_gt >= 0) hi++;  // the record goes on: take it whole
    return hi;
  };
  if (n > kSlice + kSlice / 4 && slice_end(0) < n) {
    const auto t_begin = std::chrono::steady_clock::now();
    g2s_timing total;
    memset(&total, 0, sizeof total);
    size_t lo = 0, abase = 0;
    while (lo < n) {
      const size_t hi = slice_end(lo);
      const size_t cnt = hi - lo;
      const size_t per = std::max<size_t>(256, (cnt + (size_t)nsessions - 1) / (size_t)nsessions);
      const size_t abytes = g2s_team_arena_bytes(sessions[0], gaps + lo, cnt);
      g2s_timing tm;
      memset(&tm, 0, sizeof tm);
      const int rc = g2s_team_fill(sessions, nsessions, gaps + lo, cnt, std::min(group_size, per), results + lo, arena + abase, abytes, &tm);
      if (rc != G2S_OK) return rc;
      for (size_t i = lo; i < hi; i++) results[i].fill_off += (uint64_t)abase;  // (offsets into the whole arena)
      total.ms_right_bfs += tm.ms_right_bfs; total.ms_left_dp += tm.ms_left_dp; total.ms_extract += tm.ms_extract;
      total.ms_fill_lds += tm.ms_fill_lds; total.ms_extract_lds += tm.ms_extract_lds; total.ms_d2h += tm.ms_d2h;
      total.ms_host_post += tm.ms_host_post; total.ms_prepare += tm.ms_prepare; total.ms_d3 += tm.ms_d3;
      total.xA += tm.xA; total.sA += tm.sA; total.xB += tm.xB; total.sB += tm.sB; total.xD += tm.xD; total.sD += tm.sD;
      total.flank_bytes += tm.flank_bytes; total.fill_bytes += tm.fill_bytes; total.launches_left_dp += tm.launches_left_dp;
      total.retried_gaps += tm.retried_gaps; total.x_fill_lds += tm.x_fill_lds; total.s_fill_lds += tm.s_fill_lds;
      total.lds_tier_gaps += tm.lds_tier_gaps; total.lds_launches += tm.lds_launches; total.log_pool_gaps += tm.log_pool_gaps;
      total.rs_pool_gaps += tm.rs_pool_gaps; total.ms_fill_seg += tm.ms_fill_seg; total.seg_tier_gaps += tm.seg_tier_gaps;
      total.seg_launches += tm.seg_launches; total.seg_timed_launches += tm.seg_timed_launches; total.seg_segments += tm.seg_segments; total.ms_fill_segx += tm.ms_fill_segx;
      total.segx_tier_gaps += tm.segx_tier_gaps; total.segx_launches += tm.segx_launches; total.watchdog_gaps += tm.watchdog_gaps;
      total.seg2_launches += tm.seg2_launches; total.resident_launches += tm.resident_launches;
      total.resident_fallbacks += tm.resident_fallbacks; total.draw_dependent_gaps += tm.draw_dependent_gaps;
      total.d3_table_entries += tm.d3_table_entries; total.host_finished_gaps += tm.host_finished_gaps;
      total.team_groups += tm.team_groups; total.team_sessions = tm.team_sessions;
      for (int q = 0; q < 16; q++) {
        total.team_groups_by_session[q] += tm.team_groups_by_session[q];
        total.team_ms_fill[q] += tm.team_ms_fill[q]; total.team_ms_d3[q] += tm.team_ms_d3[q]; total.team_ms_wall[q] += tm.team_ms_wall[q];
      }
      total.team_d3_sharded = (lo == 0 ? 1u : total.team_d3_sharded) & tm.team_d3_sharded;  // (every slice)
      lo = hi;
      abase += abytes;
    }
    total.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (timing_out) *timing_out = total;
    sessions[0]->last_timing = total;
    return G2S_OK;
  }
  {
    const int rr = team_resident(sessions, nsessions, gaps, n, group_size, results, arena, timing_out);  // (the host path below when it declines)
    if (rr == G2S_OK) return G2S_OK;
    if (rr < 0) return rr;
  }
  auto t_begin = std::chrono::steady_clock::now();
  g2s_session* lead = sessions[0];
  GroupQueue queue(n, group_size);
  const size_t ngroups = queue.ngroups;
  std::vector<g2s_batch*> subs(ngroups, nullptr);
  std::vector<int> rcs((size_t)nsessions, G2S_OK);
  std::vector<std::string> errs((size_t)nsessions);
  {
    const size_t need = g2s_team_arena_bytes(lead, gaps, n);
    if (arena_cap < need || (!arena && need)) return fail(G2S_ERR_ARG, "g2s_team_fill: fill arena too small");
    memset(results, 0, n * sizeof(g2s_result));
    size_t pos = 0;  // an unfilled gap reads as the empty string
    for (size_t i = 0; i < n; i++) {
      arena[pos + (size_t)std::max(0, gaps[i].lmf)] = '\0';
      pos += g2s_team_arena_bytes(lead, gaps + i, 1);
    }
  }
  lead->bg.submit([lead, gaps, n]() { lead->rcache.ensure(rand_need_of(gaps, n, lead->graph->g->k)); });
  std::vector<size_t> group_arena(ngroups + 1, 0);  // where each group's fill buffers start
  for (size_t gi = 0; gi < ngroups; gi++) {
    const size_t off = gi * group_size, cnt = std::min(group_size, n - off);
    group_arena[gi + 1] = group_arena[gi] + g2s_team_arena_bytes(lead, gaps + off, cnt);
  }
  std::vector<int> group_owner(ngroups, -1);
  auto worker = [&](int t) {
    g2s_session* s = sessions[t];
    s->tier_cursor = 0;
    size_t gi = 0, off = 0, cnt = 0;
    while (queue.pull(&gi, &off, &cnt)) {
      group_owner[gi] = t;
      g2s_batch* b = nullptr;
      auto t0 = std::chrono::steady_clock::now();
      int rc = g2s_batch_prepare(s, gaps + off, cnt, &b);
      auto t1 = std::chrono::steady_clock::now();
      if (rc == G2S_OK) {
        subs[gi] = b;
        b->timing.ms_prepare = std::chrono::duration<double, std::milli>(t1 - t0).count();
        b->arena = arena + group_arena[gi];
        b->arena_base = group_arena[gi];
        rc = batch_stage1(b, true, results + off);
      }
      if (GENV("G2S_DEBUG"))
        fprintf(stderr, "[g2s] team session %d group %zu (%zu gaps): prepare %.3f ms, stage 1 %.3f ms\n", t, gi, cnt,
                std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
      if (rc != G2S_OK) { rcs[(size_t)t] = rc; errs[(size_t)t] = tl_error; queue.abort(); break; }
    }
  };
  {
    // one host thread per session, from the lead's team pool (persistent: creating and joining a thread per
    // session and list cost a 10 000-gap list on eight sessions 0.2 ms)
    if (nsessions > 1 && (!lead->team_pool || lead->team_pool->size() < nsessions - 1)) {
      delete lead->team_pool;
      lead->team_pool = new WorkerPool(nsessions - 1);
    }
    const std::function<void(size_t)> job = [&](size_t t) { worker((int)t); };
    if (nsessions > 1) lead->team_pool->run((size_t)nsessions, job);
    else worker(0);
  }
  lead->bg.wait();
  int rc = G2S_OK;
  for (int t = 0; t < nsessions; t++) if (rcs[(size_t)t] != G2S_OK) { rc = rcs[(size_t)t]; tl_error = errs[(size_t)t]; }
  g2s_timing total;
  memset(&total, 0, sizeof total);
  if (rc == G2S_OK) {
    {
      for (g2s_batch* b : subs) {
        const g2s_timing& t = b->timing;
        total.ms_right_bfs += t.ms_right_bfs; total.ms_left_dp += t.ms_left_dp; total.ms_extract += t.ms_extract;
        total.ms_fill_lds += t.ms_fill_lds; total.ms_extract_lds += t.ms_extract_lds; total.ms_d2h += t.ms_d2h;
        total.ms_host_post += t.ms_host_post;
        total.xA += t.xA; total.sA += t.sA; total.xB += t.xB; total.sB += t.sB; total.xD += t.xD; total.sD += t.sD;
        total.flank_bytes += t.flank_bytes; total.launches_left_dp += t.launches_left_dp;
        total.retried_gaps += t.retried_gaps; total.x_fill_lds += t.x_fill_lds; total.s_fill_lds += t.s_fill_lds;
        total.lds_tier_gaps += t.lds_tier_gaps; total.lds_launches += t.lds_launches;
        total.log_pool_gaps += t.log_pool_gaps; total.rs_pool_gaps += t.rs_pool_gaps;
        total.ms_prepare += t.ms_prepare;
        total.ms_fill_seg += t.ms_fill_seg; total.seg_tier_gaps += t.seg_tier_gaps; total.seg_launches += t.seg_launches; total.seg_timed_launches += t.seg_timed_launches;
        total.seg_segments += t.seg_segments;
        total.ms_fill_segx += t.ms_fill_segx; total.segx_tier_gaps += t.segx_tier_gaps; total.segx_launches += t.segx_launches;
        total.watchdog_gaps += t.watchdog_gaps; total.seg2_launches += t.seg2_launches;
      }
      rc = batches_stage2(subs, lead, results, arena, &total, false);
      total.team_groups = (uint32_t)ngroups;
      total.team_sessions = (uint32_t)nsessions;
      for (size_t q = 0; q < ngroups; q++) if (group_owner[q] >= 0 && group_owner[q] < 16) total.team_groups_by_session[group_owner[q]]++;
    }
  }
  for (g2s_batch* b : subs) g2s_batch_free(b);
  total.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  if (timing_out) *timing_out = total;
  return rc;
}

extern "C" size_t g2s_team_arena_bytes(const g2s_session* s, const g2s_gap* gaps, size_t n) {
  if (!s || (!gaps && n)) return 0;
  size_t need = 0;
  for (size_t i = 0; i < n; i++)
    need += (size_t)(std::max(0, gaps[i].gap_len) + s->graph->g->k + s->params.d_err + std::max(0, gaps[i].lmf) +
                     std::max(0, gaps[i].rmf) + 1 + 2);
  return need;
}

// the group size a list of n gaps is cut with on this session (a list of up to that many gaps runs on the session alone)
static size_t team_group_for(const g2s_session* s, size_t n) {
  if (s->team_group == G2S_GROUP_PER_SESSION && !s->helpers.empty()) {
    const size_t ns = s->helpers.size() + 1;
    if (n < 512 * ns) return n;  // (too short to be worth the team)
    return (n + ns - 1) / ns;    // one group per session: phase D3 sharded (team_resident_sharded)
  }
  if (s->team_group && s->team_group != G2S_GROUP_PER_SESSION) return s->team_group;
  return s->helpers.empty() ? (size_t)16384 : (size_t)2048;
}

extern "C" int g2s_fill_batch(g2s_session* s, const g2s_gap* gaps, size_t n, g2s_result* results, char* fill_arena,
                              size_t arena_cap) {
  g2s_env_sync();
  if (!s) return fail(G2S_ERR_ARG, "g2s_fill_batch: bad argument");
  REFUSE_IN_FLIGHT(s, "g2s_fill_batch");
  // long lists go through the group pipeline: with helpers to use every session, without
  // them to bound the HBM the state logs of one launch take
  const size_t group = team_group_for(s, n);
  if (n > group) {
    std::vector<g2s_session*> team{s};
    team.insert(team.end(), s->helpers.begin(), s->helpers.end());
    return g2s_team_fill(team.data(), (int)team.size(), gaps, n, group, results, fill_arena, arena_cap, nullptr);
  }
  const auto t_begin = std::chrono::steady_clock::now();
  g2s_batch* b = nullptr;
  int rc = g2s_batch_prepare(s, gaps, n, &b);
  if (rc != G2S_OK) return rc;
  const double ms_prep = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  s->laps_valid = false;
  rc = g2s_batch_run(b, results, fill_arena, arena_cap);
  const auto t_free = std::chrono::steady_clock::now();
  g2s_batch_free(b);
  if (rc == G2S_OK && s->laps_valid && s->last_timing.resident_launches == 1 && s->last_timing.resident_fallbacks == 0) {
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::micro>(c - a).count(); };
    double* h = s->last_timing.host_us;
    h[0] = us(t_begin, s->lap_fill_queued); h[1] = us(s->lap_fill_queued, s->lap_d3_queued); h[2] = us(s->lap_d3_queued, s->lap_handed);
    h[3] = us(s->lap_handed, s->lap_finished); h[4] = us(s->lap_finished, s->lap_synced); h[5] = us(s->lap_synced, std::chrono::steady_clock::now());
  }
  if (GENV("G2S_DEBUG"))
    fprintf(stderr, "[g2s] fill_batch: prepare %.3f ms, free %.3f ms\n", ms_prep,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_free).count());
  s->last_timing.ms_prepare = ms_prep;
  s->last_timing.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  return rc;
}

// ---- lists in flight (Gap2Seq-core -stream-gaps, bench.py --stream-lists): the product's steady state is a
// SEQUENCE of lists, and a list's step is a chain — preparation, look-ups, fill kernel, phase D3, of which the last
// writes the results through the link.  g2s_fill_begin queues all of it and returns; g2s_fill_end finishes the oldest
// list begun.  With further lists begun before the first is ended, their look-ups and fill kernels run on the device
// while the older ones' phase D3 writes through the link, and the host prepares the next list meanwhile.  The lists
// take turns on the session and two twins of it on the same device (own streams and buffers); the rand() stream is the
// session's: on the host it is handed to whichever session ends a list; on the device a list's stream is generated
// from the state the list in front of it leaves in device memory (D3Work.link), so that the host is not in the chain
// from one list's draws to the next one's.  The results are those of g2s_fill_batch called list by list.
namespace {
// Phase D3 of every list in flight that can have it now, oldest first.  The oldest list's stream starts where the
// host's generator stands (every list in front of it has ended); a list behind one whose phase D3 is queued
// continues that one's stream on the device.  A list behind one that is to run again, or that is not on the device
// at all, waits for it to end.
int inflight_settle(g2s_session* s) {
  const bool no_chain = GENV("G2S_NO_DEVICE_CHAIN") != nullptr;  // (read per call: the tests switch it)
  for (int i = 0; i < s->n_inflight; i++) {
    g2s_session::InFlight& f = s->inflight[i];
    if (!f.b) break;  // (a list for g2s_fill_batch: it draws on the host, when it is ended)
    if (f.b->d3_queued) continue;
    if (!f.b->pre_launched) break;  // (not a list for resident mode: the host path, when it is ended)
    g2s_session* chain_from = nullptr;
    if (i > 0) {
      const g2s_session::InFlight& o = s->inflight[i - 1];
      if (no_chain || !o.b || !o.b->d3_queued || o.b->chain_broken) break;
      chain_from = o.on;
    }
    g2s_session* on = f.on;
    if (on != s) on->rcache.swap(s->rcache);
    ResidentLaunch rl;
    const int rc = run_resident_queue(f.b, f.results, f.arena, true, &rl, chain_from);
    if (on != s) on->rcache.swap(s->rcache);
    if (rc < 0) return rc;
    if (rc != G2S_OK) break;  // (its fill kernel ran for nothing: the host path when it is ended)
    f.b->d3_queued = true; f.b->pre_units = rl.units; f.b->pre_two = rl.two_waves; f.b->pre_timed = rl.timed; f.b->pre_segw = rl.segw;
    f.chained = chain_from != nullptr;
  }
  return G2S_OK;
}
}  // namespace

extern "C" int g2s_fill_begin(g2s_session* s, const g2s_gap* gaps, size_t n, g2s_result* results, char* fill_arena, size_t arena_cap) {
  g2s_env_sync();
  if (!s || (!gaps && n) || (!results && n)) return fail(G2S_ERR_ARG, "g2s_fill_begin: bad argument");
  if (s->n_inflight >= G2S_MAX_IN_FLIGHT) return fail(G2S_ERR_ARG, "g2s_fill_begin: G2S_MAX_IN_FLIGHT lists are in flight already (g2s_fill_end first)");
  InternalCall own(s);
  g2s_session::InFlight f;
  f.results = results; f.arena = fill_arena; f.cap = arena_cap; f.gaps = gaps; f.n = n;
  const size_t group = team_group_for(s, n);
  if (n <= group && n > 0) {
    // the session, or a twin of it, that no list in flight is on (a list for g2s_fill_batch will run on the session itself)
    g2s_session* on = nullptr;
    for (int c = 0; c < G2S_MAX_IN_FLIGHT && !on; c++) {
      g2s_session* cand = c == 0 ? s : s->twins[c - 1];
      bool busy = false;
      for (int i = 0; i < s->n_inflight; i++) busy = busy || (s->inflight[i].b ? s->inflight[i].on == cand && cand != nullptr : c == 0);
      if (busy) continue;
      if (!cand) {
        const int rc = g2s_session_create(s->graph, s->device, &s->params, &s->twins[c - 1]);
        if (rc != G2S_OK) return rc;
        cand = s->twins[c - 1];
      }
      on = cand;
    }
    if (!on) return fail(G2S_ERR_STATE, "g2s_fill_begin: no session free");
    const auto t0 = std::chrono::steady_clock::now();
    int rc = g2s_batch_prepare(on, gaps, n, &f.b);
    if (rc != G2S_OK) return rc;
    f.ms_prepare = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (arena_cap < f.b->arena_bytes) { g2s_batch_free(f.b); return fail(G2S_ERR_ARG, "g2s_fill_begin: fill arena too small"); }
    f.on = on;
    f.b->arena = fill_arena;
    f.b->arena_base = 0;
    f.b->others_in_flight = s->n_inflight > 0;
    f.b->through_begin = true;
    ResidentLaunch rl;
    rc = resident_launch_fill(f.b, &rl, results, fill_arena);  // (1: not a list for resident mode — g2s_fill_end runs it on the host path)
    if (rc < 0) { g2s_batch_free(f.b); return rc; }
    if (rc == G2S_OK) {
      f.b->pre_launched = true; f.b->pre_units = rl.units; f.b->pre_two = rl.two_waves; f.b->pre_timed = rl.timed; f.b->pre_segw = rl.segw;
    }
  }  // (a list for the team pipeline, or an empty one: g2s_fill_end calls g2s_fill_batch)
  s->inflight[s->n_inflight++] = f;
  s->begun++;
  const int rs = inflight_settle(s);  // (phase D3 behind the fill kernel at once, where that can be)
  if (rs < 0) {  // (the list is not in flight then)
    s->n_inflight--;
    if (f.b) { if (f.on->d3_pending && f.b->d3_queued) ((D3Pending*)f.on->d3_pending)->discard = true; g2s_batch_free(f.b); }
    s->inflight[s->n_inflight] = g2s_session::InFlight();
    return rs;
  }
  return G2S_OK;
}
extern "C" int g2s_fill_end(g2s_session* s) {
  g2s_env_sync();
  if (!s || s->n_inflight < 1) return fail(G2S_ERR_ARG, "g2s_fill_end: no list in flight");
  InternalCall own(s);
  g2s_session::InFlight f = s->inflight[0];
  for (int i = 1; i < G2S_MAX_IN_FLIGHT; i++) s->inflight[i - 1] = s->inflight[i];
  s->inflight[G2S_MAX_IN_FLIGHT - 1] = g2s_session::InFlight();
  s->n_inflight--;
  int rc;
  bool on_device = false;
  if (!f.b) rc = g2s_fill_batch(s, f.gaps, f.n, f.results, f.arena, f.cap);
  else {
    g2s_session* on = f.on;
    if (on != s) on->rcache.swap(s->rcache);  // the one rand() stream (:178), wherever the list ends
    rc = g2s_batch_run(f.b, f.results, f.arena, f.cap);
    if (on != s) on->rcache.swap(s->rcache);
    on_device = rc == G2S_OK && f.b->timing.resident_launches > 0 && f.b->timing.resident_fallbacks == 0;
    g2s_batch_free(f.b);
    s->last_timing = on->last_timing;
    s->last_timing.ms_prepare = f.ms_prepare;
  }
  // The lists behind it that continued its stream on the device, and those that continued theirs: only good when
  // this one ended on the device.  Otherwise what their kernels wrote is dropped and they run again when they are ended
  // (g2s_batch_run: chain_broken), each from the host's generator, which then stands where it starts.
  if (!on_device)
    for (int i = 0; i < s->n_inflight; i++) {
      g2s_session::InFlight& y = s->inflight[i];
      if (!y.b || !y.chained || !y.b->d3_queued) break;
      y.b->chain_broken = true;
    }
  if (s->n_inflight >= 1) s->inflight[0].chained = false;
  if (rc >= 0) { const int rs = inflight_settle(s); if (rs < 0) return rs; }
  return rc;
}
extern "C" int g2s_fill_in_flight(const g2s_session* s) { return s ? s->n_inflight : 0; }

// ---- one list over several processes, a GPU and a share each (include/g2s.h): the three steps of team_resident_sharded
// with the exchanges between them left to the caller (gap2seq_amd/shard.py: gloo all-gathers of host scalars)
namespace {
void share_drop(g2s_session* s) {
  if (s->d3_pending) {
    (void)hipSetDevice(s->device);
    (void)hipStreamSynchronize(s->stream);
    (void)hipStreamSynchronize(s->stream2);
    delete (D3Pending*)s->d3_pending;
    s->d3_pending = nullptr;
    s->d_d3.clean = 0;
  }
  if (s->share_batch) { (void)resident_reset_fill(s, s->share_batch->jobs.size()); g2s_batch_free(s->share_batch); s->share_batch = nullptr; }
  s->share_step = 0;
  s->in_team_list = false; s->team_sharded = false; s->team_sessions = 1;
}
}  // namespace
extern "C" int g2s_share_begin(g2s_session* s, const g2s_gap* gaps, size_t n, g2s_result* results, char* fill_arena, size_t arena_cap,
                               uint64_t totals[2]) {
  g2s_env_sync();
  if (!s || !gaps || !results || !totals || n == 0) return fail(G2S_ERR_ARG, "g2s_share_begin: bad argument");
  REFUSE_IN_FLIGHT(s, "g2s_share_begin");
  if (s->share_step != 0) return fail(G2S_ERR_STATE, "g2s_share_begin: a share is open on this session (g2s_share_end it first)");
  if (gaps[0].skip_if_prev_right_fuz_gt >= 0) return fail(G2S_ERR_STATE, "g2s_share_begin: the share's first gap carries a skip rule (a record cut by a share boundary)");
  {
    void* d = nullptr;
    if (!device_pointer_of(results, &d) || (arena_cap && !device_pointer_of(fill_arena, &d)))
      return fail(G2S_ERR_ARG, "g2s_share_begin: results and fill_arena must be g2s_host_alloc memory (the kernels write them)");
  }
  s->in_team_list = true; s->team_sharded = true; s->team_sessions = 1;
  if (!resident_applicable(s, n) || s->d3_pending) { share_drop(s); return fail(G2S_ERR_STATE, "g2s_share_begin: not a list for resident mode on this session"); }
  memcpy(s->share_win, s->rcache.window(G2S_RAND_WINDOW), sizeof s->share_win);  // the generator where the LIST starts (every rank's is there)
  g2s_batch* b = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  int rc = g2s_batch_prepare(s, gaps, n, &b);
  if (rc != G2S_OK) { share_drop(s); return rc; }
  s->share_batch = b;
  b->timing.ms_prepare = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (arena_cap < b->arena_bytes) { share_drop(s); return fail(G2S_ERR_ARG, "g2s_share_begin: fill arena too small"); }
  b->arena = fill_arena;
  b->arena_base = 0;
  ResidentLaunch rl;
  rc = resident_launch_fill(b, &rl);
  if (rc == 1) { share_drop(s); return fail(G2S_ERR_STATE, "g2s_share_begin: not a list for resident mode"); }
  if (rc != G2S_OK) { share_drop(s); return rc; }
  s->share_timed = rl.timed; s->share_two = rl.two_waves;
  ResidentList L;
  L.groups.push_back(b);
  L.n = n; L.group_size = std::max<size_t>(n, 1);
  L.group_arena.push_back(0);
  L.arena_bytes = b->arena_bytes;
  L.outs_dev = (const GapOut*)s->d_outs.p;
  L.sub_dev = (const SubRec*)s->d_sub.p;
  L.sub_region = 0;
  L.pin = &s->h_d3;
  L.rnd_cap = b->rnd_cap; L.dmax = b->dmax; L.has_skip = b->has_skip;
  L.gaps_dev = (const GapDev*)s->d_gaps.p;
  rc = resident_d3_launch(s, L, rl.timed, false, results, fill_arena, true, true);
  if (rc == 1) { share_drop(s); return fail(G2S_ERR_STATE, "g2s_share_begin: not a list for phase D3 on the device"); }
  if (rc == G2S_OK && hipStreamSynchronize(s->stream) != hipSuccess) rc = fail(G2S_ERR_HIP, "g2s_share_begin: the share's first kernels");
  if (rc != G2S_OK) { share_drop(s); return rc; }
  const D3Summary* hs = ((D3Pending*)s->d3_pending)->hsum;
  if (hs->status != 0) { share_drop(s); return fail(G2S_ERR_STATE, "g2s_share_begin: a gap beyond every tier of the device, or tables beyond the budget"); }
  totals[0] = hs->draws_min; totals[1] = hs->draws_spread;
  s->share_step = 1;
  return G2S_OK;
}
extern "C" int g2s_share_tables(g2s_session* s, uint64_t base0, uint64_t R0, const uint32_t** fn) {
  g2s_env_sync();
  if (!s || !fn) return fail(G2S_ERR_ARG, "g2s_share_tables: bad argument");
  if (s->share_step != 1) return fail(G2S_ERR_STATE, "g2s_share_tables: no share begun on this session");
  if (base0 + R0 >= 0xF0000000ull) { share_drop(s); return fail(G2S_ERR_STATE, "g2s_share_tables: the list draws more than phase D3 on the device counts"); }
  int rc = resident_d3_sharded_tables(s, (uint32_t)base0, (uint32_t)R0, s->share_win);
  if (rc == G2S_OK && hipStreamSynchronize(s->stream) != hipSuccess) rc = fail(G2S_ERR_HIP, "g2s_share_tables: the share's tables");
  if (rc != G2S_OK) { share_drop(s); return rc; }
  if (((D3Pending*)s->d3_pending)->hsum->status != 0) { share_drop(s); return fail(G2S_ERR_STATE, "g2s_share_tables: tables beyond the budget"); }
  *fn = ((D3Pending*)s->d3_pending)->group_fn;
  s->share_step = 2;
  return G2S_OK;
}
extern "C" int g2s_share_trace(g2s_session* s, uint32_t d_in) {
  g2s_env_sync();
  if (!s) return fail(G2S_ERR_ARG, "g2s_share_trace: bad argument");
  if (s->share_step != 2) return fail(G2S_ERR_STATE, "g2s_share_trace: no tables on this session (g2s_share_tables first)");
  g2s_batch* b = s->share_batch;
  const size_t n = b->jobs.size();
  int rc = resident_d3_sharded_trace(s, d_in);
  bool fb = false;
  double ms_d3 = 0;
  if (rc == G2S_OK) rc = resident_d3_wait(s, &b->timing, &ms_d3, &fb);
  if (rc == G2S_OK) rc = resident_reset_fill(s, n);
  if (rc != G2S_OK) { share_drop(s); return rc; }
  if (fb) { share_drop(s); return fail(G2S_ERR_STATE, "g2s_share_trace: the share was not finished on the device"); }
  float ms_fill = 0;
  if (s->share_timed && hipEventElapsedTime(&ms_fill, s->ev[1], s->ev[2]) == hipSuccess) { b->timing.ms_fill_seg += ms_fill; b->timing.seg_timed_launches++; }
  (void)hipGetLastError();
  b->timing.seg_launches++;
  if (s->share_two) b->timing.seg2_launches++;
  b->timing.team_d3_sharded = 1;
  s->last_timing = b->timing;
  s->share_step = 3;
  return G2S_OK;
}
extern "C" int g2s_share_end(g2s_session* s, uint64_t list_draws) {
  if (!s) return fail(G2S_ERR_ARG, "g2s_share_end: bad argument");
  if (s->share_step != 3) { share_drop(s); return fail(G2S_ERR_STATE, "g2s_share_end: the share was not traced (dropped)"); }
  g2s_batch_free(s->share_batch);
  s->share_batch = nullptr;
  s->share_step = 0;
  s->in_team_list = false; s->team_sharded = false; s->team_sessions = 1;
  s->rcache.skip(list_draws);  // (every rank's generator moves past the whole list)
  return G2S_OK;
}

extern "C" int g2s_session_set_team(g2s_session* lead, g2s_session* const* helpers, int nhelpers, size_t group_size) {
  g2s_env_sync();
  if (!lead || nhelpers < 0 || (nhelpers && !helpers)) return fail(G2S_ERR_ARG, "g2s_session_set_team: bad argument");
  for (int i = 0; i < nhelpers; i++)
    if (!helpers[i] || helpers[i] == lead || helpers[i]->graph != lead->graph)
      return fail(G2S_ERR_ARG, "g2s_session_set_team: helpers must be other sessions on the same graph");
  REFUSE_IN_FLIGHT(lead, "g2s_session_set_team");
  lead->helpers.assign(helpers, helpers + nhelpers);
  lead->team_group = group_size;
  return G2S_OK;
}

// accessors used by the host driver (g2s_execute.cpp)
extern "C" const g2s_graph* g2s_session_graph(const g2s_session* s) { return s ? s->graph : nullptr; }
extern "C" int g2s_session_get_params(const g2s_session* s, g2s_params* out) {
  if (!s || !out) return fail(G2S_ERR_ARG, "g2s_session_get_params: bad argument");
  *out = s->params;
  return G2S_OK;
}

// TEST HOOK, see include/g2s_test.h: host half of phase D on a caller-supplied DP table.
// The closure the g2s_extract kernel would compute is derived on the host (host_closure).
extern "C" int g2s_test_post_gap(const g2s_graph* gh, const g2s_params* p, const g2s_gap* gap, int32_t n_states,
                                 const uint32_t* nodes, const int32_t* depths, const uint32_t* counts,
                                 int32_t c_count, int32_t n_lengths, const int32_t* lengths, int32_t reached_j,
                                 int32_t final_d, uint32_t seed, uint32_t skip, g2s_result* res, char* buf) {
  if (!gh || !p || !gap || !res || !buf || n_states < 0) return fail(G2S_ERR_ARG, "g2s_test_post_gap: bad argument");
  const Graph& g = *gh->g;
  const int k = g.k;
  GapJob j;
  j.g = gap->gap_len; j.lmf = gap->lmf; j.rmf = gap->rmf;
  if (gap->left_len < k + j.lmf || gap->right_len < k + j.rmf) return fail(G2S_ERR_ARG, "flank too short");
  std::vector<uint32_t> flank_store;
  j.resolve_on_host(g, gap->left, (size_t)gap->left_len, gap->right, (size_t)gap->right_len, &flank_store);
  HostTable t;
  t.D = j.lmf + j.rmf + j.g + p->d_err;
  t.lvl.assign((size_t)t.D + 2, 0);
  for (int i = 0; i < n_states; i++) if (depths[i] >= 0 && depths[i] <= t.D) t.lvl[(size_t)depths[i] + 1]++;
  for (int d = 0; d <= t.D; d++) t.lvl[(size_t)d + 1] += t.lvl[(size_t)d];
  t.states.assign((size_t)n_states + 1, 0);
  {
    std::vector<uint32_t> pos(t.lvl.begin(), t.lvl.end() - 1);
    for (int i = 0; i < n_states; i++)
      if (depths[i] >= 0 && depths[i] <= t.D)
        t.states[pos[(size_t)depths[i]]++] = ((uint64_t)nodes[i] << 32) | std::min<uint32_t>(counts[i], G2S_MAX_PATHS);
    for (int d = 0; d <= t.D; d++) std::sort(t.states.begin() + t.lvl[(size_t)d], t.states.begin() + t.lvl[(size_t)d + 1]);
  }
  GapOut go;
  memset(&go, 0, sizeof go);
  go.c_count = c_count; go.n_len = n_lengths; go.reached_j = reached_j; go.final_d = final_d;
  for (int i = 0; i < n_lengths && i < 2; i++) go.len[i] = lengths[i];
  FillParams fp;
  fp.k = k; fp.d_err = p->d_err; fp.skip_confident = p->skip_confident != 0; fp.all_paths = p->all_paths != 0;
  fp.unique_paths = p->unique_paths != 0;
  std::vector<SubState> closure;
  uint32_t q7 = 0;
  host_closure(g, fp, j, t, go, &closure, &q7);
  std::vector<SubRec> recs;
  std::vector<uint64_t> xps;
  sub_convert(closure.data(), (uint32_t)closure.size(), &recs, &xps);
  SubView v;
  v.out = &go; v.st = recs.data(); v.n = (uint32_t)recs.size(); v.xp = xps.data(); v.n_xp = (uint32_t)xps.size();
  SubPrep prep;
  sub_analyze(fp, j, v, &prep);
  memset(res, 0, sizeof *res);
  memset(buf, 0, j.buf_bytes(k, fp.d_err));
  res->phaseC_count = c_count;
  res->count = prep.count;
  res->flags |= prep.flags | q7;
  res->fill_off = (uint64_t)j.lmf;
  if (prep.phase_d) {
    GlibcRand rng;
    rng.seed(seed);
    for (uint32_t i = 0; i < skip; i++) rng.next();
    std::vector<uint32_t> rands((size_t)t.D + 4);
    for (auto& x : rands) x = (uint32_t)rng.next() << 1;  // raw words: value = word >> 1
    const int pick = (int)((rands[0] >> 1) % (uint32_t)go.n_len);
    const int fixed = sub_fixed_draws(v, prep, pick);
    sub_traceback(g, fp, j, v, prep, rands.data(), buf, res);
    if (fixed >= 0 && fixed != res->draws) return fail(G2S_ERR_STATE, "stop-depth analysis disagrees with the traceback");
    // the two shortcuts of the batch path, checked here on the CPU: the draw-count walk, and
    // the fill written without rand() values when no state has a second parent
    if (sub_count_draws(g, v, prep, rands.data()) != res->draws)
      return fail(G2S_ERR_STATE, "draw-count walk disagrees with the traceback");
    if (go.n_len == 1 && v.n_xp == 0 && fixed >= 0) {
      std::vector<char> buf2(j.buf_bytes(k, fp.d_err), 0);
      g2s_result r2;
      memset(&r2, 0, sizeof r2);
      sub_traceback(g, fp, j, v, prep, nullptr, buf2.data(), &r2);
      if (r2.draws != res->draws || r2.left_fuz != res->left_fuz || memcmp(buf2.data(), buf, buf2.size()) != 0)
        return fail(G2S_ERR_STATE, "rand()-free traceback disagrees with the traceback");
    }
    res->vertices = prep.sub[0]; res->edges = prep.sub[1]; res->nontrivial_components = prep.sub[2];
    res->size_nontrivial_components = prep.sub[3]; res->vertices_final = prep.sub[4]; res->edges_final = prep.sub[5];
    res->fill_off = (uint64_t)(j.lmf - res->left_fuz);
    res->fill_len = (int32_t)strlen(buf + res->fill_off);
  }
  return G2S_OK;
}

extern "C" int g2s_test_post_closure(const g2s_graph* gh, const g2s_params* p, const g2s_gap* gap, uint32_t n_records,
                                     const uint32_t* records, uint32_t n_xp, const uint64_t* xp, int32_t c_count,
                                     int32_t n_lengths, const int32_t* lengths, int32_t reached_j, int32_t final_d,
                                     uint32_t seed, uint64_t skip, g2s_result* res, char* buf) {
  if (!gh || !p || !gap || !res || !buf || (n_records && !records) || (n_xp && !xp))
    return fail(G2S_ERR_ARG, "g2s_test_post_closure: bad argument");
  const Graph& g = *gh->g;
  const int k = g.k;
  GapJob j;
  j.g = gap->gap_len; j.lmf = gap->lmf; j.rmf = gap->rmf;
  if (gap->left_len < k + j.lmf || gap->right_len < k + j.rmf) return fail(G2S_ERR_ARG, "flank too short");
  std::vector<uint32_t> flank_store;
  j.resolve_on_host(g, gap->left, (size_t)gap->left_len, gap->right, (size_t)gap->right_len, &flank_store);
  GapOut go;
  memset(&go, 0, sizeof go);
  go.c_count = c_count; go.n_len = n_lengths; go.reached_j = reached_j; go.final_d = final_d;
  for (int i = 0; i < n_lengths && i < 2; i++) go.len[i] = lengths[i];
  FillParams fp;
  fp.k = k; fp.d_err = p->d_err; fp.skip_confident = p->skip_confident != 0; fp.all_paths = p->all_paths != 0;
  fp.unique_paths = p->unique_paths != 0;
  std::vector<SubRec> recs(n_records);
  if (n_records) memcpy(recs.data(), records, (size_t)n_records * sizeof(SubRec));
  std::vector<uint64_t> xps(xp, xp + n_xp);
  std::sort(xps.begin(), xps.end());
  SubView v;
  v.out = &go; v.st = recs.data(); v.n = n_records; v.xp = xps.data(); v.n_xp = n_xp;
  SubPrep prep;
  sub_analyze(fp, j, v, &prep);
  memset(res, 0, sizeof *res);
  memset(buf, 0, j.buf_bytes(k, fp.d_err));
  res->phaseC_count = c_count;
  res->n_lengths = n_lengths;
  for (int i = 0; i < n_lengths && i < 2; i++) res->lengths[i] = lengths[i];
  res->count = prep.count;
  res->flags |= prep.flags;
  res->fill_off = (uint64_t)j.lmf;
  if (prep.phase_d) {
    GlibcRand rng;
    rng.seed(seed);
    for (uint64_t i = 0; i < skip; i++) rng.next();
    const int D = j.lmf + j.rmf + j.g + p->d_err;
    std::vector<uint32_t> rands((size_t)D + 4);
    for (auto& x : rands) x = (uint32_t)rng.next() << 1;  // raw words: value = word >> 1
    sub_traceback(g, fp, j, v, prep, rands.data(), buf, res);
    res->vertices = prep.sub[0]; res->edges = prep.sub[1]; res->nontrivial_components = prep.sub[2];
    res->size_nontrivial_components = prep.sub[3]; res->vertices_final = prep.sub[4]; res->edges_final = prep.sub[5];
    res->fill_off = (uint64_t)(j.lmf - res->left_fuz);
    res->fill_len = (int32_t)strlen(buf + res->fill_off);
  }
  return G2S_OK;
}

extern "C" int g2s_test_post_segments(const g2s_graph* gh, const g2s_params* p, const g2s_gap* gap, uint32_t n_segs,
                                      const uint32_t* segs, int32_t c_count, int32_t n_lengths, const int32_t* lengths,
                                      int32_t reached_j, int32_t final_d, uint32_t seed, uint64_t skip, g2s_result* res,
                                      char* buf, int32_t* on_segments) {
  if (!gh || !p || !gap || !res || !buf || (n_segs && !segs)) return fail(G2S_ERR_ARG, "g2s_test_post_segments: bad argument");
  const Graph& g = *gh->g;
  const int k = g.k;
  GapJob j;
  j.g = gap->gap_len; j.lmf = gap->lmf; j.rmf = gap->rmf;
  if (gap->left_len < k + j.lmf || gap->right_len < k + j.rmf) return fail(G2S_ERR_ARG, "flank too short");
  std::vector<uint32_t> flank_store;
  j.resolve_on_host(g, gap->left, (size_t)gap->left_len, gap->right, (size_t)gap->right_len, &flank_store);
  GapOut go;
  memset(&go, 0, sizeof go);
  go.c_count = c_count; go.n_len = n_lengths; go.reached_j = reached_j; go.final_d = final_d;
  for (int i = 0; i < n_lengths && i < 2; i++) go.len[i] = lengths[i];
  FillParams fp;
  fp.k = k; fp.d_err = p->d_err; fp.skip_confident = p->skip_confident != 0; fp.all_paths = p->all_paths != 0;
  fp.unique_paths = p->unique_paths != 0;
  SubView v;
  v.out = &go; v.segs = (const SegRec*)segs; v.n_segs = n_segs;
  SubPrep prep;
  const auto t_an0 = std::chrono::steady_clock::now();
  const bool ok = seg_analyze(fp, j, v, &prep);
  if (on_segments) *on_segments = ok ? 1 : 0;
  if (n_segs >= 2000 && prep.run_mode && GENV("G2S_POST_LAPS")) {  // (tools/host_items_replay.py)
    const double* l = g2s_post_laps;
    const double t1 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    fprintf(stderr, "[g2s] %u segments, run analysis laps (us): front %.0f | collect %.0f merge+sort %.0f runs %.0f edges %.0f csr %.0f tarjan %.0f rest %.0f | all %.0f\n", n_segs,
            l[0] - std::chrono::duration<double, std::micro>(t_an0.time_since_epoch()).count(), l[1] - l[0], l[2] - l[1], l[3] - l[2], l[4] - l[3], l[5] - l[4], l[6] - l[5], l[7] - l[6],
            t1 - std::chrono::duration<double, std::micro>(t_an0.time_since_epoch()).count());
  }
  memset(res, 0, sizeof *res);
  memset(buf, 0, j.buf_bytes(k, fp.d_err));
  if (!ok) return G2S_OK;  // a k-mer at two depths: the caller takes g2s_test_seg_expand + g2s_test_post_closure
  res->phaseC_count = c_count;
  res->n_lengths = n_lengths;
  for (int i = 0; i < n_lengths && i < 2; i++) res->lengths[i] = lengths[i];
  res->count = prep.count;
  res->flags |= prep.flags;
  res->fill_off = (uint64_t)j.lmf;
  if (prep.phase_d) {
    GlibcRand rng;
    rng.seed(seed);
    for (uint64_t i = 0; i < skip; i++) rng.next();
    const int D = j.lmf + j.rmf + j.g + p->d_err;
    std::vector<uint32_t> rands((size_t)D + 4);
    for (auto& x : rands) x = (uint32_t)rng.next() << 1;
    const int pick = (int)((rands[0] >> 1) % (uint32_t)go.n_len);
    const int fixed = sub_fixed_draws(v, prep, pick);
    seg_traceback(g, fp, j, v, prep, rands.data(), buf, res);
    if (fixed >= 0 && fixed != res->draws) return fail(G2S_ERR_STATE, "stop-depth analysis disagrees with the traceback (segments)");
    if (seg_count_draws(g, v, prep, rands.data()) != res->draws)
      return fail(G2S_ERR_STATE, "draw-count walk disagrees with the traceback (segments)");
    if (go.n_len == 1 && !prep.has_choice && fixed >= 0) {
      std::vector<char> buf2(j.buf_bytes(k, fp.d_err), 0);
      g2s_result r2;
      memset(&r2, 0, sizeof r2);
      seg_traceback(g, fp, j, v, prep, nullptr, buf2.data(), &r2);
      if (r2.draws != res->draws || r2.left_fuz != res->left_fuz || memcmp(buf2.data(), buf, buf2.size()) != 0)
        return fail(G2S_ERR_STATE, "rand()-free traceback disagrees with the traceback (segments)");
    }
    res->vertices = prep.sub[0]; res->edges = prep.sub[1]; res->nontrivial_components = prep.sub[2];
    res->size_nontrivial_components = prep.sub[3]; res->vertices_final = prep.sub[4]; res->edges_final = prep.sub[5];
    res->fill_off = (uint64_t)(j.lmf - res->left_fuz);
    res->fill_len = (int32_t)strlen(buf + res->fill_off);
  }
  return G2S_OK;
}

extern "C" int g2s_test_seg_expand(const g2s_graph* gh, const g2s_params* p, const g2s_gap* gap, uint32_t n_segs,
                                   const uint32_t* segs, int32_t n_lengths, const int32_t* lengths, int32_t reached_j,
                                   uint32_t n_records, uint32_t* records, uint32_t n_xp, uint64_t* xp) {
  if (!gh || !p || !gap || (n_segs && !segs) || (n_records && !records) || (n_xp && !xp))
    return fail(G2S_ERR_ARG, "g2s_test_seg_expand: bad argument");
  const Graph& g = *gh->g;
  const int k = g.k;
  GapJob j;
  j.g = gap->gap_len; j.lmf = gap->lmf; j.rmf = gap->rmf;
  if (gap->left_len < k + j.lmf || gap->right_len < k + j.rmf) return fail(G2S_ERR_ARG, "flank too short");
  std::vector<uint32_t> flank_store;
  j.resolve_on_host(g, gap->left, (size_t)gap->left_len, gap->right, (size_t)gap->right_len, &flank_store);
  GapOut go;
  memset(&go, 0, sizeof go);
  go.n_len = n_lengths; go.reached_j = reached_j;
  for (int i = 0; i < n_lengths && i < 2; i++) go.len[i] = lengths[i];
  FillParams fp;
  fp.k = k; fp.d_err = p->d_err; fp.skip_confident = p->skip_confident != 0; fp.all_paths = p->all_paths != 0;
  fp.unique_paths = p->unique_paths != 0;
  const SegRec* sr = (const SegRec*)segs;
  size_t total = 0, nx = 0;
  for (uint32_t i = 0; i < n_segs; i++) {
    total += sr[i].depth_len >> 16;
    if (!(sr[i].flags & G2S_SUB_SOURCE)) {
      int np = ((sr[i].par01 & 0xFFFFu) != 0xFFFFu) + ((sr[i].par01 >> 16) != 0xFFFFu) + ((sr[i].par23 & 0xFFFFu) != 0xFFFFu) +
               ((sr[i].par23 >> 16) != 0xFFFFu);
      if (np > 1) nx += (size_t)np - 1;
    }
  }
  if (total != n_records || nx != n_xp) return fail(G2S_ERR_ARG, "g2s_test_seg_expand: output sizes do not match the segments");
  seg_expand(fp, j, go, sr, n_segs, (SubRec*)records, xp);
  return G2S_OK;
}

extern "C" int g2s_test_graph_tables(const g2s_graph* gh, uint32_t* succ_out, uint64_t* ustart_out) {
  if (!gh || !gh->g) return fail(G2S_ERR_ARG, "g2s_test_graph_tables: bad argument");
  const Graph& g = *gh->g;
  if (succ_out) memcpy(succ_out, g.succ.data(), (size_t)g.n * 8 * sizeof(uint32_t));
  if (ustart_out) memcpy(ustart_out, g.ustart.data(), (size_t)((g.n + 63) / 64) * 8);
  return G2S_OK;
}

extern "C" int64_t g2s_graph_validate(const g2s_graph* gr, char* msg, size_t msg_cap) {
  if (!gr || !gr->g) return -1;
  const Graph& g = *gr->g;
  std::string text;
  int64_t bad = 0;
  auto note = [&](const char* what, uint64_t i) {
    if (bad++ < 8) text += std::string(what) + " at k-mer " + std::to_string(i) + "; ";
  };
  auto only = [&](uint32_t v, uint32_t want) {  // the single valid successor of v is `want`
    int cnt = 0;
    bool hit = false;
    for (int nt = 0; nt < 4; nt++) {
      const uint32_t w = g.succ_of(v, nt);
      if (w == kInvalidNode) continue;
      cnt++;
      hit = hit || w == want;
    }
    return cnt == 1 && hit;
  };
  uint64_t starts = 0;
  for (uint64_t i = 0; i < g.n; i++) {
    const bool start = (g.ustart[i >> 6] >> (i & 63)) & 1;
    starts += start;
    if (i == 0) { if (!start) note("first k-mer is not a unitig start", i); continue; }
    if (start) continue;
    const uint32_t a = (uint32_t)(2 * (i - 1)), b = (uint32_t)(2 * i);
    if (!only(a, b)) note("internal edge is not the only successor (even orientation)", i);
    if (!only(b ^ 1u, a ^ 1u)) note("internal edge is not the only successor (odd orientation)", i);
  }
  if (starts != g.n_unitigs) note("bitmap population differs from the unitig count", starts);
  for (uint64_t v = 0; v < 2 * g.n; v++)
    for (int nt = 0; nt < 4; nt++) {
      const uint32_t w = g.succ_of((uint32_t)v, nt);
      if (w == kInvalidNode) continue;
      if (w >= 2 * g.n) { note("successor out of range", v >> 1); continue; }
      if (g.lastnt[w] != nt) note("last base of a successor differs from its slot", v >> 1);
    }
  if (msg && msg_cap) {
    const size_t c = std::min(text.size(), msg_cap - 1);
    memcpy(msg, text.data(), c);
    msg[c] = 0;
  }
  return bad;
}

extern "C" int g2s_test_worker_pool(int32_t threads, int32_t rounds, int32_t n) {
  if (threads < 1 || rounds < 0 || n < 0) return fail(G2S_ERR_ARG, "g2s_test_worker_pool: bad argument");
  WorkerPool pool(threads - 1);
  std::vector<std::atomic<int>> hits((size_t)n);
  for (int r = 0; r < rounds; r++) {
    for (auto& h : hits) h.store(0);
    std::atomic<long long> sum(0);
    const int nn = (r % 7 == 0) ? std::min(n, 3) : n;  // rounds shorter than the pool: idle workers arrive late
    pool.run((size_t)nn, [&](size_t i) { hits[i].fetch_add(1); sum.fetch_add((long long)i + 1); });
    if (sum.load() != (long long)nn * (nn + 1) / 2) return fail(G2S_ERR_STATE, "worker pool: wrong task sum");
    for (int i = 0; i < nn; i++)
      if (hits[(size_t)i].load() != 1) return fail(G2S_ERR_STATE, "worker pool: task not run exactly once");
  }
  return G2S_OK;
}

extern "C" int g2s_test_group_queue(int32_t nworkers, uint64_t n, uint64_t group_size, int32_t* owner) {
  return g2s_test_group_queue_slow(nworkers, n, group_size, -1, 0, owner);
}
extern "C" int g2s_test_group_queue_slow(int32_t nworkers, uint64_t n, uint64_t group_size, int32_t slow_worker,
                                         uint32_t slow_us, int32_t* owner) {
  if (nworkers < 1 || !owner) return fail(G2S_ERR_ARG, "g2s_test_group_queue: bad argument");
  for (uint64_t i = 0; i < n; i++) owner[i] = -1;
  GroupQueue queue((size_t)n, (size_t)group_size);
  std::atomic<int> clash(0);
  auto worker = [&](int t) {
    size_t gi = 0, off = 0, cnt = 0;
    while (queue.pull(&gi, &off, &cnt)) {
      for (size_t i = off; i < off + cnt; i++) { if (owner[i] != -1) clash.fetch_add(1); owner[i] = t; }
      // a group takes every worker 20 us (its "launch"); the slow one slow_us on top
      std::this_thread::sleep_for(std::chrono::microseconds(20 + (t == slow_worker ? slow_us : 0u)));
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nworkers; t++) th.emplace_back(worker, t);
  worker(0);
  for (auto& x : th) x.join();
  if (clash.load()) return fail(G2S_ERR_STATE, "group queue: a gap was handed out twice");
  for (uint64_t i = 0; i < n; i++) if (owner[i] < 0) return fail(G2S_ERR_STATE, "group queue: a gap was never handed out");
  return G2S_OK;
}

extern "C" int g2s_test_device_rand(int device, uint32_t seed, uint64_t skip, uint32_t n, int32_t* out) {
  if (!out) return fail(G2S_ERR_ARG, "g2s_test_device_rand: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(G2S_ERR_NO_DEVICE, "no such device");
  HIP_TRY(hipSetDevice(device));
  GlibcRandStream st;
  st.seed(seed);
  for (uint64_t left = skip; left;) {  // the host's generator up to the position, as a session would have consumed it
    const size_t c = (size_t)std::min<uint64_t>(left, 1u << 20);
    st.ensure(c);
    st.consume(c);
    left -= c;
  }
  std::vector<uint32_t> tables((128 + 256 + 64) * 31);
  rand_tables_host(tables.data(), tables.data() + 128 * 31, tables.data() + (128 + 256) * 31);
  const uint64_t cap = ((uint64_t)n + 2 * G2S_RAND_BLOCK) & ~(uint64_t)(G2S_RAND_BLOCK - 1);
  uint32_t *d_tab = nullptr, *d_rnd = nullptr;
  D3Summary* d_sum = nullptr;
  HIP_TRY(hipMalloc((void**)&d_tab, tables.size() * 4));
  HIP_TRY(hipMalloc((void**)&d_rnd, (31 + cap + 64) * 4));
  HIP_TRY(hipMalloc((void**)&d_sum, sizeof(D3Summary)));
  D3Summary sum;
  memset(&sum, 0, sizeof sum);
  sum.draws_min = n;
  HIP_TRY(hipMemcpy(d_tab, tables.data(), tables.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_sum, &sum, sizeof sum, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_rnd, st.window(G2S_RAND_WINDOW), G2S_RAND_WINDOW * 4, hipMemcpyHostToDevice));
  RandTables rt;
  rt.hi = d_tab; rt.mid = d_tab + 128 * 31; rt.lane = d_tab + (128 + 256) * 31;
  HIP_TRY(launch_rand_fill(nullptr, d_rnd, rt, d_sum, cap));
  HIP_TRY(hipDeviceSynchronize());
  std::vector<uint32_t> h(n);
  HIP_TRY(hipMemcpy(h.data(), d_rnd + 31, (size_t)n * 4, hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < n; i++) out[i] = (int32_t)(h[i] >> 1);
  (void)hipFree(d_tab); (void)hipFree(d_rnd); (void)hipFree(d_sum);
  return G2S_OK;
}

// TEST HOOK: the n values behind `skip` values that were drawn elsewhere — GlibcRandStream::skip, the jump by the
// recurrence's polynomial g2s_share_end moves a rank's generator with — for comparison with g2s_test_rand_stream
extern "C" int g2s_test_rand_skip(uint32_t seed, uint64_t skip, uint32_t n, int32_t* out) {
  if (!out) return fail(G2S_ERR_ARG, "g2s_test_rand_skip: bad argument");
  GlibcRandStream st;
  st.seed(seed);
  st.skip(skip);
  st.ensure(n);
  for (uint32_t i = 0; i < n; i++) out[i] = st.value(i);
  return G2S_OK;
}
extern "C" int g2s_test_rand_stream(uint32_t seed, uint32_t skip, uint32_t n, int32_t* out) {
  if (!out) return fail(G2S_ERR_ARG, "g2s_test_rand_stream: bad argument");
  GlibcRandStream st;
  st.seed(seed);
  // consume in two pieces so that the compaction path is exercised too
  st.ensure(skip / 2 + 1);
  st.consume(skip / 2);
  st.ensure(skip - skip / 2 + n);
  st.consume(skip - skip / 2);
  for (uint32_t i = 0; i < n; i++) out[i] = st.value(i);
  return G2S_OK;
}
