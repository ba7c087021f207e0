// gap2seq_amd/csrc/envcache.hpp — the switches the library reads from the environment (G2S_*: measurements, tests, the
// tools' instrumented runs), without a walk through the environment per look.  A 500-gap list's step read ~50 of them —
// getenv() is a linear search over every variable the process holds: 0.2-0.5 us a look on the pool's boxes, 6-10 us of a
// step in front of the fill kernel's launch.  Here a switch has a slot (registered at its first look), the slots' values
// are read once, and read again only when the environment has CHANGED — the tests switch settings between two calls of
// one process (setenv / unsetenv replace or move the entries' pointers: the cheap fingerprint below sees it).
//   GENV("G2S_X")      at a call site: the variable's value or nullptr, as getenv would give it
//   g2s_env_sync()     once at every entry point of the ABI that may look at a switch
#pragma once
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

extern char** environ;

namespace g2s {

struct EnvCache {
  static constexpr int CAP = 192;
  const char* names[CAP];
  std::atomic<const char*> values[CAP];
  std::atomic<int> n{0};
  std::atomic<unsigned long long> print{0};
  std::mutex mu;
  static unsigned long long fingerprint() {
    unsigned long long h = 1469598103934665603ull;
    size_t c = 0;
    if (environ)
      for (char** e = environ; *e; e++, c++) { h ^= (unsigned long long)(uintptr_t)*e; h *= 1099511628211ull; }
    return h ^ ((unsigned long long)c << 48) ^ (unsigned long long)(uintptr_t)environ;
  }
  int slot(const char* name) {
    std::lock_guard<std::mutex> lk(mu);
    const int k = n.load();
    for (int i = 0; i < k; i++) if (!strcmp(names[i], name)) return i;
    if (k >= CAP) return -1;
    names[k] = name;
    values[k].store(getenv(name));
    n.store(k + 1);
    return k;
  }
  void sync() {
    const unsigned long long f = fingerprint();
    if (f == print.load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> lk(mu);
    const int k = n.load();
    for (int i = 0; i < k; i++) values[i].store(getenv(names[i]));
    print.store(f, std::memory_order_release);
  }
};
inline EnvCache& env_cache() { static EnvCache c; return c; }
inline const char* env_value(int slot, const char* name) { return slot < 0 ? getenv(name) : env_cache().values[slot].load(std::memory_order_relaxed); }

}  // namespace g2s

#define GENV(name) ([]() -> const char* { static const int slot_ = g2s::env_cache().slot(name); return g2s::env_value(slot_, name); }())
inline void g2s_env_sync() { g2s::env_cache().sync(); }
