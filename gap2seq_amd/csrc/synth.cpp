// gap2seq_amd/csrc/synth.cpp — seeded synthetic workloads of BASELINE.md /
// random genome (+ planted repeats, + second haplotype) and
// one-gap-per-record scaffolds in GapCutter's shape
// (/root/reference/src/GapCutter.cpp:193,203: flanks of k+fuz bases around an N run).
// PRNG: xoshiro256** seeded through SplitMix64; no dependence on libc/Python RNGs.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/g2s.h"
#include "fastx.hpp"

namespace {

struct Rng {
  uint64_t s[4];
  explicit Rng(uint64_t seed) {
    for (int i = 0; i < 4; i++) {
      seed += 0x9e3779b97f4a7c15ULL;
      uint64_t z = seed;
      z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
      z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
      s[i] = z ^ (z >> 31);
    }
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  // uniform in [lo, hi]
  uint64_t range(uint64_t lo, uint64_t hi) { return lo + next() % (hi - lo + 1); }
};

char* dup_text(const std::string& s) {
  char* p = (char*)malloc(s.size() + 1);
  if (p) memcpy(p, s.c_str(), s.size() + 1);
  return p;
}

}  // namespace

extern "C" int g2s_synth_genome(uint64_t length, uint32_t variant, uint64_t seed, char** reads_fasta) {
  if (!reads_fasta || length < 1000) return G2S_ERR_ARG;
  static const char kNt[4] = {'A', 'C', 'G', 'T'};
  Rng rng(seed);
  std::string g((size_t)length, 'A');
  for (uint64_t i = 0; i < length; i += 32) {
    uint64_t w = rng.next();
    for (uint64_t j = i; j < std::min(length, i + 32); j++) { g[(size_t)j] = kNt[w & 3]; w >>= 2; }
  }
  if (variant & 1u) {  // V1: exact repeats, 200-2000 bp, 2-5 copies each
    Rng r1(seed + 1);
    const uint64_t nrep = std::max<uint64_t>(1, length / 10000);
    for (uint64_t r = 0; r < nrep; r++) {
      const uint64_t len = r1.range(200, std::min<uint64_t>(2000, length / 8));
      const uint64_t src = r1.range(0, length - len);
      const uint64_t copies = r1.range(2, 5);
      const std::string unit = g.substr((size_t)src, (size_t)len);
      for (uint64_t c = 1; c < copies; c++) {
        const uint64_t dst = r1.range(0, length - len);
        memcpy(&g[(size_t)dst], unit.data(), (size_t)len);
      }
    }
  }
  std::string out;
  g2s::append_fasta(&out, "hap1", g);
  if (variant & 2u) {  // V2: second haplotype, one substitution every ~500 bp
    Rng r2(seed + 3);
    std::string h = g;
    for (uint64_t p = 250; p + 100 < length; p += 500) {
      const uint64_t q = p + r2.range(0, 200) - 100;
      char c;
      do { c = kNt[r2.next() & 3]; } while (c == h[(size_t)q]);
      h[(size_t)q] = c;
    }
    g2s::append_fasta(&out, "hap2", h);
  }
  *reads_fasta = dup_text(out);
  return *reads_fasta ? G2S_OK : G2S_ERR_NOMEM;
}

extern "C" int g2s_synth_gaps(const char* reads_fasta, int k, int fuz, int ngaps, int min_len, int max_len,
                              uint64_t seed, char** scaffolds_fasta) {
  if (!reads_fasta || !scaffolds_fasta || k < 1 || fuz < 0 || ngaps < 0 || min_len < 1 || max_len < min_len)
    return G2S_ERR_ARG;
  std::vector<g2s::FastxRecord> recs;
  g2s::parse_fastx(std::string(reads_fasta), &recs);
  if (recs.empty()) return G2S_ERR_ARG;
  const std::string& g = recs[0].seq;
  const uint64_t flank = (uint64_t)(k + fuz);
  if (g.size() < 2 * flank + (uint64_t)max_len + 2) return G2S_ERR_ARG;
  Rng rng(seed);
  std::string out;
  for (int i = 0; i < ngaps; i++) {
    const uint64_t len = rng.range((uint64_t)min_len, (uint64_t)max_len);
    const uint64_t start = rng.range(flank, g.size() - len - flank);
    std::string seq = g.substr((size_t)(start - flank), (size_t)flank);
    seq.append((size_t)len, 'N');
    seq += g.substr((size_t)(start + len), (size_t)flank);
    g2s::append_fasta(&out, "g" + std::to_string(i) + " len=" + std::to_string(len), seq);
  }
  *scaffolds_fasta = dup_text(out);
  return *scaffolds_fasta ? G2S_OK : G2S_ERR_NOMEM;
}
