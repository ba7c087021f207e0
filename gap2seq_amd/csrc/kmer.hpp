// gap2seq_amd/csrc/kmer.hpp — 2-bit k-mer arithmetic with GATB-core's codec.
//
// Replaces the k-mer model behind gatb Graph::buildNode / Node::kmer used at
// /root/reference/src/Gap2Seq.cpp:879,955,995,1084,1114 (GATB-core 1.4.1 is
// un-vendored; semantics from SURVEY.md Appendix B.1-B.3):
//   code(c) = (c>>1)&3  -> A 0, C 1, T 2, G 3 (any byte maps to some base)
//   invalid(c) = (c>>3)&1 -> true for N/n, used only when counting k-mers
//   first base most significant; complement = code ^ 2
//   canonical = numeric min(forward, revcomp); strand 0 iff forward < revcomp
//   64-bit words for k <= 31, 128-bit for 32 <= k <= 63.
#pragma once
#include <cstdint>
#include <string>

namespace g2s {

typedef unsigned __int128 u128;

static inline int nt_code(char c) { return (c >> 1) & 3; }
static inline bool nt_invalid(char c) { return ((c >> 3) & 1) != 0; }
static const char kNtChar[4] = {'A', 'C', 'T', 'G'};

static inline uint64_t revcomp32(uint64_t x) {  // all 32 bases of a word
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = __builtin_bswap64(x);
  return x ^ 0xAAAAAAAAAAAAAAAAULL;
}

template <class KT>
struct KmerOps;

template <>
struct KmerOps<uint64_t> {
  static uint64_t mask(int k) { return k >= 32 ? ~0ULL : ((1ULL << (2 * k)) - 1); }
  static uint64_t revcomp(uint64_t x, int k) { return revcomp32(x) >> (64 - 2 * k); }
  static uint64_t hash(uint64_t z) {
    z = (z ^ (z >> 33)) * 0xff51afd7ed558ccdULL;
    z = (z ^ (z >> 33)) * 0xc4ceb9fe1a85ec53ULL;
    return z ^ (z >> 33);
  }
};

template <>
struct KmerOps<u128> {
  static u128 mask(int k) { return k >= 64 ? ~(u128)0 : (((u128)1 << (2 * k)) - 1); }
  static u128 revcomp(u128 x, int k) {
    u128 y = ((u128)revcomp32((uint64_t)x) << 64) | (u128)revcomp32((uint64_t)(x >> 64));
    return y >> (128 - 2 * k);
  }
  static uint64_t hash(u128 x) {
    return KmerOps<uint64_t>::hash((uint64_t)x) ^ KmerOps<uint64_t>::hash((uint64_t)(x >> 64) ^ 0x9e3779b97f4a7c15ULL);
  }
};

// Rolling forward/reverse encoder over a sequence.
template <class KT>
struct KmerRoller {
  int k;
  KT mask, fwd = 0, rev = 0;
  int valid = 0;
  explicit KmerRoller(int kk) : k(kk), mask(KmerOps<KT>::mask(kk)) {}
  // returns true when a full valid k-mer ends at this character
  bool push(char c) {
    if (nt_invalid(c)) { valid = 0; fwd = 0; rev = 0; return false; }
    KT code = (KT)nt_code(c);
    fwd = ((fwd << 2) | code) & mask;
    rev = (rev >> 2) | ((code ^ 2) << (2 * (k - 1)));
    return ++valid >= k;
  }
  KT canonical() const { return fwd < rev ? fwd : rev; }
};

// buildNode: encode the first k chars regardless of validity.
template <class KT>
static inline void encode_kmer(const char* s, int k, KT* canon, int* strand) {
  KT f = 0;
  for (int i = 0; i < k; i++) f = (f << 2) | (KT)nt_code(s[i]);
  KT r = KmerOps<KT>::revcomp(f, k);
  if (f < r) { *canon = f; *strand = 0; } else { *canon = r; *strand = 1; }
}

template <class KT>
static inline std::string decode_kmer(KT canon, int strand, int k) {
  KT seq = strand == 0 ? canon : KmerOps<KT>::revcomp(canon, k);
  std::string s((size_t)k, 'A');
  for (int i = k - 1; i >= 0; i--) { s[(size_t)i] = kNtChar[(int)(seq & 3)]; seq >>= 2; }
  return s;
}

}  // namespace g2s
