// gap2seq_amd/csrc/glibc_rand.hpp — glibc's rand()/srand() reproduced so the
// traceback draws (/root/reference/src/Gap2Seq.cpp:178,1440,1513) do not depend
// on process-global libc state and can be owned by a session.
// Algorithm: glibc stdlib/random_r.c, TYPE_3 (x^31 + x^3 + 1 additive feedback),
// seeded by a 16807 Lehmer sequence, first 310 outputs discarded.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace g2s {

class GlibcRand {
 public:
  GlibcRand() { seed(1); }
  void seed(uint32_t s) {
    if (s == 0) s = 1;
    int64_t w = (int32_t)s;
    state_[0] = (uint32_t)w;
    for (int i = 1; i < 31; i++) {
      w = (16807 * w) % 2147483647;  // same value as glibc's Schrage form
      if (w < 0) w += 2147483647;
      state_[i] = (uint32_t)w;
    }
    front_ = 3;
    rear_ = 0;
    for (int i = 0; i < 310; i++) next();
  }
  int next() {
    state_[front_] += state_[rear_];
    const uint32_t out = state_[front_] >> 1;
    front_ = front_ == 30 ? 0 : front_ + 1;
    rear_ = rear_ == 30 ? 0 : rear_ + 1;
    return (int)out;
  }

 private:
  uint32_t state_[31];
  int front_, rear_;
};

// The same stream materialised into a flat array, many values at a time.  glibc's ring
// update r[f] += r[f-3] is the lagged Fibonacci recurrence x[n] = x[n-31] + x[n-3]; laid
// out flat (e[0..30] = seeded state rotated by 3, e[n] = e[n-31] + e[n-3]) three values per
// step are independent, so this runs several times faster than calling next() in a loop.
class GlibcRandStream {
 public:
  GlibcRandStream() {}
  ~GlibcRandStream() { free(e_); }
  GlibcRandStream(const GlibcRandStream&) = delete;
  GlibcRandStream& operator=(const GlibcRandStream&) = delete;
  void swap(GlibcRandStream& o) { std::swap(e_, o.e_); std::swap(size_, o.size_); std::swap(cap_, o.cap_); std::swap(first_, o.first_); }
  void seed(uint32_t s) {
    if (s == 0) s = 1;
    uint32_t r[31];
    int64_t w = (int32_t)s;
    r[0] = (uint32_t)w;
    for (int i = 1; i < 31; i++) {
      w = (16807 * w) % 2147483647;
      if (w < 0) w += 2147483647;
      r[i] = (uint32_t)w;
    }
    reserve(31 + 310 + 4096);
    size_ = 31;
    for (int i = 0; i < 31; i++) e_[i] = r[(i + 3) % 31];
    extend(310);          // srand() discards the first 310 outputs
    first_ = 31 + 310;    // e_[first_ + k] >> 1 is the k-th value rand() returns
  }
  // make the next `upto` values of the stream available
  void ensure(size_t upto) { if (first_ + upto > size_) extend(first_ + upto - size_ + 4096); }
  bool would_grow(size_t upto) const { return first_ + upto > size_; }
  int32_t value(size_t k) const { return (int32_t)(e_[first_ + k] >> 1); }
  // raw words of the upcoming values: value k = raw()[k] >> 1
  const uint32_t* raw() const { return e_ + first_; }
  // words [first value - 31, first value - 31 + words) of the flat stream: the generator's state, then the next values
  const uint32_t* window(size_t words) {
    if (words > 31) ensure(words - 31);
    return e_ + first_ - 31;
  }
  // n values were consumed elsewhere (on the device); state31 = the 31 words in front of the next value
  void jump(size_t n, const uint32_t state31[31]) {
    if (first_ + n <= size_) { consume(n); return; }
    reserve(31 + 4096);
    memcpy(e_, state31, 31 * sizeof(uint32_t));
    size_ = 31;
    first_ = 31;
  }
  // n values were consumed elsewhere and nobody kept the state behind them: the recurrence is linear, so the state n
  // values on is a fixed combination (x^n modulo the recurrence's polynomial) of the 61 words from the state here
  void skip(uint64_t n) {
    if (first_ + n <= size_) { consume((size_t)n); return; }
    ensure(31);
    uint32_t q[31], st[31];
    jump_poly(n, q);
    const uint32_t* e = e_ + first_ - 31;
    for (int j = 0; j < 31; j++) {
      uint32_t acc = 0;
      for (int i = 0; i < 31; i++) acc += q[i] * e[j + i];
      st[j] = acc;
    }
    reserve(31 + 4096);
    memcpy(e_, st, sizeof st);
    size_ = 31;
    first_ = 31;
  }
  // drop the first n values (they were consumed)
  void consume(size_t n) {
    first_ += n;
    if (first_ > (1u << 22) && first_ <= size_) {  // keep the array from growing without bound
      const size_t keep_from = first_ - 31;
      memmove(e_, e_ + keep_from, (size_ - keep_from) * sizeof(uint32_t));
      size_ -= keep_from;
      first_ = 31;
    }
  }
 private:
  void reserve(size_t n) {
    if (n <= cap_) return;
    size_t want = n + n / 2;
    e_ = (uint32_t*)realloc(e_, want * sizeof(uint32_t));  // no zero fill: every word is written below
    cap_ = want;
  }
  // x^N modulo x^31 - x^28 - 1 with coefficients modulo 2^32: e[m + N] = sum_i q[i] e[m + i] for every m
  // (the recurrence is linear), which lets a thread start in the middle of the stream
  static void jump_poly(uint64_t N, uint32_t q[31]) {
    auto mulmod = [](const uint32_t* a, const uint32_t* b, uint32_t* out) {
      uint32_t c[61];
      memset(c, 0, sizeof c);
      for (int i = 0; i < 31; i++) {
        if (!a[i]) continue;
        for (int j = 0; j < 31; j++) c[i + j] += a[i] * b[j];
      }
      for (int d = 60; d >= 31; d--) { c[d - 3] += c[d]; c[d - 31] += c[d]; }  // x^d = x^(d-3) + x^(d-31)
      memcpy(out, c, 31 * sizeof(uint32_t));
    };
    uint32_t r[31], base[31], t[31];
    memset(r, 0, sizeof r);
    memset(base, 0, sizeof base);
    r[0] = 1;
    base[1] = 1;
    for (; N; N >>= 1) {
      if (N & 1) { mulmod(r, base, t); memcpy(r, t, sizeof t); }
      mulmod(base, base, t);
      memcpy(base, t, sizeof t);
    }
    memcpy(q, r, 31 * sizeof(uint32_t));
  }
  // e[from .. to) by the recurrence; the 31 words in front of `from` are there
  static void fill_range(uint32_t* e, size_t from, size_t to) {
    // the three most recent values stay in registers: every load is 31 elements behind the
    // stores, so the loop runs at load/store throughput instead of store-forwarding latency
    if (to <= from) return;
    uint32_t a = e[from - 3], b = e[from - 2], c = e[from - 1];
    size_t i = from;
    for (; i + 3 <= to; i += 3) {
      a += e[i - 31]; e[i] = a;
      b += e[i - 30]; e[i + 1] = b;
      c += e[i - 29]; e[i + 2] = c;
    }
    for (; i < to; i++) e[i] = e[i - 31] + e[i - 3];
  }
  void extend(size_t n) {
    const size_t old = size_;
    reserve(old + n);
    size_ = old + n;
    uint32_t* e = e_;
    // Millions of values (a 10 000-gap list draws 6.5 M: 4.5 ms on one thread, more than the GPU needs for the
    // whole list): several threads, each starting from the 31 words in front of its block, which it computes
    // from the words in front of the whole extension with the jump polynomial of its offset.
    const size_t kParallelFrom = (size_t)1 << 20;
    if (n >= kParallelFrom) {
      const int T = (int)std::min<size_t>(8, std::max<size_t>(2, std::min<size_t>(n >> 19, std::thread::hardware_concurrency())));
      const size_t B = n / (size_t)T;
      fill_range(e, old, old + 30);  // e[old - 31 .. old + 30): what every jump reads
      std::vector<std::thread> th;
      for (int t = 1; t < T; t++)
        th.emplace_back([=]() {
          const size_t start = old + (size_t)t * B, end = t == T - 1 ? old + n : old + (size_t)(t + 1) * B - 31;
          uint32_t q[31];
          jump_poly((uint64_t)t * B, q);
          for (int j = 0; j < 31; j++) {  // the 31 words in front of this block
            uint32_t acc = 0;
            for (int i = 0; i < 31; i++) acc += q[i] * e[old - 31 + (size_t)j + (size_t)i];
            e[start - 31 + (size_t)j] = acc;
          }
          fill_range(e, start, end);
        });
      fill_range(e, old + 30, old + B - 31);  // (block 0; the 31 words behind it are block 1's start)
      for (auto& x : th) x.join();
      return;
    }
    // the three most recent values stay in registers: every load is 31 elements behind the
    // stores, so the loop runs at load/store throughput instead of store-forwarding latency
    uint32_t a = e[old - 3], b = e[old - 2], c = e[old - 1];
    size_t i = old;
    for (; i + 3 <= old + n; i += 3) {
      a += e[i - 31]; e[i] = a;
      b += e[i - 30]; e[i + 1] = b;
      c += e[i - 29]; e[i + 2] = c;
    }
    for (; i < old + n; i++) e[i] = e[i - 31] + e[i - 3];
  }
  uint32_t* e_ = nullptr;
  size_t size_ = 0, cap_ = 0, first_ = 31;
};

}  // namespace g2s
