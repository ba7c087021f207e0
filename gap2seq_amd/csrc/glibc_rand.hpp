// gap2seq_amd/csrc/glibc_rand.hpp — glibc's rand()/srand() reproduced so the
// traceback draws (/root/reference/src/Gap2Seq.cpp:178,1440,1513) do not depend
// on process-global libc state and can be owned by a session.
// Algorithm: glibc stdlib/random_r.c, TYPE_3 (x^31 + x^3 + 1 additive feedback),
// seeded by a 16807 Lehmer sequence, first 310 outputs discarded.
#pragma once
#include <cstdint>

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

}  // namespace g2s
