// gap2seq_amd/csrc/sync_debug.h — race-hunting builds of the kernels (tests/test_gpu_resident.py, tools/race_hunt.sh).
// Every kernel of the fill path is hand-synchronised code: wave-local waits for the LDS queue (lds_sync), workgroup
// barriers, spin-waits across streams.  A missing or misplaced synchronisation shows only when the waves' relative
// timing happens to expose it — round 5's missing barrier in g2s_fill_segw's tail lost a traceback start once in a
// hundred runs and had passed two rounds of green suites.  Two instrumented builds of the SAME sources make that
// timing move, and their results must be the normal build's, bit for bit, on every path:
//   -DG2S_JITTER         at every synchronisation point a wave sleeps for a pseudo-random time (a hash of the point,
//                        the wave, the workgroup and the clock): 0 at seven points in eight, else 1 .. 16 k cycles —
//                        the waves of a workgroup, and the workgroups of a launch, drift apart and meet in other orders
//   -DG2S_PARANOID_SYNC  every wave-local wait becomes a wait for ALL outstanding memory operations (vector memory
//                        and LDS) plus a wave barrier, every workgroup barrier waits the same way and is taken twice:
//                        a result that changes says some point relied on less than it waited for
// Without either macro this header changes nothing.  Include it in front of the other device headers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#if defined(G2S_JITTER) || defined(G2S_PARANOID_SYNC)
#define G2S_SYNC_DEBUG 1
__device__ __forceinline__ void g2s_sync_jitter(uint32_t site) {
#ifdef G2S_JITTER
  const uint32_t clk = (uint32_t)__builtin_amdgcn_s_memtime();
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t h = clk ^ (site * 0x9E3779B1u) ^ (wave * 0x85EBCA6Bu) ^ ((uint32_t)blockIdx.x * 0xC2B2AE35u);
  h ^= h >> 13; h *= 0x5BD1E995u; h ^= h >> 15;
  h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
  if ((h & 7u) == 0u)
    for (uint32_t i = 0, n = ((h >> 3) & 15u) + 1u; i < n; i++) __builtin_amdgcn_s_sleep(16);  // 16 x 64 cycles a step
#else
  (void)site;
#endif
}
__device__ __forceinline__ void g2s_wait_all() {
#ifdef G2S_PARANOID_SYNC
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
}
// __syncthreads() as the HIP headers define it (release fence, s_barrier, acquire fence), behind the jitter / the wait
__device__ __forceinline__ void g2s_syncthreads_debug(uint32_t site) {
  g2s_sync_jitter(site);
  g2s_wait_all();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef G2S_PARANOID_SYNC
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}
#define __syncthreads() g2s_syncthreads_debug((uint32_t)__LINE__)
#endif
