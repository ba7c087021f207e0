/* include/g2s_test.h — TEST HOOKS of libg2s_hip.so, kept apart from the reference-facing C ABI (include/g2s.h).
 *
 * Nothing here is a fill path (none of these can compute the DP) and nothing of the reference binds to them: they let
 * the unit tests (tests/test_host.py, tests/test_seg_model.py, tests/test_shard.py, tests/test_gpu_*.py) drive single
 * pieces of the host side — the host half of phase D on caller-supplied tables, the graph tables, the rand() stream
 * of the host and of the device, the worker pool, the shared group counter — without a kernel run. */
#ifndef G2S_TEST_H_
#define G2S_TEST_H_
#include "g2s.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------
 *  TEST HOOK (CPU unit tests of the host half of phase D only; not a fill
 *  path: it cannot compute the DP).  Runs D1/D2/D3 for ONE gap on a DP table
 *  supplied by the caller: n_states states (oriented node, depth, count) plus
 *  the phase C outcome.  The rand() stream is srand(seed) advanced by `skip`
 *  draws.  `buf` needs gap_len + k + d_err + lmf + rmf + 3 bytes.
 * ------------------------------------------------------------------------ */
int g2s_test_post_gap(const g2s_graph* g, const g2s_params* p, const g2s_gap* gap, int32_t n_states,
                      const uint32_t* nodes, const int32_t* depths, const uint32_t* counts, int32_t c_count,
                      int32_t n_lengths, const int32_t* lengths, int32_t reached_j, int32_t final_d, uint32_t seed,
                      uint32_t skip, g2s_result* res, char* buf);

/* TEST HOOK: the host half of phase D (D2 + D3) on a backward closure supplied by the caller in
 * the layout the kernels emit: n records of 16 bytes {node, count, depth | flags << 27, first
 * parent index | G2S more-parents bit, or -1} in an order in which every parent comes AFTER its
 * children, plus the side list of further parents (state << 32 | parent).  Used by the CPU
 * tests of the kernel's algorithm model (tests/seg_model.py); cannot compute the DP. */
int g2s_test_post_closure(const g2s_graph* g, const g2s_params* p, const g2s_gap* gap, uint32_t n_records,
                          const uint32_t* records /* 4 words each */, uint32_t n_xp, const uint64_t* xp, int32_t c_count,
                          int32_t n_lengths, const int32_t* lengths, int32_t reached_j, int32_t final_d, uint32_t seed,
                          uint64_t skip, g2s_result* res, char* buf);

/* TEST HOOK: the host half of phase D run directly on closure segments (the segment tier's output,
 * layout as for g2s_test_seg_expand), as the batch path does when no k-mer occurs at two depths of
 * the closure; *on_segments = 0 when that does not hold (nothing is computed then: take
 * g2s_test_seg_expand + g2s_test_post_closure, as the batch path does). */
int g2s_test_post_segments(const g2s_graph* g, const g2s_params* p, const g2s_gap* gap, uint32_t n_segs,
                           const uint32_t* segs, int32_t c_count, int32_t n_lengths, const int32_t* lengths,
                           int32_t reached_j, int32_t final_d, uint32_t seed, uint64_t skip, g2s_result* res, char* buf,
                           int32_t* on_segments);

/* TEST HOOK: the host's expansion of a closure given as unitig segments (what the segment tier's
 * kernel emits: 8 words per segment {node, depth | len << 16, count, ts | tt << 16, parents 0-1,
 * parents 2-3, flags, 0}, children before parents) into the per-state records and side list of
 * g2s_test_post_closure.  n_records / n_xp must be the exact output sizes. */
int g2s_test_seg_expand(const g2s_graph* g, const g2s_params* p, const g2s_gap* gap, uint32_t n_segs,
                        const uint32_t* segs, int32_t n_lengths, const int32_t* lengths, int32_t reached_j,
                        uint32_t n_records, uint32_t* records, uint32_t n_xp, uint64_t* xp);

/* TEST HOOK: copies of the tables the kernels walk: the successor table (2 * kmers * 4 words,
 * G2S_INVALID_NODE = none) and the unitig-start bitmap ((kmers + 63) / 64 words; bit i set = the
 * edge 2(i-1) -> 2i is not unitig-internal). */
int g2s_test_graph_tables(const g2s_graph* g, uint32_t* succ_out, uint64_t* ustart_out);

/* TEST HOOK: values [skip, skip+n) of the session-style rand() stream after srand(seed)
 * (the flat glibc TYPE_3 generator the tracebacks read), for comparison with libc. */
int g2s_test_rand_stream(uint32_t seed, uint32_t skip, uint32_t n, int32_t* out);

/* TEST HOOK: the same values reached by a JUMP over the `skip` values in front (the recurrence's polynomial: what
 * g2s_share_end moves a rank's generator with when the values were drawn on other ranks' devices). */
int g2s_test_rand_skip(uint32_t seed, uint64_t skip, uint32_t n, int32_t* out);

/* TEST HOOK: the same values from the DEVICE's generator (d3_device.hip: g2s_rand_fill — the state behind `skip`
 * values handed over by the host, every block of 4096 values reached with three jump polynomials). */
int g2s_test_device_rand(int device, uint32_t seed, uint64_t skip, uint32_t n, int32_t* out);

/* TEST HOOK: the host worker pool that runs the per-gap analysis and tracebacks: `rounds`
 * parallel-for rounds of `n` tasks on `threads` threads (task i adds i+1 to a per-round
 * sum); returns G2S_OK when every task of every round ran exactly once. */
int g2s_test_worker_pool(int32_t threads, int32_t rounds, int32_t n);

/* TEST HOOK: the shared group counter g2s_team_fill's sessions pull from, with `nworkers`
 * host threads in place of sessions: owner[i] receives the worker that was handed gap i.
 * G2S_OK when every gap of [0, n) was handed out exactly once, in contiguous groups. */
int g2s_test_group_queue(int32_t nworkers, uint64_t n, uint64_t group_size, int32_t* owner);
/* The same with one worker that needs slow_us microseconds more for every group it takes than the others (a busy
 * or slower device): what it does not get to is taken by the others — the queue hands a group to whoever asks. */
int g2s_test_group_queue_slow(int32_t nworkers, uint64_t n, uint64_t group_size, int32_t slow_worker, uint32_t slow_us,
                              int32_t* owner);

#ifdef __cplusplus
}
#endif
#endif /* G2S_TEST_H_ */
