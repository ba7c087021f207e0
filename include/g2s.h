/* ===========================================================================
 *  include/g2s.h — C ABI of the MI355X-native Gap2Seq-core fill path.
 *
 *  This is the drop-in boundary.  The reference has no FFI of its own: the hot
 *  path is the in-process C++ member
 *      int Gap2Seq::fill_gap(const Graph&, const std::string& kmer_left,
 *                            const std::string& kmer_right, int gap_len, int k,
 *                            int gap_err, int left_max_fuz, int right_max_fuz,
 *                            int* left_fuz, int* right_fuz, long long max_mem,
 *                            char* fill, bool skip_confident, bool all_paths,
 *                            subgraph_stats*)
 *  (/root/reference/src/Gap2Seq.hpp:74-76, Gap2Seq.cpp:858-1556), called once
 *  per gap from Gap2Seq::execute() (Gap2Seq.cpp:252 and :380) on a GATB Graph
 *  built once (Gap2Seq.cpp:193-219).  Each entry point below names the piece of
 *  that interface it replaces.  INTEGRATION.md shows the stub a maintainer of
 *  the reference would add to call it.
 *
 *  Plain C: opaque handles, caller-owned buffers, integer status codes, no
 *  exceptions cross this boundary.  One g2s_session drives one GPU; sessions
 *  on different devices may run on different host threads.
 *  There is NO CPU fallback: every fill entry point returns G2S_ERR_NO_DEVICE
 *  when no gfx950 device is usable.
 * ======================================================================== */
#ifndef G2S_H_
#define G2S_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G2S_ABI_VERSION 6

/* status codes */
#define G2S_OK 0
#define G2S_ERR_ARG (-1)
#define G2S_ERR_IO (-2)
#define G2S_ERR_NO_DEVICE (-3)
#define G2S_ERR_HIP (-4)
#define G2S_ERR_NOMEM (-5)
#define G2S_ERR_STATE (-6)

/* Gap2Seq.cpp:38 */
#define G2S_MAX_PATHS (2147483647 / 2 - 1)
#define G2S_INVALID_NODE 0xFFFFFFFFu

typedef struct g2s_graph g2s_graph;     /* replaces gatb Graph (host + device copies) */
typedef struct g2s_session g2s_session; /* one device, one rand() stream, reusable workspaces */
typedef struct g2s_batch g2s_batch;     /* a prepared list of gaps resident in HBM */

int g2s_abi_version(void);
/* Thread-local text of the last error returned on this thread. */
const char* g2s_last_error(void);

/* ---------------------------------------------------------------------------
 *  Graph: replaces Graph::create(BankAlbum, "-kmer-size k -abundance-min solid
 *  ...") / Graph::load (Gap2Seq.cpp:193-219).  Exact solid canonical k-mer set
 *  (GATB codec A0 C1 T2 G3, canonical = min(fwd, revcomp), k-mers containing
 *  N/n skipped), numbered in unitig order, with a 4-slot successor table per
 *  oriented node in GATB enumeration order (A,C,T,G).  Predecessors are read
 *  from the same table: pred(v)[i] = succ(v^1)[i]^1.
 *  The build itself (k-mer sort, successor table, unitig numbering) runs on
 *  the GPU named by the environment variable G2S_DEVICE (default 0) when there
 *  is one, and the graph then already resides on that device; without a
 *  device, for even k, or with G2S_HOST_BUILD=1 it runs on `nthreads` host
 *  threads.  Either way the same graph results, up to the numbering of nodes.
 * ------------------------------------------------------------------------ */
int g2s_graph_build_files(const char* reads_csv, int k, int solid, int nthreads, g2s_graph** out);
int g2s_graph_build_seqs(const char* const* seqs, const uint64_t* lens, int nseqs, int k, int solid,
                         int nthreads, g2s_graph** out);
/* Own binary cache (the reference reuses "<reads>.h5", Gap2Seq.cpp:171,195-197). */
int g2s_graph_save(const g2s_graph* g, const char* path);
int g2s_graph_load(const char* path, g2s_graph** out);
void g2s_graph_free(g2s_graph* g);
int g2s_graph_k(const g2s_graph* g);
/* -solid the set was built with (0: a cache file written before this was recorded). */
int g2s_graph_solid(const g2s_graph* g);
uint64_t g2s_graph_num_kmers(const g2s_graph* g);
uint64_t g2s_graph_num_unitigs(const g2s_graph* g);
/* graph.buildNode + graph.contains: oriented node id (2*index + strand) of the
 * first k characters of `kmer`, or G2S_INVALID_NODE when it is not solid. */
uint32_t g2s_graph_node(const g2s_graph* g, const char* kmer);
/* graph.successors / graph.predecessors in GATB order; returns the count. */
int g2s_graph_successors(const g2s_graph* g, uint32_t node, uint32_t out[4]);
int g2s_graph_predecessors(const g2s_graph* g, uint32_t node, uint32_t out[4]);
/* graph.toString(node): writes k chars + NUL. */
int g2s_graph_node_string(const g2s_graph* g, uint32_t node, char* out);
/* Copy the successor table into the HBM of `device` (idempotent per device). */
int g2s_graph_upload(g2s_graph* g, int device);
/* Bytes resident in HBM for this graph on `device` (0 when not uploaded). */
uint64_t g2s_graph_device_bytes(const g2s_graph* g, int device);

/* ---------------------------------------------------------------------------
 *  Parameters: the Gap2Seq-core options that reach fill_gap
 *  (Gap2Seq.cpp:164-175).
 * ------------------------------------------------------------------------ */
typedef struct g2s_params {
  int32_t d_err;          /* -dist-error (gap_err)                       */
  int32_t skip_confident; /* -all-upper                                  */
  int32_t all_paths;      /* !-best-only                                 */
  int32_t unique_paths;   /* -unique (applied by the caller of fill_gap) */
  int64_t max_mem;        /* bytes per gap, already divided by threads
                             (Gap2Seq.cpp:170,302); device-budget analogue */
  uint32_t randseed;      /* -randseed: srand() argument; 0 = time(NULL) (Gap2Seq.cpp:178) */
  int32_t host_threads;   /* threads for the per-gap host post-process; 0 = all */
} g2s_params;

/* One fill_gap call's inputs (Gap2Seq.cpp:380-383 / :252-254). */
typedef struct g2s_gap {
  const char* left;   /* kmer_left : k+lmf chars (longer allowed, as in -left)  */
  const char* right;  /* kmer_right: k+rmf chars                                */
  int32_t left_len;
  int32_t right_len;
  int32_t gap_len;
  int32_t lmf;        /* left_max_fuz  */
  int32_t rmf;        /* right_max_fuz */
  /* Scaffold scanning continues at i += right_fuz after a filled gap
   * (Gap2Seq.cpp:402,415), which can make the NEXT gap of the same record
   * ineligible.  If >= 0: skip this gap (status G2S_GAP_SKIPPED, no rand()
   * draws) when the previous gap of the batch was filled with right_fuz
   * greater than this value.  -1: independent. */
  int32_t skip_if_prev_right_fuz_gt;
} g2s_gap;

/* g2s_result.flags */
#define G2S_GAP_SKIPPED 0x1        /* see skip_if_prev_right_fuz_gt                     */
#define G2S_GAP_Q7 0x2             /* both strands of one k-mer met: result is
                                      well defined but outside the bit-exact claim        */
#define G2S_GAP_MEM_EXCEEDED 0x4   /* count == -1                                         */
#define G2S_GAP_BACKTRACE_FAIL 0x8 /* "Unable to backtrace!" (Gap2Seq.cpp:1493-1510)      */
#define G2S_GAP_BAD_FLANK 0x10     /* flank shorter than k+fuz (reference would throw)    */
#define G2S_GAP_PHASE_D 0x20       /* phase D ran: fill / left_fuz / right_fuz written    */

typedef struct g2s_result {
  int32_t count;      /* fill_gap return value: >0 paths (saturating), 0, or -1 */
  int32_t left_fuz;
  int32_t right_fuz;
  uint32_t flags;
  /* `fill` as the reference leaves it: the useful text starts at
   * buf[lmf - left_fuz]; here fill_off points at exactly that position inside
   * the caller's arena and fill_len = strlen(&buf[lmf-left_fuz]) (it INCLUDES
   * the right k-mer, Gap2Seq.cpp:856-857).  NUL terminated. */
  uint64_t fill_off;
  int32_t fill_len;
  int32_t draws;      /* rand() calls consumed */
  /* subgraph_stats (Gap2Seq.hpp:50-58); valid when count > 0 and !skip_confident */
  uint64_t vertices, edges, nontrivial_components, size_nontrivial_components, vertices_final, edges_final;
  /* phase C result before the D1 recount */
  int32_t phaseC_count;
  int32_t n_lengths;
  int32_t lengths[2];
  /* G2S_GAP_BACKTRACE_FAIL: the two numbers of the reference's "Unable to backtrace!" line (Gap2Seq.cpp:1494:
   * currentD2, currentD); g2s_backtrace_text() makes the line.  (ABI 3 carried the 96-byte text in every record:
   * 1.9 MB of a 10 000-gap list's 7.7 MB over the link, for a line no gap of the test lists ever printed.) */
  int32_t backtrace_depth, backtrace_final_d;
  int32_t reserved[2];    /* the record is 112 bytes: seven 16-byte stores of the trace kernel */
} g2s_result;

/* Lists in flight on one GPU, for a caller that streams lists (Gap2Seq-core -stream-gaps; the reference prints a gap
 * when it is done and goes on, Gap2Seq.cpp:385,426-431).  g2s_fill_begin() takes a list as g2s_fill_batch() does,
 * queues its kernels and returns; g2s_fill_end() finishes the OLDEST list begun (results and fill text in that list's
 * buffers, which must stay valid until then) and returns what g2s_fill_batch() would.  At most G2S_MAX_IN_FLIGHT lists
 * may be begun and not ended: the younger ones' kernels then run while the oldest one's results cross the link and the
 * host prepares the next.  Lists end in the order they were begun and draw from the session's one rand() stream in
 * that order: results are identical to g2s_fill_batch() list by list.  g2s_fill_in_flight(): lists begun, not ended.
 * While lists are in flight the session's buffers and its rand() stream belong to them: g2s_fill_batch,
 * g2s_batch_prepare, g2s_batch_run, g2s_team_fill (with this session in the team), g2s_session_set_team,
 * g2s_session_srand and g2s_session_skip_draws return G2S_ERR_STATE and do nothing until every list has been ended
 * (ABI 5; g2s_session_destroy may be called at any time: it waits for the lists' kernels and drops them). */
#define G2S_MAX_IN_FLIGHT 3
int g2s_fill_begin(g2s_session* s, const g2s_gap* gaps, size_t n, g2s_result* results, char* fill_arena, size_t arena_cap);
int g2s_fill_end(g2s_session* s);
int g2s_fill_in_flight(const g2s_session* s);

/* The reference's line for a gap flagged G2S_GAP_BACKTRACE_FAIL (Gap2Seq.cpp:1494): "Unable to backtrace! <currentD2>
 * <currentD> <toString(reachedTarget)>", without the newline; the k-mer is right[right_fuz .. right_fuz + k) in upper
 * case (the node was built from that text, Gap2Seq.cpp:1113).  Returns the length, or 0 when the gap is not flagged. */
size_t g2s_backtrace_text(const g2s_gap* gap, const g2s_result* r, int k, char* out, size_t cap);

/* Per-batch measurements (bench.py). */
typedef struct g2s_timing {
  double ms_right_bfs;    /* kernel g2s_right_bfs (HBM tier), HIP events on the session stream */
  double ms_left_dp;      /* kernel g2s_left_dp  (HBM tier frontier kernel)         */
  double ms_extract;      /* kernel g2s_extract  (HBM tier)                         */
  double ms_d2h;
  double ms_host_post;    /* SCC / branch rule / traceback on the host              */
  double ms_total;        /* wall time of g2s_batch_run                             */
  uint64_t xA, sA, xB, sB, xD, sD; /* expansions / newly set states per phase, counted on device */
  uint64_t flank_bytes;   /* sum over gaps of (k+lmf)+(k+rmf)                       */
  uint64_t fill_bytes;    /* sum over gaps of fill_len                              */
  uint32_t launches_left_dp; /* >1 when overflowing gaps were retried with larger tables */
  uint32_t retried_gaps;
  /* LDS tier (kernels g2s_fill_lds = phases A-C fused, g2s_extract_lds = D1) */
  double ms_fill_lds;
  double ms_extract_lds;
  uint64_t x_fill_lds;       /* expansions (A+B) of the gaps that completed in the LDS tier */
  uint64_t s_fill_lds;       /* states set (A+B) by those gaps */
  uint32_t lds_tier_gaps;    /* gaps that completed in the LDS tier */
  uint32_t lds_launches;     /* launches of g2s_fill_lds (2 when a second pass with larger LDS tables ran) */
  uint32_t log_pool_gaps;    /* gaps whose state log moved to a chunk of the launch's log pool */
  uint32_t rs_pool_gaps;     /* gaps whose right set moved from LDS to a chunk of the launch's spill pool */
  double ms_prepare;         /* g2s_fill_batch / g2s_team_fill: flank k-mer -> node resolution + descriptor upload
                                (g2s_batch_prepare), inside ms_total; summed over sessions for a team */
  /* segment tier (kernel g2s_fill_seg = phases A-D1 over unitig segments, fill_seg.hip) */
  double ms_fill_seg;        /* HIP events on the session stream, summed over the launches that were bracketed with
                                events: all on the host path, one in eight in resident mode (seg_timed_launches;
                                G2S_KERNEL_TIMING=all times every launch) */
  uint32_t seg_tier_gaps;    /* gaps that completed in the segment tier */
  uint32_t seg_launches;
  uint64_t seg_segments;     /* segments those gaps took (a config-2 gap: ~25 for ~1000 DP states) */
  double ms_fill_segx;       /* the tier's large variant (g2s_fill_segx): gaps that outgrow the LDS-resident capacities */
  uint32_t segx_tier_gaps;
  uint32_t segx_launches;
  uint32_t watchdog_gaps;    /* gaps on which a probe loop of the large variant ran past its bound (a defect; expected 0) */
  uint32_t seg2_launches;    /* segment-tier launches that ran two waves per gap (g2s_fill_seg2: short lists) */
  /* resident mode: the whole list on the device, phase D3 included (d3_device.hip) */
  double ms_d3;              /* kernels of phase D3 behind the fill kernel (scan, rand() stream, tables, chain, tracebacks) */
  uint32_t resident_launches;   /* lists (groups) finished on the device */
  uint32_t resident_fallbacks;  /* lists the device gave back to the host path */
  uint64_t draw_dependent_gaps; /* gaps whose number of rand() draws depends on the values drawn */
  uint64_t d3_table_entries;    /* entries of the draw-count tables that resolved their offsets */
  uint32_t host_finished_gaps;  /* gaps of resident lists whose closure the host analysed and traced (a k-mer at two depths) */
  /* g2s_team_fill: the dispatcher's groups and which session took how many (sessions beyond the 16th are not listed) */
  uint32_t team_groups, team_sessions;
  uint32_t team_groups_by_session[16];
  uint32_t seg_timed_launches;  /* segment-tier launches whose duration is in ms_fill_seg (and, resident mode, in ms_d3) */
  uint32_t team_d3_sharded;     /* g2s_team_fill: every session traced its own group and wrote its own results (phase D3 sharded) */
  uint32_t traced_in_fill_gaps; /* (ABI 6) gaps of resident lists whose fill kernel's wave wrote fill text and record itself: a traceback
                                   with one path length and no choice between parents writes the same whatever rand() returns */
  /* ...and, per session (the first 16), what its group took: fill kernel(s) and phase D3 kernels by HIP events (timed
   * launches only), and the wall time of the session's thread from the call to its last result */
  double team_ms_fill[16], team_ms_d3[16], team_ms_wall[16];
  /* (ABI 6) g2s_fill_batch on one session, resident mode: where the host's time inside the call goes, in microseconds:
   * [0] entry -> fill kernel queued (preparation and launch set-up: the device idles until then), [1] -> phase D3
   * queued, [2] -> the device's hand-over seen (waiting), [3] -> the host-finished gaps done, [4] -> stream
   * synchronised, [5] -> return.  Zero when the list took another path. */
  double host_us[8];
  /* (ABI 6) ... and whose fill kernel's wave wrote a GUESS of a traceback that has choices (first path length, first
   * parent at every choice); of those gaps' text, in groups of 64 bases as the trace kernel's first waves compared them:
   * all / sent through the link again because the real traceback differs there */
  uint32_t guessed_in_fill_gaps, guessed_groups, guessed_groups_resent, reserved1;
} g2s_timing;

/* ---------------------------------------------------------------------------
 *  Session: owns the device, the stream, the reusable workspaces and the
 *  glibc-compatible rand() stream that the traceback consumes in gap order.
 * ------------------------------------------------------------------------ */
int g2s_session_create(g2s_graph* g, int device, const g2s_params* p, g2s_session** out);
void g2s_session_destroy(g2s_session* s);
/* srand(seed) (Gap2Seq.cpp:178).  G2S_OK, or G2S_ERR_STATE with lists in flight (ABI 5: void before). */
int g2s_session_srand(g2s_session* s, uint32_t seed);
/* Discard the next n values of the session's rand() stream: what n calls of rand() by the
 * caller between two fill_gap calls would do to the reference's libc stream.  Returns as g2s_session_srand. */
int g2s_session_skip_draws(g2s_session* s, uint64_t n);

/* Resolve flank k-mers to node ids on the host and upload the gap descriptors
 * to HBM.  Replaces the argument marshalling of Gap2Seq.cpp:380-383. */
int g2s_batch_prepare(g2s_session* s, const g2s_gap* gaps, size_t n, g2s_batch** out);
/* The hot path: phases A-D of fill_gap for every gap of the batch (kernels on
 * the session stream, device->host of the state logs, host SCC/branch rule and
 * the in-order traceback).  `fill_arena` receives the NUL-terminated fills;
 * it needs sum(gap_len + k + d_err + lmf + rmf + 3) bytes (g2s_batch_arena_bytes). */
int g2s_batch_run(g2s_batch* b, g2s_result* results, char* fill_arena, size_t arena_cap);
size_t g2s_batch_arena_bytes(const g2s_batch* b);
int g2s_batch_timing(const g2s_batch* b, g2s_timing* out);
void g2s_batch_free(g2s_batch* b);
/* Measurements of the last g2s_fill_batch / g2s_batch_run on this session (ms_total = wall
 * time of that call, preparation included for g2s_fill_batch). */
int g2s_session_last_timing(const g2s_session* s, g2s_timing* out);
/* prepare + run + free. */
int g2s_fill_batch(g2s_session* s, const g2s_gap* gaps, size_t n, g2s_result* results, char* fill_arena,
                   size_t arena_cap);

/* ---------------------------------------------------------------------------
 *  Several sessions on one gap list = the reference's dispatcher
 *  (`-nb-cores`, Gap2Seq.cpp:300-304: threads pulling scaffolds from a shared
 *  iterator), with GPUs in place of threads.  The list is cut into groups of
 *  `group_size` gaps (0 = default); each session's host thread pulls the next
 *  group from a shared counter and runs phases A-D2 for it; the rand()
 *  dependent part (D3) then runs once, in gap order, on sessions[0]'s stream,
 *  so results equal g2s_fill_batch(sessions[0], ...) bit for bit.  Sessions
 *  must share one graph; they may sit on different devices (graph replicated,
 *  no exchange step) or on the same device (one session's host analysis then
 *  overlaps the other's kernels).  `timing` (optional) receives the sums over
 *  groups with ms_total = wall time of the call.
 *  g2s_session_set_team makes g2s_fill_batch / g2s_execute_* on `lead` use
 *  lead + helpers this way for lists longer than group_size; helpers must
 *  outlive the lead or be detached with nhelpers = 0.
 * ------------------------------------------------------------------------ */
int g2s_team_fill(g2s_session* const* sessions, int nsessions, const g2s_gap* gaps, size_t n, size_t group_size,
                  g2s_result* results, char* fill_arena, size_t arena_cap, g2s_timing* timing);
size_t g2s_team_arena_bytes(const g2s_session* s, const g2s_gap* gaps, size_t n);
int g2s_session_set_team(g2s_session* lead, g2s_session* const* helpers, int nhelpers, size_t group_size);
/* group_size for g2s_session_set_team: every list of at least 512 gaps per session is cut into ONE group per session
 * (a whole share of the list per GPU: each GPU fills, traces and writes its own share, the rand() stream chained from
 * share to share by draw totals — what a caller that streams long lists over several GPUs wants; Gap2Seq-core -devices) */
#define G2S_GROUP_PER_SESSION ((size_t)-1)

/* ---------------------------------------------------------------------------
 *  One list over several PROCESSES, one GPU each (ABI 6) — for a launcher that pins a device per rank (torchrun with
 *  HIP_VISIBLE_DEVICES per rank), where g2s_team_fill's one process cannot see the other GPUs.  The reference's threads
 *  share one rand() stream (Gap2Seq.cpp:178,296-306, at -nb-cores 1: in input order); here every rank fills,
 *  classifies, traces and writes ITS contiguous share of the list, and the shares are placed in the one stream by two
 *  exchanges of host scalars between the ranks (any transport: gloo all-gathers in gap2seq_amd/shard.py — nothing
 *  crosses between the devices, no RCCL):
 *
 *    g2s_share_begin   the share's kernels up to its draw totals: totals[0] = draws if every draw-dependent gap took
 *                      its fewest, totals[1] = the summed spreads.          --> all ranks learn all totals
 *    g2s_share_tables  base0 / R0 = the sums of totals[0] / totals[1] over the ranks in front: the tables, and the
 *                      share's function fn[d] = deviation behind the share for deviation d in front of it, d = 0 .. R0
 *                      (*fn: valid until g2s_share_end).                    --> all ranks learn all functions
 *    g2s_share_trace   d_in = the functions of the ranks in front composed from 0: tracebacks, results, fill text.
 *    g2s_share_end     list_draws = what the whole list drew (the sum of all totals[0] + the last function's value):
 *                      this rank's generator moves past the list, as every other rank's does.
 *
 *  Every rank holds the same session state in front of the list (same seed, same lists before).  results / fill_arena:
 *  the share's own, g2s_host_alloc memory.  A share of fewer than 256 gaps, a gap with a skip rule at a share's head, a
 *  list resident mode does not take: G2S_ERR_STATE from g2s_share_begin on that rank (the caller falls back to one rank
 *  filling the list).  Results are those of g2s_fill_batch on the whole list, bit for bit (tests/test_gpu_resident.py).
 * ------------------------------------------------------------------------ */
int g2s_share_begin(g2s_session* s, const g2s_gap* gaps, size_t n, g2s_result* results, char* fill_arena, size_t arena_cap,
                    uint64_t totals[2]);
int g2s_share_tables(g2s_session* s, uint64_t base0, uint64_t R0, const uint32_t** fn);
int g2s_share_trace(g2s_session* s, uint32_t d_in);
int g2s_share_end(g2s_session* s, uint64_t list_draws);

/* ---------------------------------------------------------------------------
 *  Gap2Seq::execute() after the graph exists (Gap2Seq.cpp:224-438): scaffold
 *  scanner, per-gap statistics text, splice, FASTA text.  Outputs are malloc'ed
 *  strings released with g2s_free.  `reads_label`/`filled_label` only feed the
 *  parameter echo (Gap2Seq.cpp:180-191).
 * ------------------------------------------------------------------------ */
typedef struct g2s_run_opts {
  int32_t k, solid, max_fuz, nb_cores;
  double max_mem_gb;
} g2s_run_opts;
int g2s_execute_scaffolds(g2s_session* s, const g2s_run_opts* o, const char* reads_label,
                          const char* filled_label, const char* scaffolds_text, char** fasta, char** log,
                          int32_t* gaps, int32_t* filled);
/* The same run with the text handed over as the records are done (the reference prints a gap's statistics when
 * the gap is done, Gap2Seq.cpp:385, and appends a record to the output bank when the record is, :426-431):
 * the gaps are filled in batches of about chunk_gaps (cut at record boundaries; 0 = one batch) and after every
 * batch on_log / on_fasta receive, in input order, what that batch decided.  The concatenated text does not
 * depend on chunk_gaps. */
typedef void (*g2s_text_fn)(const char* text, size_t len, void* user);
int g2s_execute_scaffolds_stream(g2s_session* s, const g2s_run_opts* o, const char* reads_label,
                                 const char* filled_label, const char* scaffolds_text, size_t chunk_gaps,
                                 g2s_text_fn on_fasta, g2s_text_fn on_log, void* user, int32_t* gaps, int32_t* filled);
int g2s_execute_single(g2s_session* s, const g2s_run_opts* o, const char* reads_label, const char* filled_label,
                       const char* left, const char* right, int32_t length, char** fasta, char** log);
void g2s_free(void* p);

/* ---------------------------------------------------------------------------
 *  The formats either side of Gap2Seq-core in the reference's pipeline
 *  (Gap2Seq.py:294-326: GapCutter -> Gap2Seq-core -> GapMerger), host string work.
 *  g2s_cut_scaffolds: GapCutter.cpp:119-321 — scaffolds FASTA/Q text -> contigs FASTA,
 *  one-gap records FASTA (comment "<name> scaffold S contig C gap G[ split 1| split 2 k]",
 *  flanks of at most k+fuz bases), BED lines, and the tool's stdout text.
 *  g2s_merge_scaffolds: GapMerger.cpp:142-235 — contigs + (filled) gap records -> scaffolds
 *  FASTA with the markers stripped, and the tool's stdout text.
 *  The *_label arguments only feed the echoed file names.  Outputs are malloc'ed
 *  strings released with g2s_free.
 * ------------------------------------------------------------------------ */
int g2s_cut_scaffolds(const char* scaffolds_text, int k, int fuz, int mask, int no_split, const char* scaffolds_label,
                      const char* contigs_label, const char* gaps_label, const char* bed_label, char** contigs_out,
                      char** gaps_out, char** bed_out, char** log_out);
int g2s_merge_scaffolds(const char* contigs_text, const char* gaps_text, const char* scaffolds_label,
                        const char* contigs_label, const char* gaps_label, char** scaffolds_out, char** log_out);

/* ---------------------------------------------------------------------------
 *  The reference's ReadFilter (ReadFilter.cpp:344-415; called once per gap and library by Gap2Seq.py:145-149,
 *  once per library with unmapped_only by Gap2Seq.py:64-72): from a BAM file of aligned read pairs, the reads
 *  that can belong to one gap, as FASTA text (">name/1" or ">name/2", the read as sequenced).  Host work beside
 *  the fill path: BGZF/BAM are read with zlib, no index file is needed (the reference wants "<bam>.bai").
 *  fasta_out is "" when nothing was extracted (the reference then leaves no output file, Gap2Seq.py:151-153);
 *  log_out is the tool's stdout line (:406), warn_out what it writes to stderr (:188-190).  The strings are
 *  malloc'ed, released with g2s_free.  g2s_filter_last_error() describes a G2S_ERR_IO.
 * ------------------------------------------------------------------------ */
typedef struct g2s_filter_opts {
  int32_t mean_insert, std_dev;      /* -mean, -std-dev */
  int32_t breakpoint;                /* -breakpoint: 0-based position of the gap in the scaffold */
  int32_t gap_length;                /* -gap-length (default -1, as the reference's parser has it) */
  int32_t flank_length;              /* -flank-length; -1 = do not add the reads overlapping the flanks */
  int32_t unmapped_only;             /* -unmapped-only */
  int32_t threads;                   /* inflating threads; 0 = up to 8 */
  const char* scaffold;              /* -scaffold: reference sequence name in the BAM header */
} g2s_filter_opts;
int g2s_filter_reads(const char* bam_path, const g2s_filter_opts* o, char** fasta_out, char** log_out, char** warn_out,
                     int64_t* extracted, int64_t* total);
int g2s_filter_reads_mem(const void* bam_bytes, size_t n, const g2s_filter_opts* o, char** fasta_out, char** log_out,
                         char** warn_out, int64_t* extracted, int64_t* total);
const char* g2s_filter_last_error(void);

/* ---------------------------------------------------------------------------
 *  Page-locked host memory the GPUs can write: a `results` array or fill arena
 *  allocated here is written by the kernels directly (no staging copy) when a
 *  list is finished on the device.  Any other memory works too.  The reference's
 *  caller allocates `fill` with new[] (Gap2Seq.cpp:374).
 * ------------------------------------------------------------------------ */
void* g2s_host_alloc(size_t bytes);
void g2s_host_free(void* p);

/* Accessors. */
const g2s_graph* g2s_session_graph(const g2s_session* s);
int g2s_session_get_params(const g2s_session* s, g2s_params* out);

/* (The g2s_test_* entry points the unit tests call are declared in include/g2s_test.h: they are not part of the
 * reference-facing interface.) */

/* Checks the invariants the kernels rely on between the unitig-start bitmap and
 * the successor table (every edge the bitmap calls unitig-internal is the only edge out of
 * its source and the only edge into its target, in both orientations; the last-base table
 * agrees with the successor slots).  Returns the number of violations found (0 = consistent)
 * and describes the first few in msg (NUL-terminated, at most msg_cap bytes). */
int64_t g2s_graph_validate(const g2s_graph* g, char* msg, size_t msg_cap);

/* Number of usable gfx950 devices (0 when none / no driver). */
int g2s_device_count(void);

/* ---------------------------------------------------------------------------
 *  Synthetic workloads of BASELINE.md / SURVEY.md §8(d) (seeded, generator
 *  lives here so that bench.py, the tests and the CLI agree byte for byte).
 *  variant bit 0: plant repeats (V1), bit 1: second haplotype with one
 *  substitution every ~500 bp (V2).  Returns malloc'ed FASTA texts.
 * ------------------------------------------------------------------------ */
int g2s_synth_genome(uint64_t length, uint32_t variant, uint64_t seed, char** reads_fasta);
int g2s_synth_gaps(const char* reads_fasta, int k, int fuz, int ngaps, int min_len, int max_len, uint64_t seed,
                   char** scaffolds_fasta);

#ifdef __cplusplus
}
#endif
#endif /* G2S_H_ */
