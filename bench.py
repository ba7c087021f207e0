#!/usr/bin/env python3
"""bench.py — gaps filled/sec of the MI355X fill path on BASELINE.json's workloads.

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE call of the boundary's hot-path entry point on one list of synthetic gaps
handed over as host C structs (g2s_gap[]): g2s_fill_batch at N=1, g2s_team_fill at N>1.
The timed region is that call and nothing else, so it contains what SURVEY.md 8(d)
defines as the metric: flank k-mer -> node resolution and the upload of the gap
descriptors (g2s_batch_prepare), the kernels (g2s_fill_lds = phases A-D1 of every gap,
results written into pinned host memory; the HBM-tier kernels for what outgrows the LDS),
the host part of phase D (D2 under the kernel, in-order rand() offsets, tracebacks) and
the results in the caller's buffers.  The one-off graph build + upload is outside and
reported separately.  Python builds the g2s_gap array once, before the timed region.

Workloads (synthetic, seeded): --config C2 (default at N=1: 3 Mbp genome
V3 = planted repeats + second haplotype, k=31, -fuz 10, -dist-error 500, 500 gaps of
200-1000 bp), C3 (default at N>1: the same graph, ONE list of 10 000 gaps whatever N is:
strong scaling), C4 (60 Mbp, k=63, 2 000 gaps), C5 (-dist-error 2000, 1 000 gaps of 2-5
kbp); individual parameters can be overridden.

N>1: one process drives N GPUs — the reference's dispatcher (Gap2Seq.cpp:296-306: threads
pulling scaffolds from a shared iterator) with one session (host thread + stream) per GPU:
the graph is replicated in every GPU's HBM, the gap list is cut into contiguous groups,
every session pulls the next group from a shared counter (static start, stealing by
construction), there is no collective; the rand()-dependent part runs once in gap order,
so the results equal the one-GPU results bit for bit (checked in the run).  Under
torch.distributed.run the other ranks only join the barriers around the timed region
(their GPUs are driven by rank 0's sessions); exit status is non-zero when fewer than N
devices are usable.

Rank 0 prints ONE JSON line (README / DESIGN.md §5 describe the fields).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CONFIGS = {
    #      genome    k  gaps  min   max  d_err  BASELINE.json configs[] text
    "C2": (3000000, 31, 500, 200, 1000, 500, "configs[1]: synthetic 3 Mbp genome, 500 gaps len 200-1000 bp, k=31, 1xMI355X"),
    "C3": (3000000, 31, 10000, 200, 1000, 500, "configs[2]: same 3 Mbp DBG, 10 000 gaps, static shard across the GPUs of one node"),
    "C4": (60000000, 63, 2000, 200, 1000, 500, "configs[3]: k=63 (128-bit k-mers), ~60 Mbp DBG, 2 000 gaps, 1xMI355X"),
    "C5": (3000000, 31, 1000, 2000, 5000, 2000, "configs[4]: --dist-error 2000 deep-DP stress, 1 000 gaps len 2-5 kbp"),
}


def parse_gaps(scaffolds_text, fuz):
    lines = scaffolds_text.splitlines()
    out = []
    for j in range(0, len(lines), 2):
        s = lines[j + 1]
        a = s.index("N")
        b = len(s) - s[::-1].index("N")
        out.append(dict(left=s[:a], right=s[b:], gap_len=b - a, lmf=fuz, rmf=fuz))
    return out


UNITS_FILE = os.path.join(ROOT, "profiles", "oracle_units.json")
GENOME_SEED, GAP_SEED = 20240101, 20240103


def units_key(genome_bp, variant, k, ngaps, min_len, max_len, fuz, d_err):
    """Key of a synthetic workload in profiles/oracle_units.json (written by tools/oracle_units.py)."""
    return "genome%d.V%d.seed%d|k%d|gaps%d.len%d-%d.seed%d|fuz%d|e%d" % (
        genome_bp, variant, GENOME_SEED, k, ngaps, min_len, max_len, GAP_SEED, fuz, d_err)


def load_oracle_units():
    try:
        with open(UNITS_FILE) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def oracle_units(key, count_now=None):
    """(X, S, source) of SURVEY.md 8(d) for one workload, ALWAYS the CPU oracle's counts of the reference
    algorithm's expansions and newly set states over phases A, B and D1 (Gap2Seq.cpp:912,930,1031,1049-1065,
    1266-1301): from the committed table (tools/oracle_units.py ran the oracle over the whole list), else counted
    in this run by count_now() (the cpu_baseline leg, all host cores), else (None, None, why) — and then no
    roofline fraction is printed.  The kernels' own counters are never used for this: the segment tier does not
    perform those expansions one by one, its phase A counter is an upper bound (1.64x on config 3)."""
    u = load_oracle_units().get(key)
    if u:
        return (u["xA"] + u["xB"] + u["xD"], u["sA"] + u["sB"] + u["sD"],
                "oracle (profiles/oracle_units.json: %s threads over all %d gaps)" % (u.get("threads", "?"), u["gaps"]))
    if count_now is not None:
        c = count_now()
        if c is not None:
            return c[0] + c[2] + c[4], c[1] + c[3] + c[5], "oracle (counted in this run over the whole list)"
    return None, None, "unavailable: workload not in profiles/oracle_units.json and the oracle did not cover the whole list"


def algorithmic_bytes(x, s, io_bytes):
    """24 B per expansion (4 B frontier id + 16 B successor record +
    4 B next-frontier write) + 8 B per newly set state + per-gap flank/fill I/O."""
    return 24 * x + 8 * s + io_bytes


class Runner:
    """The boundary call of one step, with every buffer allocated once."""

    def __init__(self, P, sessions, gaps, group, pinned=True):
        self.P, self.lib = P, P.load_library()
        self.sessions = sessions
        self.n = len(gaps)
        self.group = group
        self.arr, self._keep = P._gap_array([P.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gaps])
        self.nbytes = self.lib.g2s_team_arena_bytes(sessions[0].h, self.arr, self.n)
        # the caller's buffers, allocated once: page-locked through the ABI (g2s_host_alloc) so that a list finished on
        # the device is written there by the kernels themselves; --pageable-buffers takes ordinary memory (staged copy)
        if pinned:
            self._bufs = [P.HostBuffer(max(1, self.nbytes)), P.HostBuffer(C.sizeof(P.g2s_result) * max(1, self.n))]
            self.arena = C.cast(self._bufs[0].p, C.c_char_p)
            self.res = self._bufs[1].array(P.g2s_result, max(1, self.n))
        else:
            self._bufs = []
            self.arena = C.create_string_buffer(max(1, self.nbytes))
            self.res = (P.g2s_result * max(1, self.n))()
        self.hs = (C.c_void_p * len(sessions))(*[s.h for s in sessions])
        self.tm = P.g2s_timing()

    def step(self):
        """srand(1) + ONE ABI call; returns its wall time in seconds (the timed region of a step)."""
        self.sessions[0].srand(1)
        t0 = time.perf_counter()
        if len(self.sessions) == 1 and self.group == 0:
            rc = self.lib.g2s_fill_batch(self.sessions[0].h, self.arr, self.n, self.res, self.arena, self.nbytes)
        else:
            rc = self.lib.g2s_team_fill(self.hs, len(self.sessions), self.arr, self.n, self.group, self.res, self.arena,
                                        self.nbytes, C.byref(self.tm))
        dt = time.perf_counter() - t0
        self.P._check(rc)
        return dt

    def stream(self, k, keep=False):
        """srand(1) once, then K consecutive calls on the same list: the rand() stream goes on from call to call (the
        reference's one stream over the whole run, Gap2Seq.cpp:178).  Returns (seconds, per-call result keys when keep)."""
        self.sessions[0].srand(1)
        out = []
        t0 = time.perf_counter()
        for _ in range(k):
            if len(self.sessions) == 1 and self.group == 0:
                rc = self.lib.g2s_fill_batch(self.sessions[0].h, self.arr, self.n, self.res, self.arena, self.nbytes)
            else:
                rc = self.lib.g2s_team_fill(self.hs, len(self.sessions), self.arr, self.n, self.group, self.res, self.arena,
                                            self.nbytes, C.byref(self.tm))
            self.P._check(rc)
            if keep:
                out.append([result_key(r) for r in self.results()])
        return time.perf_counter() - t0, out

    def timing(self):
        if len(self.sessions) == 1 and self.group == 0:
            self.P._check(self.lib.g2s_session_last_timing(self.sessions[0].h, C.byref(self.tm)))
        return self.tm

    def results(self):
        raw = self._bufs[0].raw if self._bufs else self.arena.raw
        return [self.P.FillResult(self.res[i], raw) for i in range(self.n)]

    def free(self):
        for hb in self._bufs:
            hb.free()
        self._bufs = []


class StreamRunner:
    """K consecutive lists on one session, `depth` in flight (g2s_fill_begin / g2s_fill_end): list i+depth-1 is begun —
    its kernels queued — before list i is ended, so they run while list i's phase D3 writes its results through the
    link and the host prepares the next list.  The product's steady state (Gap2Seq-core -stream-gaps) is such a
    sequence of lists."""

    def __init__(self, P, session, gaps, nlists, pinned=True, depth=3):
        self.P, self.lib, self.s, self.k = P, P.load_library(), session, nlists
        self.depth = max(2, min(depth, P.G2S_MAX_IN_FLIGHT))
        self.n = len(gaps)
        self.arr, self._keep = P._gap_array([P.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gaps])
        self.nbytes = self.lib.g2s_team_arena_bytes(session.h, self.arr, self.n)
        self.sets = []
        for _ in range(self.depth):  # (a list's buffers are free again once it has ended)
            if pinned:
                a, r = P.HostBuffer(max(1, self.nbytes)), P.HostBuffer(C.sizeof(P.g2s_result) * max(1, self.n))
                self.sets.append((a, r, C.cast(a.p, C.c_void_p), r.array(P.g2s_result, max(1, self.n))))
            else:
                a = C.create_string_buffer(max(1, self.nbytes))
                self.sets.append((a, None, C.cast(a, C.c_void_p), (P.g2s_result * max(1, self.n))()))

    def _keys(self, q):
        a, _, _, res = self.sets[q]
        raw = a.raw
        return [result_key(self.P.FillResult(res[i], raw)) for i in range(self.n)]

    def run(self, overlapped, keep=False):
        """srand(1), then the K lists; returns (seconds, per-list result keys when keep)."""
        self.s.srand(1)
        out = []
        t0 = time.perf_counter()
        if overlapped:
            D, ended = self.depth, 0
            for i in range(self.k):
                _, _, ap, res = self.sets[i % D]
                self.P._check(self.lib.g2s_fill_begin(self.s.h, self.arr, self.n, res, ap, self.nbytes))
                if i >= D - 1:
                    self.P._check(self.lib.g2s_fill_end(self.s.h))
                    if keep:
                        out.append(self._keys(ended % D))
                    ended += 1
            while ended < self.k:
                self.P._check(self.lib.g2s_fill_end(self.s.h))
                if keep:
                    out.append(self._keys(ended % D))
                ended += 1
        else:
            for i in range(self.k):
                _, _, ap, res = self.sets[i % self.depth]
                self.P._check(self.lib.g2s_fill_batch(self.s.h, self.arr, self.n, res, C.cast(ap, C.c_char_p), self.nbytes))
                if keep:
                    out.append(self._keys(i % self.depth))
        return time.perf_counter() - t0, out

    def free(self):
        # (after an error between begin and end the lists still in flight write into these buffers: end them first)
        while self.lib.g2s_fill_in_flight(self.s.h) > 0:
            self.lib.g2s_fill_end(self.s.h)
        for a, r, _, _ in self.sets:
            if r is not None:
                a.free()
                r.free()
        self.sets = []


def result_key(r):
    return (r.count, r.left_fuz, r.right_fuz, r.flags, r.draws, r.fill, tuple(r.substats), r.phaseC_count, tuple(r.lengths))


def run_rank_per_gpu(args, dist, rank, world, seen):
    """N ranks, one GPU each: rank r fills, traces and writes the r-th contiguous share of the list on the device it
    sees; the shares are placed in the one rand() stream by two all-gathers of host scalars (shard.fill_share).  The
    timed region, the barriers and the reduction are the contract's; rank 0 prints the line."""
    from gap2seq_amd import lib as P
    from gap2seq_amd import shard
    cfg_name = args.config or "C3"
    genome_bp, k, ngaps, min_len, max_len, d_err, cfg_text = CONFIGS[cfg_name]
    genome_bp = args.genome or genome_bp
    k = args.k or k
    ngaps = args.gaps or ngaps
    if args.weak:
        ngaps *= world
    min_len, max_len, d_err = args.min_len or min_len, args.max_len or max_len, args.dist_error or d_err
    steps = args.steps or {"C2": 200, "C3": 30, "C4": 30, "C5": 3}[cfg_name]
    warmup = args.warmup if args.warmup >= 0 else (10 if steps >= 100 else 3 if steps >= 10 else 1)
    os.environ.setdefault("G2S_KERNEL_TIMING", "all")
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    device = 0 if (seen[rank] == 1 or args.share_device) else local % seen[rank]
    reads = P.G2S.synth_genome(genome_bp, args.variant, GENOME_SEED)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = parse_gaps(P.G2S.synth_gaps(reads, k, args.fuz, ngaps, min_len, max_len, GAP_SEED), args.fuz)
    os.environ["G2S_DEVICE"] = str(device)
    t0 = time.time()
    graph = P.Graph.from_seqs(seqs, k, 1)
    graph.upload(device)
    t_graph = time.time() - t0
    sess = P.Session(graph, device, d_err=d_err, randseed=1, host_threads=args.host_threads)
    comm = shard.DistComm(dist)
    lo, hi = shard.share_bounds(len(gaps), world)[rank]
    lib = P.load_library()
    arr, _keep = P._gap_array([P.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gaps[lo:hi]])
    n = hi - lo
    nbytes = lib.g2s_team_arena_bytes(sess.h, arr, n)
    abuf, rbuf = P.HostBuffer(max(1, nbytes)), P.HostBuffer(C.sizeof(P.g2s_result) * max(1, n))
    res = rbuf.array(P.g2s_result, max(1, n))
    ap_ = C.cast(abuf.p, C.c_void_p)

    def step():
        sess.srand(1)
        t = time.perf_counter()
        draws = shard.fill_share(P, sess, comm, arr, n, res, ap_, nbytes)
        return time.perf_counter() - t, draws

    # ---- the shares must give the one-GPU result, bit for bit: every rank fills the WHOLE list once on its own GPU
    # (g2s_fill_batch, untimed) and compares its share with that
    whole = Runner(P, [sess], gaps, 0, True)
    whole.step()
    want = [result_key(r) for r in whole.results()][lo:hi]
    whole.free()
    _, draws = step()
    got = None if draws is None else [result_key(P.FillResult(res[i], abuf.raw)) for i in range(n)]
    same = comm.all_gather([1 if got == want else 0, 1 if draws is not None else 0])
    if not all(x[1] for x in same):
        if rank == 0:
            sys.stderr.write("bench.py: one rank per GPU: a share could not be taken on its device (shares of %s gaps)\n" % [b - a for a, b in shard.share_bounds(len(gaps), world)])
        dist.destroy_process_group()
        return 3
    if not all(x[0] for x in same):
        if rank == 0:
            sys.stderr.write("bench.py: one rank per GPU: the shares differ from the one-GPU results on rank(s) %s\n" % [r for r, x in enumerate(same) if not x[0]])
        dist.destroy_process_group()
        return 4
    for _ in range(warmup):
        step()
    kern_ms, timed = 0.0, 0
    dist.barrier()
    t_begin = time.perf_counter()
    for _ in range(steps):
        step()
        tm = sess.last_timing()
        kern_ms += tm.ms_fill_seg
        timed += tm.seg_timed_launches
    elapsed = time.perf_counter() - t_begin
    dist.barrier()
    elapsed, units = shard.reduce_timing(elapsed, float(n * steps), dist)
    tm = sess.last_timing()
    per_rank = comm.all_gather([int(kern_ms * 1e6), timed, n, int(tm.fill_bytes), int(tm.flank_bytes), tm.seg_tier_gaps + tm.segx_tier_gaps,
                                sum(1 for i in range(n) if res[i].count > 0)])
    if rank == 0:
        x_units, s_units, counted_by = oracle_units(units_key(genome_bp, args.variant, k, len(gaps), min_len, max_len, args.fuz, d_err))
        kms = [p[0] / 1e6 / max(1, p[1]) for p in per_rank]  # average launch duration per rank
        kern = sum(kms) / len(kms)
        io = sum(p[3] + p[4] for p in per_rank)
        alg = algorithmic_bytes(x_units, s_units, io) / world if x_units is not None else None  # per launch: a share each
        ach = alg / (kern / 1e3) / 1e9 if alg is not None and kern > 0 else None
        out = {
            "metric": "gaps filled/sec (whole node), k=%d synthetic %s DBG" % (k, "3 Mbp" if genome_bp == 3000000 else "%d bp" % genome_bp),
            "value": round(units / elapsed, 2), "unit": "gaps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak" if args.weak else "strong",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "BASELINE %s: %d bp genome V%d, k=%d, %d gaps len %d-%d, fuz %d, dist-error %d" % (
                           cfg_text, genome_bp, args.variant, k, len(gaps), min_len, max_len, args.fuz, d_err),
                       "config": cfg_name, "gaps": len(gaps), "genome_bp": genome_bp, "variant": args.variant, "k": k,
                       "timed_region": "per rank: srand(1) + g2s_share_begin / _tables / _trace / _end on its share, with the two all-gathers "
                                       "of host scalars between them (gap2seq_amd/shard.py: fill_share)",
                       "parallelism": "%d processes, one GPU and one contiguous share of the list each (%s gaps), graph replicated, the shares "
                                      "placed in the one rand() stream by two %s all-gathers of draw totals and share functions; no "
                                      "collective on the data path, results stay with the ranks" % (world, [p[2] for p in per_rank], dist.get_backend()),
                       "mode": "rank_per_gpu", "devices_seen_by_rank": seen, "ranks_under_torchrun": world},
            "roofline": dict(bound="hbm", kernel="g2s_fill_seg", achieved=round(ach, 3) if ach is not None else None, peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=round(ach / HBM_PEAK_GBS, 6) if ach is not None else None, traffic=None,
                             algorithmic_bytes_per_launch=alg, expansions=x_units, states=s_units, units_counted_by=counted_by,
                             kernel_ms_per_launch=round(kern, 4), kernel_ms_per_launch_by_rank=[round(x, 4) for x in kms], launches_per_step=float(world)),
            "cpu_baseline": None,
            "filled": sum(p[6] for p in per_rank),
            "equals_one_gpu_result": True,
            "list_draws": draws,
            "setup_s": {"graph_build_and_upload_per_rank": round(t_graph, 3)},
        }
        print(json.dumps(out))
        sys.stdout.flush()
    abuf.free(); rbuf.free()
    sess.destroy()
    graph.free()
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (0 = 200 for C2, fewer for the larger configs)")
    ap.add_argument("--warmup", type=int, default=-1)
    ap.add_argument("--config", default="", help="C2 | C3 | C4 | C5 (default: C2 at --gpus 1, C3 beyond)")
    ap.add_argument("--variant", type=int, default=3, help="graph variant: bit0 repeats (V1), bit1 bubbles (V2)")
    ap.add_argument("--genome", type=int, default=0)
    ap.add_argument("--gaps", type=int, default=0, help="gaps in the list (whole job)")
    ap.add_argument("--min-len", type=int, default=0)
    ap.add_argument("--max-len", type=int, default=0)
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--fuz", type=int, default=10)
    ap.add_argument("--dist-error", type=int, default=0)
    ap.add_argument("--sessions", type=int, default=1, help="sessions (host thread + stream) per GPU")
    ap.add_argument("--group", type=int, default=-1,
                    help="gaps per group of the dispatcher (-1 = one group per session at N>1, one batch at N=1)")
    ap.add_argument("--groups-per-session", type=int, default=1,
                    help="N>1: groups of the list per session (pulled from a shared counter: whoever is free takes the next)")
    ap.add_argument("--host-threads", type=int, default=0, help="host worker threads per session (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c3-beside", action="store_true", help="N=1/C2: skip the C3-on-one-GPU measurement beside it")
    ap.add_argument("--prime-seconds", type=float, default=0.3,
                    help="untimed steps run for this long before the W warm-up steps (a fresh box starts with idle CPU "
                         "clocks, sleeping worker threads and unallocated pinned buffers: the first ~0.1 s of steps run "
                         "up to 40 %% slower); reported as priming_steps")
    ap.add_argument("--pageable-buffers", action="store_true",
                    help="results and fill arena in ordinary memory instead of g2s_host_alloc's page-locked memory")
    ap.add_argument("--backend", default="gloo", help="torch.distributed backend for the barriers under torchrun")
    ap.add_argument("--share-device", action="store_true", help="testing only: all N sessions on HIP device 0")
    ap.add_argument("--weak", action="store_true",
                    help="N>1: weak scaling — the workload's gap count PER GPU (config 3: 10 000 each) instead of one list "
                         "for all of them")
    ap.add_argument("--in-flight", type=int, default=3, help="--stream-lists: lists begun and not ended (2 or 3)")
    ap.add_argument("--stream-lists", type=int, default=0,
                    help="N=1: also measure K consecutive lists of the workload with --in-flight of them in flight (g2s_fill_begin / "
                         "g2s_fill_end) against the same K lists one at a time; reported as `stream_lists`")
    ap.add_argument("--rank-per-gpu", action="store_true",
                    help="N>1 under torchrun: every rank drives ONE GPU and fills its share of the list (g2s_share_*, "
                         "gap2seq_amd/shard.py: fill_share); chosen by itself when rank 0 cannot see N devices (a launcher "
                         "that pins a device per rank)")
    ap.add_argument("--dry-run", action="store_true",
                    help="testing only (CPU): the launch protocol — rendezvous, barriers, timing reduction, one JSON "
                         "line from rank 0 — without touching a device or measuring anything")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:  # launched by torch.distributed.run: barriers only, no data-path collective exists
        import torch.distributed as dist
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    ngpu = max(1, args.gpus)

    def barrier():
        if dist is not None:
            dist.barrier()

    from gap2seq_amd import shard
    # ---- which form a multi-rank launch takes.  One process drives all N GPUs (rank 0; the other ranks only join the
    # barriers) whenever rank 0 can see N devices.  A launcher that pins a device per rank leaves it one: every rank
    # then drives the GPU it sees and fills its share of the list (run_rank_per_gpu).  The ranks agree on the form by
    # one all-gather of what each of them sees (counting devices does not touch them).
    if world > 1 and not args.dry_run:
        from gap2seq_amd import lib as P0
        seen = shard.DistComm(dist).all_gather([P0.G2S.device_count()])
        per_rank = args.rank_per_gpu or (seen[0][0] < ngpu and not args.share_device)
        if per_rank:
            if world != ngpu or min(x[0] for x in seen) < 1:
                if rank == 0:
                    sys.stderr.write("bench.py: --gpus %d with %d rank(s) seeing %s gfx950 device(s): neither one process for all "
                                     "GPUs (rank 0 sees %d) nor one rank per GPU is possible\n" % (ngpu, world, [x[0] for x in seen], seen[0][0]))
                dist.destroy_process_group()
                return 2
            return run_rank_per_gpu(args, dist, rank, world, [x[0] for x in seen])
    if rank != 0:
        # this rank's GPU is driven by a session of rank 0's process (one process, N devices)
        barrier()  # timed region begins
        t_begin = time.perf_counter()
        barrier()  # timed region ends
        shard.reduce_timing(time.perf_counter() - t_begin, 0.0, dist)
        dist.destroy_process_group()
        return 0
    if args.dry_run:
        barrier()
        t_begin = time.perf_counter()
        time.sleep(0.01)
        elapsed = time.perf_counter() - t_begin
        barrier()
        elapsed, units = shard.reduce_timing(elapsed, 1.0, dist)
        print(json.dumps({"dry_run": True, "n_gpus": ngpu, "ranks_under_torchrun": world, "value": None,
                          "elapsed_max_over_ranks_s": round(elapsed, 4), "units": units,
                          "group": shard.group_size(10000, ngpu), "groups": len(shard.group_bounds(10000, shard.group_size(10000, ngpu)))}))
        if dist is not None:
            dist.destroy_process_group()
        return 0

    from gap2seq_amd import lib as P

    ndev = P.G2S.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no gfx950 device; the fill path has no CPU fallback")
    if ndev < ngpu and not args.share_device:
        # ONE line that names what usually hides the devices from this rank (a launcher that pins a device per rank)
        vis = {v: os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if os.environ.get(v) is not None}
        sys.stderr.write("bench.py: rank %d sees %d gfx950 device(s) but --gpus %d needs %d in THIS process (one process drives all GPUs; the "
                         "other ranks only join the barriers): %s\n"
                         % (rank, ndev, ngpu, ngpu, ("unset " + ", ".join("%s=%s" % kv for kv in vis.items()) + " for rank 0") if vis else
                            "HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES are not set here - the node has fewer devices, or the container hides them"))
        raise SystemExit("bench.py: --gpus %d but only %d gfx950 device(s) are usable" % (ngpu, ndev))
    devices = [0] * ngpu if args.share_device else list(range(ngpu))

    cfg_name = args.config or ("C2" if ngpu == 1 else "C3")
    genome_bp, k, ngaps, min_len, max_len, d_err, cfg_text = CONFIGS[cfg_name]
    genome_bp = args.genome or genome_bp
    k = args.k or k
    ngaps = args.gaps or ngaps
    if args.weak and ngpu > 1:
        ngaps *= ngpu
    min_len = args.min_len or min_len
    max_len = args.max_len or max_len
    d_err = args.dist_error or d_err
    custom = any([args.genome, args.k, args.gaps, args.min_len, args.max_len, args.dist_error, args.variant != 3,
                  args.fuz != 10])
    steps = args.steps or {"C2": 200, "C3": 30, "C4": 30, "C5": 3}[cfg_name]
    warmup = args.warmup if args.warmup >= 0 else (10 if steps >= 100 else 3 if steps >= 10 else 1)
    # Resident mode brackets one fill-kernel launch in eight with HIP events (the events cost the stream ~10 us per
    # list): enough for an average over 100 steps and more; a shorter run brackets more of its launches.
    if ngpu > 1 and "G2S_KERNEL_TIMING" not in os.environ:
        # (a team's per-session times — resident.team_ms_by_session — are the LAST step's: every launch bracketed, so that
        # the line of a multi-GPU run explains itself; ~10 us per list and session)
        os.environ["G2S_KERNEL_TIMING"] = "all"
    if steps < 100 and "G2S_KERNEL_TIMING" not in os.environ:
        # (at least three bracketed launches inside the timed steps, however few those are)
        os.environ["G2S_KERNEL_TIMING"] = "sample:%d" % max(1, min(8, (steps + 2) // 3))

    # ---- workload (untimed) -------------------------------------------------------
    t0 = time.time()
    reads = P.G2S.synth_genome(genome_bp, args.variant, GENOME_SEED)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = parse_gaps(P.G2S.synth_gaps(reads, k, args.fuz, ngaps, min_len, max_len, GAP_SEED), args.fuz)
    t_synth = time.time() - t0
    t0 = time.time()
    os.environ["G2S_DEVICE"] = str(devices[0])  # the graph is built on (and stays on) the first GPU
    graph = P.Graph.from_seqs(seqs, k, 1)
    t_build = time.time() - t0
    t0 = time.time()
    for d in sorted(set(devices)):
        graph.upload(d)  # replicas for the other GPUs
    t_upload = time.time() - t0

    def make_sessions(devs, per_dev):
        return [P.Session(graph, d, d_err=d_err, randseed=1, host_threads=args.host_threads)
                for d in devs for _ in range(per_dev)]

    sessions = make_sessions(devices, max(1, args.sessions))
    group = args.group
    if group < 0:
        # groups are cut for DEVICES: sessions that share a device (--share-device, --sessions) pull from the same
        # counter, and a device gains nothing from two launches where one would do
        # (--share-device: one group per SESSION, as N GPUs would get them — the sharded phase D3 is what is being tested)
        group = 0 if len(sessions) == 1 else shard.group_size(len(gaps), len(sessions) if args.share_device else len(set(devices)),
                                                                per_session=args.groups_per_session)
    run = Runner(P, sessions, gaps, group, not args.pageable_buffers)

    # what a one-shot run pays: the first call on a fresh session (buffers are allocated and page-locked in it,
    # sleeping workers and idle clocks wake up), and the second (everything allocated, nothing warm yet)
    first_call_ms = run.step() * 1e3
    second_call_ms = run.step() * 1e3
    priming_steps = 0
    t_prime = time.perf_counter()
    while time.perf_counter() - t_prime < args.prime_seconds:
        run.step()
        priming_steps += 1
    for _ in range(warmup):
        run.step()
    acc = dict(ms_right_bfs=0.0, ms_left_dp=0.0, ms_extract=0.0, ms_fill_lds=0.0, ms_extract_lds=0.0, ms_d2h=0.0,
               ms_host_post=0.0, ms_prepare=0.0, ms_total=0.0, ms_fill_seg=0.0, ms_fill_segx=0.0, ms_d3=0.0, launches=0,
               lds_launches=0, seg_launches=0, segx_launches=0, seg_timed=0, d3_timed_steps=0)
    in_call = 0.0
    host_us = [0.0] * 6
    # (inside the timed steps only the call and a copy of its timing record — 0.3 us; the sums over the records are
    # taken behind the loop: two dozen attribute reads per step were 4-5 us of Python between two 170 us calls)
    snaps = (P.g2s_timing * steps)()
    tm_bytes = C.sizeof(P.g2s_timing)
    step_s = [0.0] * steps
    barrier()
    t_begin = time.perf_counter()
    for i in range(steps):
        step_s[i] = run.step()
        C.memmove(C.byref(snaps[i]), C.byref(run.timing()), tm_bytes)
    elapsed = time.perf_counter() - t_begin
    barrier()
    in_call = sum(step_s)
    for tm in snaps:
        for key in ("ms_right_bfs", "ms_left_dp", "ms_extract", "ms_fill_lds", "ms_extract_lds", "ms_d2h",
                    "ms_host_post", "ms_prepare", "ms_total", "ms_fill_seg", "ms_fill_segx", "ms_d3"):
            acc[key] += getattr(tm, key)
        acc["launches"] += tm.launches_left_dp
        acc["lds_launches"] += tm.lds_launches
        acc["seg_launches"] += tm.seg_launches
        acc["seg_timed"] += tm.seg_timed_launches  # resident mode brackets one launch in eight with HIP events
        acc["d3_timed_steps"] += 1 if tm.ms_d3 > 0 else 0
        acc["segx_launches"] += tm.segx_launches
        for q in range(6):
            host_us[q] += tm.host_us[q]
    elapsed, units = shard.reduce_timing(elapsed, float(len(gaps) * steps), dist)

    tm = run.timing()
    res = run.results()
    filled = sum(1 for r in res if r.count > 0)
    q7 = sum(1 for r in res if r.flags & P.G2S_GAP_Q7)

    # ---- N>1: the list sharded over N GPUs must give the one-GPU result, bit for bit ----------
    one_gpu = None
    if len(sessions) > 1:
        solo = make_sessions(devices[:1], 1)
        r1 = Runner(P, solo, gaps, 0, not args.pageable_buffers)
        r1.step()
        same = [result_key(a) for a in r1.results()] == [result_key(b) for b in res]
        if not same:
            raise SystemExit("bench.py: results on %d sessions differ from the one-session results" % len(sessions))
        n1 = max(3, min(steps, 10))
        t1 = 0.0
        for _ in range(n1):
            t1 += r1.step()
        one_gpu = dict(value=round(len(gaps) * n1 / t1, 2), unit="gaps/s", steps=n1, ms_per_step=round(t1 / n1 * 1e3, 4),
                       note="the same list through g2s_fill_batch on device %d alone, same run" % devices[0])
        for s in solo:
            s.destroy()

    # ---- N=1 on C2: config 3's list on this one GPU, beside the headline ----------------------
    c3_beside = None
    if ngpu == 1 and cfg_name == "C2" and not custom and not args.no_c3_beside:
        g3 = parse_gaps(P.G2S.synth_gaps(reads, k, args.fuz, 10000, min_len, max_len, GAP_SEED), args.fuz)
        r3 = Runner(P, sessions[:1], g3, 0, not args.pageable_buffers)
        for _ in range(5):  # (the first three calls on a list of another size grow the session's buffers)
            r3.step()
        n3, t3, k3, timed3 = 10, 0.0, 0.0, 0
        keep_mode = os.environ.get("G2S_KERNEL_TIMING")
        os.environ["G2S_KERNEL_TIMING"] = "sample:4"  # (ten steps: three launches bracketed with HIP events, ~10 us each)
        step_ms3, hus3 = [], []
        for _ in range(n3):
            dt3 = r3.step()
            step_ms3.append(round(dt3 * 1e3, 3))
            t3 += dt3
            t = r3.timing()
            hus3.append([round(t.host_us[q], 1) for q in range(6)])
            k3 += t.ms_fill_seg if t.seg_tier_gaps else t.ms_fill_lds
            timed3 += t.seg_timed_launches if t.seg_tier_gaps else t.lds_launches
        if keep_mode is None:
            del os.environ["G2S_KERNEL_TIMING"]
        else:
            os.environ["G2S_KERNEL_TIMING"] = keep_mode
        tm3 = r3.timing()
        nl3 = max(1, tm3.seg_launches if tm3.seg_tier_gaps else tm3.lds_launches)
        x3, s3, by3 = oracle_units(units_key(genome_bp, args.variant, k, 10000, min_len, max_len, args.fuz, d_err))
        kms3 = k3 / max(1, timed3)  # average duration of the launches that were timed
        ab3 = algorithmic_bytes(x3, s3, tm3.flank_bytes + tm3.fill_bytes) / nl3 if x3 is not None else None
        c3_beside = dict(workload=CONFIGS["C3"][6], value=round(10000 * n3 / t3, 2), unit="gaps/s", steps=n3,
                         ms_per_step=round(t3 / n3 * 1e3, 4), step_ms=step_ms3,
                         value_of_the_median_step=round(10000 / (sorted(step_ms3)[len(step_ms3) // 2] * 1e-3), 2),
                         slowest_step_host_us_inside_the_call=dict(zip(("entry_to_fill_kernel_queued", "to_phase_d3_queued", "to_hand_over_seen", "to_host_finished_gaps_done", "to_stream_synchronised", "to_return"), hus3[step_ms3.index(max(step_ms3))])),
                         kernel=("g2s_fill_seg2" if tm3.seg2_launches else "g2s_fill_seg") if tm3.seg_tier_gaps else "g2s_fill_lds",
                         kernel_ms_per_launch=round(kms3, 4), kernel_launches_timed="%d of %d" % (timed3, n3 * nl3), gaps_left_to_other_kernels=10000 - max(tm3.seg_tier_gaps, tm3.lds_tier_gaps),
                         launches_per_step=nl3, algorithmic_bytes_per_launch=ab3, expansions=x3, states=s3,
                         units_counted_by=by3,
                         roofline_frac=round(ab3 / (kms3 / 1e3) / 1e9 / HBM_PEAK_GBS, 6) if ab3 is not None else None,
                         filled=sum(1 for r in r3.results() if r.count > 0))

    # ---- N=1: K consecutive lists, --in-flight of them in flight, against the same lists one at a time ---------------
    stream_lists = None
    if ngpu == 1 and len(sessions) == 1 and args.stream_lists >= 2:
        sr = StreamRunner(P, sessions[0], gaps, args.stream_lists, not args.pageable_buffers, args.in_flight)
        _, want = sr.run(False, keep=True)
        _, got = sr.run(True, keep=True)
        if got != want:
            raise SystemExit("bench.py: %d lists in flight differ from the lists one at a time" % args.stream_lists)
        reps = max(3, min(steps, 10))
        for _ in range(2):
            sr.run(True)
        t_ov = sum(sr.run(True)[0] for _ in range(reps))
        for _ in range(2):
            sr.run(False)
        t_seq = sum(sr.run(False)[0] for _ in range(reps))
        tot = float(len(gaps) * args.stream_lists * reps)
        stream_lists = dict(lists=args.stream_lists, in_flight=sr.depth, gaps_per_list=len(gaps), repetitions=reps,
                            value=round(tot / t_ov, 2), unit="gaps/s", ms_per_list=round(t_ov / (args.stream_lists * reps) * 1e3, 4),
                            one_list_at_a_time=round(tot / t_seq, 2), ms_per_list_one_at_a_time=round(t_seq / (args.stream_lists * reps) * 1e3, 4),
                            results="identical to the lists one at a time, list by list (checked in this run)",
                            how="g2s_fill_begin(list i+D-1) before g2s_fill_end(list i), D = in_flight: the younger lists' kernels run while the oldest one's results cross the link; the rand() stream continues from list to list on the device")
        sr.free()

    # ---- N>1: a STREAM of lists over the team — every call hands each GPU a whole share of the list (one group per
    # session: each GPU fills, traces and writes its own share, the rand() stream chained from share to share and from
    # list to list), K calls in a row without reseeding; against the same K lists on one session, list by list
    if len(sessions) > 1 and args.stream_lists >= 2:
        solo = make_sessions(devices[:1], 1)
        r1 = Runner(P, solo, gaps, 0, not args.pageable_buffers)
        _, want = r1.stream(args.stream_lists, keep=True)
        _, got = run.stream(args.stream_lists, keep=True)
        if got != want:
            raise SystemExit("bench.py: a stream of %d lists on %d sessions differs from the same lists on one session" % (args.stream_lists, len(sessions)))
        reps = max(3, min(steps, 10))
        run.stream(args.stream_lists)
        t_team = sum(run.stream(args.stream_lists)[0] for _ in range(reps))
        r1.stream(args.stream_lists)
        t_one = sum(r1.stream(args.stream_lists)[0] for _ in range(reps))
        tot = float(len(gaps) * args.stream_lists * reps)
        stream_lists = dict(lists=args.stream_lists, sessions=len(sessions), gaps_per_list=len(gaps), gaps_per_session_and_list=group or len(gaps),
                            repetitions=reps, value=round(tot / t_team, 2), unit="gaps/s",
                            ms_per_list=round(t_team / (args.stream_lists * reps) * 1e3, 4),
                            one_session=round(tot / t_one, 2), ms_per_list_one_session=round(t_one / (args.stream_lists * reps) * 1e3, 4),
                            results="identical to the same lists on one session, list by list (checked in this run)",
                            how="g2s_team_fill per list, one group per session (phase D3 sharded); the rand() stream runs on from list to list; one list at a time (lists in flight are a one-session feature so far)")
        r1.free()
        for s_ in solo:
            s_.destroy()

    # ---- N>1, the default line: the driver passes no flags, and the strong headline (config 3's ONE list over N GPUs)
    # is bound by its 1 250-gap shares at N = 8 — beside it, in the same line: WEAK scaling (the config's gap count per
    # GPU in one list, each GPU a share of 10 000) and a STREAM of such lists (the rand() stream running on from list
    # to list), each checked against one session and timed against it
    weak_beside = None
    if len(sessions) > 1 and not custom and not args.weak and args.stream_lists < 2 and group != 0:
        gw = parse_gaps(P.G2S.synth_gaps(reads, k, args.fuz, ngaps * ngpu, min_len, max_len, GAP_SEED), args.fuz)
        gweak = shard.group_size(len(gw), len(sessions) if args.share_device else len(set(devices)), per_session=args.groups_per_session)
        rw = Runner(P, sessions, gw, gweak, not args.pageable_buffers)
        solo = make_sessions(devices[:1], 1)
        r1 = Runner(P, solo, gw, 0, not args.pageable_buffers)
        rw.step()
        r1.step()
        if [result_key(a) for a in r1.results()] != [result_key(b) for b in rw.results()]:
            raise SystemExit("bench.py: the weak list (%d gaps) on %d sessions differs from the one-session results" % (len(gw), len(sessions)))
        nw = max(3, min(steps, 6))
        rw.step()
        tw = sum(rw.step() for _ in range(nw))
        t1w = sum(r1.step() for _ in range(nw))
        kl = 4
        _, want = r1.stream(kl, keep=True)
        _, got = rw.stream(kl, keep=True)
        if got != want:
            raise SystemExit("bench.py: a stream of %d weak lists on %d sessions differs from the same lists on one session" % (kl, len(sessions)))
        rw.stream(kl)
        ts = sum(rw.stream(kl)[0] for _ in range(3))
        t1s = sum(r1.stream(kl)[0] for _ in range(3))
        weak_beside = dict(
            weak=dict(scaling="weak", gaps=len(gw), gaps_per_session=gweak, steps=nw, value=round(len(gw) * nw / tw, 2), unit="gaps/s",
                      ms_per_step=round(tw / nw * 1e3, 4), one_session=round(len(gw) * nw / t1w, 2),
                      ms_per_step_one_session=round(t1w / nw * 1e3, 4), results="identical to one session (checked in this run)"),
            stream=dict(lists=kl, gaps_per_list=len(gw), gaps_per_session_and_list=gweak, repetitions=3,
                        value=round(len(gw) * kl * 3 / ts, 2), unit="gaps/s", ms_per_list=round(ts / (kl * 3) * 1e3, 4),
                        one_session=round(len(gw) * kl * 3 / t1s, 2), ms_per_list_one_session=round(t1s / (kl * 3) * 1e3, 4),
                        results="identical to the same lists on one session, list by list (checked in this run)"),
            how="g2s_team_fill, one share of the list per session: every GPU fills, traces and writes its own share, the shares "
                "placed in the one rand() stream by draw totals and share functions between the sessions' threads; no collective")
        rw.free(); r1.free()
        for s_ in solo:
            s_.destroy()

    # ---- CPU baseline: the oracle (port of the reference algorithm) on the GPU box's host cores,
    # N=1 only, on a bounded sample of the same gaps
    cpu = None
    octr = None
    if ngpu == 1 and not args.no_cpu_baseline:
        import oracle_lib as O
        og = O.OracleGraph(seqs, k, 1)
        sample = gaps
        # ~1 k gaps/s on one core for C2-shaped gaps; deep gaps (C5) take ~25 ms each
        budget = 1500 if d_err <= 500 else 300
        if len(sample) > budget:
            sample = gaps[:budget]
        passes = 3 if len(sample) <= 500 else 1
        secs1 = 0.0
        for _ in range(passes):
            s1, ofilled, octr = O.time_fill_batch(og, sample, d_err, 1)
            secs1 += s1
        ncpu = os.cpu_count() or 1
        # What the box lets this process RUN at once: its affinity mask, cut by the container's CPU quota (an average over
        # 100 ms — 16 CPUs' worth on the pool's boxes of 256 logical CPUs).  That many threads are started: 256 threads
        # time-sliced on 16 CPUs gave 4.3 k, 13.5 k and 14.1 k gaps/s for one build on three boxes.
        quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota = round(int(q) / int(per), 2)
        except (OSError, ValueError):
            pass
        try:
            affinity = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            affinity = ncpu
        threads_used = max(1, min(affinity, int(quota) if quota and quota >= 1 else affinity))
        passes_all = 3 if len(sample) <= 1500 else 1
        O.time_fill_batch(og, sample, d_err, threads_used)  # (untimed: the threads' first touch of the graph)
        sN_each = [O.time_fill_batch(og, sample, d_err, threads_used)[0] for _ in range(passes_all)]
        sN = sorted(sN_each)[len(sN_each) // 2]  # the median pass
        cpu = dict(value=round(len(sample) * passes / secs1, 2), unit="gaps/s", cores=1, kind="port",
                   sample="%s %d gaps of the bench list, %d pass(es), oracle fill_gap only (graph build excluded)"
                          % ("all" if len(sample) == len(gaps) else "the first", len(sample), passes),
                   value_all_cores=round(len(sample) / sN, 2), threads_used=threads_used, passes_all_cores=passes_all,
                   value_all_cores_each_pass=[round(len(sample) / x, 1) for x in sN_each],
                   cores_all=ncpu, cpu_affinity=affinity, cpu_quota_cgroup=quota, filled=ofilled,
                   oracle_expansions_A_B_D1=[octr[0], octr[2], octr[4]],
                   oracle_states_A_B_D1=[octr[1], octr[3], octr[5]])
        if len(sample) != len(gaps):
            octr = None  # the oracle's unit counts cover the sample only
            key = units_key(genome_bp, args.variant, k, len(gaps), min_len, max_len, args.fuz, d_err)
            if key not in load_oracle_units() and len(gaps) / (len(sample) / sN) < 90.0:
                # a list the committed table does not hold: the units of the WHOLE list, all cores, untimed
                _, _, octr = O.time_fill_batch(og, gaps, d_err, threads_used)
                cpu["oracle_units_counted_over"] = "the whole list of %d gaps, %d threads" % (len(gaps), threads_used)
        og.free()

    # ---- roofline of the dominant kernel, its duration measured with HIP events on the session
    # streams inside the timed steps (g2s_timing.ms_fill_lds, summed over launches).  g2s_fill_lds
    # runs phases A (right search), B (left DP), C (target check) and D1 (closure) of every gap that
    # fits the LDS tier.  Algorithmic bytes = 24 B per expansion + 8 B per newly set
    # state over phases A, B and D1 + per-gap flank/fill I/O, with the unit counts of the REFERENCE
    # algorithm as counted by the CPU ORACLE over the whole list (oracle_units above: the committed
    # table profiles/oracle_units.json, or counted by the cpu_baseline leg of this run).  Without an
    # oracle count no fraction is printed; the kernels' own counters never price a roofline.
    io_bytes = tm.flank_bytes + tm.fill_bytes
    seg_gaps = tm.seg_tier_gaps + tm.segx_tier_gaps
    x_units, s_units, counted_by = oracle_units(
        units_key(genome_bp, args.variant, k, len(gaps), min_len, max_len, args.fuz, d_err),
        (lambda: octr) if octr is not None else None)
    if seg_gaps > 0 and seg_gaps >= tm.lds_tier_gaps:
        # the segment tier took (most of) the list: phases A-D2 over unitig segments, one wave per gap
        # (short lists run two waves per gap: the same code as g2s_fill_seg, phase A on the second wave)
        kname = "g2s_fill_seg2" if tm.seg2_launches else "g2s_fill_seg"
        launches = acc["seg_launches"] / float(steps)
        # average duration of the launches bracketed with HIP events inside the timed steps: every launch on the host
        # path, one in eight in resident mode (an event between two kernels costs the stream 4-5 us;
        # G2S_KERNEL_TIMING=all brackets every launch)
        kern_ms = acc["ms_fill_seg"] / max(1, acc["seg_timed"])
        if tm.segx_tier_gaps > 0:
            # deep gaps took the tier's large variant as well: the two kernels' launches of a step as one unit
            kname = kname + " + g2s_fill_segw"
            kern_ms = (acc["ms_fill_seg"] + acc["ms_fill_segx"]) / max(1, acc["seg_timed"])
    elif tm.lds_tier_gaps > 0:
        kname = "g2s_fill_lds"
        launches = acc["lds_launches"] / float(steps)
        kern_ms = acc["ms_fill_lds"] / max(1, acc["lds_launches"])  # average launch duration
    else:
        kname = "g2s_left_dp"
        launches = max(1.0, acc["launches"] / float(steps))
        kern_ms = acc["ms_left_dp"] / max(1, acc["launches"])
    if x_units is not None:
        alg_bytes = algorithmic_bytes(x_units, s_units, io_bytes) / max(1.0, launches)  # per launch
        achieved = alg_bytes / (kern_ms / 1e3) / 1e9 if kern_ms > 0 else 0.0
    else:
        alg_bytes, achieved = None, None  # no oracle count for this list: no fraction is quoted
    traffic, traffic_src = None, None
    # HBM bytes per launch of that kernel from the committed counter passes of this workload (rocprofv3 --pmc
    # FETCH_SIZE / WRITE_SIZE in passes of their own, gfx950 read-side correction: tools/pmc_summary.py)
    pmc = None
    for rnd in ("r06", "r05", "r04", "r03"):  # (the latest round that holds counter passes of this workload)
        pmc = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (rnd, cfg_name.lower()))
        if os.path.exists(pmc):
            break
    if os.path.exists(pmc) and not custom and ngpu == 1:
        try:
            pj = json.load(open(pmc))
            if pj.get("kernel", "") == kname.split(" ")[0]:
                traffic = pj.get("hbm_bytes_per_launch")
                traffic_src = "from_profile: profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of " \
                              "this command; not measured in this run)" % os.path.basename(pmc)
        except Exception:
            traffic = None
    roofline = dict(bound="hbm", kernel=kname, achieved=round(achieved, 3) if achieved is not None else None,
                    peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 6) if achieved is not None else None,
                    traffic=traffic, traffic_source=traffic_src,
                    achieved_is="algorithmic bytes of the REFERENCE algorithm (SURVEY 8d: 24 B per expansion + 8 B per state, "
                                "oracle counts) per second of the kernel — not bytes moved: the segment search touches a fraction of them",
                    measured_traffic_GBps=round(traffic / (kern_ms / 1e3) / 1e9, 3) if traffic and kern_ms > 0 else None,
                    algorithmic_bytes_per_launch=alg_bytes, expansions=x_units, states=s_units,
                    units_counted_by=counted_by, kernel_ms_per_launch=round(kern_ms, 4),
                    launches_per_step=round(launches, 3),
                    launches_timed_with_hip_events=(acc["seg_timed"] if kname.startswith("g2s_fill_seg") else None),
                    launches_in_the_timed_steps=(acc["seg_launches"] if kname.startswith("g2s_fill_seg") else None),
                    seg_tier_gaps=tm.seg_tier_gaps, segx_tier_gaps=tm.segx_tier_gaps, lds_tier_gaps=tm.lds_tier_gaps,
                    segments=tm.seg_segments)
    workload = "BASELINE %s%s: %d bp genome V%d, k=%d, %d gaps len %d-%d, fuz %d, dist-error %d" % (
        cfg_text if not custom else "custom (based on %s)" % cfg_name, "", genome_bp, args.variant, k, len(gaps), min_len,
        max_len, args.fuz, d_err)
    per_step = lambda key: round(acc[key] / steps, 4)  # noqa: E731
    out = {
        "metric": "gaps filled/sec (whole node), k=%d synthetic %s DBG" % (k, "3 Mbp" if genome_bp == 3000000 else "%d bp" % genome_bp),
        "value": round(units / elapsed, 2),
        "unit": "gaps/s",
        "n_gpus": len(set(devices)) if not args.share_device else ngpu,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak" if (ngpu == 1 or args.weak) else "strong",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": workload, "config": cfg_name if not custom else cfg_name + "-custom",
                   "gaps": len(gaps), "genome_bp": genome_bp, "variant": args.variant, "k": k,
                   "timed_region": "one %s call per step: flank lookup + descriptor upload + kernels + host phase D"
                                   % ("g2s_fill_batch" if (len(sessions) == 1 and group == 0) else "g2s_team_fill"),
                   "parallelism": "one process, %d GPU(s) x %d session(s), graph replicated, list cut into groups of %s "
                                  "pulled from a shared counter, no collective"
                                  % (ngpu, max(1, args.sessions), group if group else "all"),
                   "devices": devices, "group": group, "ranks_under_torchrun": world},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "filled": filled,
        "q7_gaps": q7,
        "priming_steps": priming_steps,
        "first_call_ms": round(first_call_ms, 3),
        "second_call_ms": round(second_call_ms, 3),
        "retried_gaps": tm.retried_gaps,
        "resident": {"lists_finished_on_the_device": tm.resident_launches, "lists_given_back_to_the_host_path": tm.resident_fallbacks,
                     "draw_dependent_gaps": tm.draw_dependent_gaps, "draw_count_table_entries": tm.d3_table_entries,
                     "gaps_traced_by_the_fill_kernel": tm.traced_in_fill_gaps,
                     "gaps_guessed_by_the_fill_kernel": tm.guessed_in_fill_gaps,
                     "guessed_text_groups_of_64_bases_compared_and_sent_again": [tm.guessed_groups, tm.guessed_groups_resent],
                     "gaps_finished_by_the_host": tm.host_finished_gaps,
                     "team_groups": tm.team_groups,
                     "team_groups_by_session": [tm.team_groups_by_session[i] for i in range(min(16, max(1, len(sessions))))],
                     "team_phase_d3": ("sharded: every session traces its own group and writes its own results" if tm.team_d3_sharded
                                       else "on the lead's device (groups gathered)") if len(sessions) > 1 else None,
                     "team_ms_by_session": {"fill_kernel": [round(tm.team_ms_fill[i], 4) for i in range(min(16, len(sessions)))],
                                            "phase_d3_kernels": [round(tm.team_ms_d3[i], 4) for i in range(min(16, len(sessions)))],
                                            "thread_wall": [round(tm.team_ms_wall[i], 4) for i in range(min(16, len(sessions)))],
                                            "of": "the last timed step"} if len(sessions) > 1 else None,
                     "buffers": "pageable" if args.pageable_buffers else "page-locked (g2s_host_alloc)"},
        "breakdown_ms_per_step": {"wall_inside_the_abi_call": round(in_call / steps * 1e3, 4),
                                  "wall_inside_the_abi_call_min_median_max": [round(x * 1e3, 4) for x in (min(step_s), sorted(step_s)[len(step_s) // 2], max(step_s))],
                                  "wall_inside_the_abi_call_each": [round(x * 1e3, 4) for x in step_s] if steps <= 40 else None,
                                  "prepare_flank_lookup_and_upload": per_step("ms_prepare"),
                                  "fill_seg_kernel": round(acc["ms_fill_seg"] / max(1, acc["seg_timed"]) * acc["seg_launches"] / steps, 4),
                                  "fill_segx_kernel": per_step("ms_fill_segx"),
                                  "phase_d3_kernels": round(acc["ms_d3"] / max(1, acc["d3_timed_steps"]), 4),
                                  "fill_lds_kernel": per_step("ms_fill_lds"),
                                  "extract_lds_kernel": per_step("ms_extract_lds"),
                                  "hbm_tier_kernels": round((acc["ms_right_bfs"] + acc["ms_left_dp"] +
                                                             acc["ms_extract"]) / steps, 4),
                                  "d2h_closures": per_step("ms_d2h"),
                                  "host_phase_d": per_step("ms_host_post"),
                                  "library_total": per_step("ms_total"),
                                  "host_us_inside_the_call": dict(zip(
                                      ("entry_to_fill_kernel_queued", "to_phase_d3_queued", "to_hand_over_seen", "to_host_finished_gaps_done",
                                       "to_stream_synchronised", "to_return"), [round(x / steps, 2) for x in host_us])),
                                  "note": "kernel and host figures are sums over sessions and overlap at N>1"},
        "setup_s": {"synth": round(t_synth, 3), "graph_build": round(t_build, 3), "graph_upload": round(t_upload, 3)},
        "graph": {"kmers": graph.num_kmers, "unitigs": graph.num_unitigs, "hbm_bytes": graph.device_bytes(devices[0])},
    }
    if one_gpu is not None:
        out["one_gpu_same_list"] = one_gpu
        out["equals_one_gpu_result"] = True
    if c3_beside is not None:
        out["c3_on_one_gpu"] = c3_beside
    if stream_lists is not None:
        out["stream_lists"] = stream_lists
    if weak_beside is not None:
        out["weak_and_stream_beside"] = weak_beside
    print(json.dumps(out))
    for s in sessions:
        s.destroy()
    graph.free()
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
