#!/usr/bin/env python3
"""bench.py — gaps filled/sec of the MI355X fill path on BASELINE.json's workload.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (g2s_batch_run: the fused kernel g2s_fill_lds =
phases A-D1 of every gap, writing its results into pinned host memory [+ the HBM-tier
kernels for gaps that outgrow the LDS], host phase D2 overlapping the kernel, the
in-order rand() offset pass and the tracebacks) over one batch of synthetic gaps whose
descriptors are already resident in HBM (g2s_batch_prepare is outside the timed region,
as is the one-off graph build + upload, reported separately).

Workload at N=1: BASELINE config 2 — synthetic 3 Mbp genome (seed 20240101),
k=31, -fuz 10, -dist-error 500, 500 gaps of 200-1000 bp (seed 20240103), one gap
per record with (k+fuz)-base flanks.  Default graph variant V3 = planted repeats
(V1) + second haplotype with a SNP every ~500 bp (V2); --variant 0/1/2 select the
others.  N>1: one process per GPU (torchrun), the graph replicated per GPU, every
rank fills its own 500-gap set (seed + rank) -> weak scaling, no data-path
collective; torch.distributed is used only for the barrier and the max/sum
reduction of the timing.

Rank 0 prints ONE JSON line (see README/DESIGN.md §Measurement for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_gaps(scaffolds_text, fuz):
    lines = scaffolds_text.splitlines()
    out = []
    for j in range(0, len(lines), 2):
        s = lines[j + 1]
        a = s.index("N")
        b = len(s) - s[::-1].index("N")
        out.append(dict(left=s[:a], right=s[b:], gap_len=b - a, lmf=fuz, rmf=fuz))
    return out


def algorithmic_bytes(x, s, io_bytes):
    """24 B per expansion (4 B frontier id + 16 B successor record +
    4 B next-frontier write) + 8 B per newly set state + per-gap flank/fill I/O."""
    return 24 * x + 8 * s + io_bytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (a step takes ~1.3 ms: 200 of them ride out host noise)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--variant", type=int, default=3, help="graph variant: bit0 repeats (V1), bit1 bubbles (V2)")
    ap.add_argument("--genome", type=int, default=3000000)
    ap.add_argument("--gaps", type=int, default=500, help="gaps per GPU")
    ap.add_argument("--min-len", type=int, default=200)
    ap.add_argument("--max-len", type=int, default=1000)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--fuz", type=int, default=10)
    ap.add_argument("--dist-error", type=int, default=500)
    ap.add_argument("--sessions", type=int, default=1,
                    help="sessions per GPU; >1 (or --group) times g2s_team_fill: groups of gaps pipelined over the "
                         "sessions, host flank lookup + upload INSIDE the timed region")
    ap.add_argument("--group", type=int, default=0, help="gaps per group for --sessions (0 = library default)")
    ap.add_argument("--host-threads", type=int, default=0, help="host worker threads per session (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the barrier/timing reduction")
    ap.add_argument("--share-device", action="store_true",
                    help="testing only: every rank uses HIP device 0 (needs --backend gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.share_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    from gap2seq_amd import lib as P
    from gap2seq_amd import shard

    if P.G2S.device_count() < 1:
        raise SystemExit("bench.py: no gfx950 device; the fill path has no CPU fallback")

    # ---- workload (untimed) -------------------------------------------------------
    t0 = time.time()
    reads = P.G2S.synth_genome(args.genome, args.variant, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    scaf = P.G2S.synth_gaps(reads, args.k, args.fuz, args.gaps, args.min_len, args.max_len, 20240103 + rank)
    gaps = parse_gaps(scaf, args.fuz)
    t_synth = time.time() - t0
    t0 = time.time()
    os.environ["G2S_DEVICE"] = str(local_rank)  # the graph is built on (and stays on) this rank's GPU
    graph = P.Graph.from_seqs(seqs, args.k, 1)
    t_build = time.time() - t0
    t0 = time.time()
    graph.upload(local_rank)
    t_upload = time.time() - t0
    sess = P.Session(graph, local_rank, d_err=args.dist_error, randseed=1, host_threads=args.host_threads)
    batch = sess.prepare([P.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gaps])

    def sync_all():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    team = None
    if args.sessions > 1 or args.group > 0:
        team = [sess] + [P.Session(graph, local_rank, d_err=args.dist_error, randseed=1)
                         for _ in range(args.sessions - 1)]
        team_gaps = [P.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gaps]
        team_prep, _ = P.team_fill(team, team_gaps, args.group, "raw")

    def one_step():
        sess.srand(1)
        if team is None:
            batch.run()  # synchronous: returns after kernels, copies and host phase D
            return batch.timing()
        return P.team_fill(team, team_gaps, args.group, "raw", team_prep)[1]

    for _ in range(args.warmup):
        one_step()
    acc = dict(ms_right_bfs=0.0, ms_left_dp=0.0, ms_extract=0.0, ms_fill_lds=0.0, ms_extract_lds=0.0, ms_d2h=0.0,
               ms_host_post=0.0, ms_total=0.0, launches=0)
    sync_all()
    t_begin = time.perf_counter()
    for _ in range(args.steps):
        tm = one_step()
        acc["ms_right_bfs"] += tm.ms_right_bfs
        acc["ms_left_dp"] += tm.ms_left_dp
        acc["ms_extract"] += tm.ms_extract
        acc["ms_fill_lds"] += tm.ms_fill_lds
        acc["ms_extract_lds"] += tm.ms_extract_lds
        acc["ms_d2h"] += tm.ms_d2h
        acc["ms_host_post"] += tm.ms_host_post
        acc["ms_total"] += tm.ms_total
        acc["launches"] += tm.launches_left_dp
    sync_all()
    elapsed = time.perf_counter() - t_begin
    elapsed, units = shard.reduce_timing(elapsed, float(len(gaps) * args.steps), dist)

    # host-buffers-in / host-buffers-out rate (prepare + run), reported beside `value`, never as `value`
    t_pcie = time.perf_counter()
    sess.srand(1)
    sess.fill_batch([P.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gaps])
    t_pcie = time.perf_counter() - t_pcie
    tm_team = one_step() if team is not None else None
    sess.srand(1)
    batch.run()
    tm = batch.timing()
    res = batch.results()
    if tm_team is not None:
        tm_team.fill_bytes = tm.fill_bytes
        tm_team.flank_bytes = tm.flank_bytes
        tm = tm_team
    filled = sum(1 for r in res if r.count > 0)
    q7 = sum(1 for r in res if r.flags & P.G2S_GAP_Q7)

    if rank == 0:
        steps = max(1, args.steps)
        # ---- CPU baseline: the oracle (faithful port of the reference algorithm), same gaps
        cpu = None
        octr = None
        if world == 1 and not args.no_cpu_baseline:
            import oracle_lib as O
            og = O.OracleGraph(seqs, args.k, 1)
            passes = 3
            secs1 = 0.0
            for _ in range(passes):
                s1, ofilled, octr = O.time_fill_batch(og, gaps, args.dist_error, 1)
                secs1 += s1
            ncpu = os.cpu_count() or 1
            sN, _, _ = O.time_fill_batch(og, gaps, args.dist_error, ncpu)
            cpu = dict(value=round(len(gaps) * passes / secs1, 2), unit="gaps/s", cores=1, kind="port",
                       sample="all %d gaps of the bench workload, %d passes, oracle fill_gap only (graph build excluded)"
                              % (len(gaps), passes),
                       value_all_cores=round(len(gaps) / sN, 2), cores_all=ncpu, filled=ofilled,
                       oracle_expansions_A_B_D1=[octr[0], octr[2], octr[4]],
                       oracle_states_A_B_D1=[octr[1], octr[3], octr[5]])
            og.free()
        # ---- roofline of the dominant kernel, measured live with HIP events on the session stream.
        # g2s_fill_lds runs phases A (right search), B (left DP), C (target check) and D1 (closure)
        # of every gap that fits the LDS tier, one wave per gap.  Algorithmic bytes
        # = 24 B per expansion + 8 B per newly set state over phases A, B and D1 + per-gap
        # flank/fill I/O, with the expansion/state counts of the REFERENCE algorithm as counted by
        # the CPU oracle on the same gaps (the product's own counters are lower for phase A: it
        # visits every node once, the reference re-expands nodes reached by walks of several
        # lengths); without the CPU leg the product's counters are used and labelled so.
        io_bytes = tm.flank_bytes + tm.fill_bytes
        if tm.lds_tier_gaps > 0:
            kname = "g2s_fill_lds"
            if octr is not None and tm.lds_tier_gaps == len(gaps):
                x_units, s_units, counted_by = octr[0] + octr[2] + octr[4], octr[1] + octr[3] + octr[5], "oracle"
            else:
                x_units, s_units, counted_by = tm.xA + tm.xB + tm.xD, tm.sA + tm.sB + tm.sD, "product"
            launches = float(max(1, tm.lds_launches))
            kern_ms = acc["ms_fill_lds"] / steps / launches  # average launch duration
        else:
            kname, x_units, s_units, counted_by = "g2s_left_dp", tm.xB, tm.sB, "product"
            kern_ms = acc["ms_left_dp"] / steps
            launches = acc["launches"] / steps
        alg_bytes = algorithmic_bytes(x_units, s_units, io_bytes) / launches  # per launch
        achieved = alg_bytes / (kern_ms / 1e3) / 1e9 if kern_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_v10_pmc_fill_lds.json")
        if os.path.exists(pmc) and kname == "g2s_fill_lds" and args.gaps == 500 and args.variant == 3:
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = dict(bound="hbm", kernel=kname, achieved=round(achieved, 3), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 6), traffic=traffic,
                        algorithmic_bytes_per_launch=alg_bytes, expansions=x_units, states=s_units,
                        units_counted_by=counted_by, kernel_ms_per_launch=round(kern_ms, 4),
                        launches_per_step=launches, lds_tier_gaps=tm.lds_tier_gaps)
        out = {
            "metric": "gaps filled/sec (whole node), k=31 synthetic 3 Mbp DBG",
            "value": round(units / elapsed, 2),
            "unit": "gaps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 2 (C2): %d bp genome V%d, k=%d, %d gaps/GPU len %d-%d, fuz %d, "
                                   "dist-error %d" % (args.genome, args.variant, args.k, args.gaps, args.min_len,
                                                      args.max_len, args.fuz, args.dist_error),
                       "gaps_per_gpu": args.gaps, "genome_bp": args.genome, "variant": args.variant, "k": args.k,
                       "parallelism": "gap-sharded x%d, graph replicated, no collective" % world,
                       "sessions_per_gpu": args.sessions, "group": args.group},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "filled": filled,
            "fill_batch_ms_incl_prepare_and_python_marshalling": round(t_pcie * 1e3, 3),
            "q7_gaps": q7,
            "retried_gaps": tm.retried_gaps,
            "breakdown_ms_per_step": {"fill_lds_kernel": round(acc["ms_fill_lds"] / steps, 4),
                                      "extract_lds_kernel": round(acc["ms_extract_lds"] / steps, 4),
                                      "hbm_tier_kernels": round((acc["ms_right_bfs"] + acc["ms_left_dp"] +
                                                                 acc["ms_extract"]) / steps, 4),
                                      "d2h_closures": round(acc["ms_d2h"] / steps, 4),
                                      "host_phase_d": round(acc["ms_host_post"] / steps, 4),
                                      "batch_run_total": round(acc["ms_total"] / steps, 4)},
            "setup_s": {"synth": round(t_synth, 3), "graph_build": round(t_build, 3),
                        "graph_upload": round(t_upload, 3)},
            "graph": {"kmers": graph.num_kmers, "unitigs": graph.num_unitigs,
                      "hbm_bytes": graph.device_bytes(local_rank)},
        }
        print(json.dumps(out))
    batch.free()
    for t in (team or [])[1:]:
        t.destroy()
    sess.destroy()
    graph.free()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
