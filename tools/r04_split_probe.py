#!/usr/bin/env python3
"""Probe: one long list against the same list cut into K parts that are in flight together (g2s_fill_begin /
g2s_fill_end) — is a single list's step shorter when its parts overlap each other?"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from gap2seq_amd import lib as P

genome_bp, k, ngaps, min_len, max_len, d_err, _ = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
reads = P.G2S.synth_genome(genome_bp, 3, bench.GENOME_SEED)
seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
gaps = bench.parse_gaps(P.G2S.synth_gaps(reads, k, 10, ngaps, min_len, max_len, bench.GAP_SEED), 10)
graph = P.Graph.from_seqs(seqs, k, 1)
s = P.Session(graph, 0, d_err=d_err, randseed=1)
lib = P.load_library()
G = [P.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gaps]


def make(parts):
    out = []
    n = len(G)
    for q in range(parts):
        sub = G[q * n // parts:(q + 1) * n // parts]
        arr, keep = P._gap_array(sub)
        nb = lib.g2s_team_arena_bytes(s.h, arr, len(sub))
        a, r = P.HostBuffer(max(1, nb)), P.HostBuffer(C.sizeof(P.g2s_result) * len(sub))
        out.append((arr, keep, len(sub), nb, a, r, C.cast(a.p, C.c_void_p), r.array(P.g2s_result, len(sub))))
    return out


def run(parts, reps, depth=3):
    t0 = time.perf_counter()
    for _ in range(reps):
        ended = 0
        for i, (arr, keep, n, nb, a, r, ap, res) in enumerate(parts):
            P._check(lib.g2s_fill_begin(s.h, arr, n, res, ap, nb))
            if i >= depth - 1:
                P._check(lib.g2s_fill_end(s.h)); ended += 1
        while ended < len(parts):
            P._check(lib.g2s_fill_end(s.h)); ended += 1
    return (time.perf_counter() - t0) / reps


for K in (1, 2, 3, 4, 6):
    parts = make(K)
    run(parts, 20)
    t = min(run(parts, 30) for _ in range(3))
    print("%d part(s): %.4f ms per list, %.2f M gaps/s" % (K, t * 1e3, len(G) / t / 1e6), flush=True)
