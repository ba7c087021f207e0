#!/usr/bin/env python3
"""GPU box: one bench list N times through one session (srand(1) before every call), every call's results — every field
and the fill text of every gap — against the first call's: what a race between the waves of a kernel shows as when it
strikes once in hundreds of runs (the missing barrier in g2s_fill_segw's tail did, round 5).  The first call is the one the
parity suite checks against the oracle at full size; this only asks whether every later call says the same.

  python tools/self_consistency.py C2|C3|C4|C5 [N]        (G2S_DEVICE_D2, G2S_FORCE_SEGX, G2S_TRACE_WAVES ... as usual)
"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gap2seq_amd import lib as P  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
n_runs = int(sys.argv[2]) if len(sys.argv) > 2 else 200
genome_bp, k, ngaps, min_len, max_len, d_err, _ = bench.CONFIGS[cfg]
reads = P.G2S.synth_genome(genome_bp, 3, 20240101)
seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
gaps = bench.parse_gaps(P.G2S.synth_gaps(reads, k, 10, ngaps, min_len, max_len, 20240103), 10)
graph = P.Graph.from_seqs(seqs, k, 1)
sess = P.Session(graph, 0, d_err=d_err, randseed=1)
run = bench.Runner(P, [sess], gaps, 0)


def digest():
    return [bench.result_key(r) for r in run.results()]


run.step()
first = digest()
h0 = hashlib.sha256(repr(first).encode()).hexdigest()[:16]
bad = 0
for it in range(1, n_runs):
    run.step()
    now = digest()
    if now != first:
        bad += 1
        where = [i for i, (a, b) in enumerate(zip(first, now)) if a != b]
        print("run %d differs from the first at %d gaps: %s" % (it, len(where), where[:8]), flush=True)
        for i in where[:2]:
            print("   gap %d: first %r\n            now   %r" % (i, str(first[i])[:300], str(now[i])[:300]), flush=True)
tm = run.timing()
print("%s: %d gaps, %d runs, %d differ from the first (digest %s); filled %d; the last call: resident %d, fallbacks %d, host-finished %d" % (
    cfg, len(gaps), n_runs, bad, h0, sum(1 for r in run.results() if r.count > 0), tm.resident_launches, tm.resident_fallbacks, tm.host_finished_gaps))
sys.exit(1 if bad else 0)
