#!/usr/bin/env python3
"""GPU box, library built with -DG2S_SEG_PROFILE: which gaps of a chip-filling list are its slowest, and where the launch
order (longest gap first) puts them.  usage: G2S_LIBRARY=gap2seq_amd/_prof/libg2s_hip.so G2S_SEG_WAVES=1 python tools/r04_heavy.py [C3]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gap2seq_amd import lib as P  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
genome_bp, k, ngaps, min_len, max_len, d_err, _ = bench.CONFIGS[cfg]
reads = P.G2S.synth_genome(genome_bp, 3, 20240101)
seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
gaps = bench.parse_gaps(P.G2S.synth_gaps(reads, k, 10, ngaps, min_len, max_len, 20240103), 10)
dump = tempfile.mktemp()
os.environ["G2S_SEG_DUMP"] = dump
pg = P.Graph.from_seqs(seqs, k, 1)
sess = P.Session(pg, 0, d_err=d_err, randseed=1)
res, tm = sess.fill_batch([P.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gaps], True)
rows = {}
cur = None
for ln in open(dump):
    p = ln.split()
    if p[0] == "gap":
        cur = dict(gap=int(p[1]), nseg=int(p[5]), rounds=int(p[11]), rounds_a=int(p[9]), pa=[0, 0, 0, 0], prof=None)
        rows[cur["gap"]] = cur
    elif p[0] == "PA":
        cur["pa"] = [int(x) for x in p[1:5]]
    elif p[0] == "P":
        cur["prof"] = [int(x) for x in p[1:5]]
        cur["tail"] = [int(x) for x in p[5:9]]
rows = [r for r in rows.values() if r["prof"]]
for r in rows:
    r["len"] = gaps[r["gap"]]["gap_len"]
    r["A"] = sum(r["pa"]); r["B"] = sum(r["prof"]); r["T"] = sum(r["tail"]); r["all"] = r["A"] + r["B"] + r["T"]
order = sorted(rows, key=lambda r: -r["len"])
rank = {r["gap"]: i for i, r in enumerate(order)}
tot = sum(r["all"] for r in rows)
print("gaps %d, cycles A %d B %d tail %d, all %d (%.0f per gap)" % (len(rows), sum(r["A"] for r in rows), sum(r["B"] for r in rows), sum(r["T"] for r in rows), tot, tot / len(rows)))
print("the slowest gaps: gap, length, place in the launch order, cycles A + B + tail")
for r in sorted(rows, key=lambda r: -r["all"])[:25]:
    print("gap %5d len %4d place %5d: %7d = A %6d + B %6d + tail %6d (segments %d, rounds %d)" % (r["gap"], r["len"], rank[r["gap"]], r["all"], r["A"], r["B"], r["T"], r["nseg"], r["rounds"]))
# by decile of the launch order: mean and max cycles
n = len(order)
for d in range(10):
    part = order[d * n // 10:(d + 1) * n // 10]
    print("launch order %3d%%-%3d%%: lengths %d-%d, mean %6.0f, max %7d cycles" % (10 * d, 10 * d + 10, part[-1]["len"], part[0]["len"], sum(r["all"] for r in part) / len(part), max(r["all"] for r in part)))
