#!/usr/bin/env python3
"""GPU box: the slowest gaps of a bench configuration's list in the regular tier's fill kernel — wave cycles from the
kernel's own clock reads (GapOut.stat: phase A | phases B + C | tail), through G2S_DUMP_STATS (host path of the same
kernels: the look-up kernel in front, results fetched by the host).  A short list's launch IS its slowest gap.
usage: python tools/slow_gaps.py [C2|C3] [top]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gap2seq_amd import lib as P  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
genome_bp, k, ngaps, min_len, max_len, d_err, _ = bench.CONFIGS[cfg]
reads = P.G2S.synth_genome(genome_bp, 3, 20240101)
seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
gaps = bench.parse_gaps(P.G2S.synth_gaps(reads, k, 10, ngaps, min_len, max_len, 20240103), 10)
dump = tempfile.mktemp()
os.environ["G2S_DUMP_STATS"] = dump
pg = P.Graph.from_seqs(seqs, k, 1)
sess = P.Session(pg, 0, d_err=d_err, randseed=1)
glist = [P.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gaps]
sess.fill_batch(glist, True)
open(dump, "w").close()
sess.fill_batch(glist, True)  # (the second call: caches warm)
rows = []
for ln in open(dump):
    if ln.startswith("#"):
        continue
    p = ln.split()
    rows.append(dict(gap=int(p[0]), g=int(p[1]), ra=int(p[3]), na=int(p[4]), rb=int(p[5]), nseg=int(p[6]), a=int(p[7]), b=int(p[8]), d=int(p[11])))
two = len(gaps) <= 2048 and os.environ.get("G2S_SEG_WAVES") != "1"
for r in rows:
    r["t"] = (r["b"] if two else r["a"] + r["b"]) + r["d"]
rows.sort(key=lambda r: -r["t"])
tot = sum(r["t"] for r in rows)
print("%s: %d gaps, %s per gap | wave cycles of a gap: mean %.0f k, median %.0f k, max %.0f k" % (
    cfg, len(rows), "two waves (phase A beside B)" if two else "one wave", tot / len(rows) / 1e3, rows[len(rows) // 2]["t"] / 1e3, rows[0]["t"] / 1e3))
for r in rows[:top]:
    print("gap %5d (length %4d): %4d k cycles = A %4d k (%3d rounds, %3d entries) | B+C %4d k (%3d rounds, %3d segments, %.0f a round) | tail %3d k" % (
        r["gap"], r["g"], r["t"] // 1000, r["a"] // 1000, r["ra"], r["na"], r["b"] // 1000, r["rb"], r["nseg"], r["b"] / max(1, r["rb"]), r["d"] // 1000))
