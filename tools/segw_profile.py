#!/usr/bin/env python3
"""GPU box, library built with -DG2S_SEGW_PROFILE (gap2seq_amd/_prof/libg2s_hip.so, see csrc/Makefile): where the large
variant on eight waves (fill_segw.hip) spends its cycles, as wave 0 sees them — per round of phases A and B, and the
tail — summed over the gaps that take it and for the slowest ones.  usage: python tools/segw_profile.py [C5] [ngaps]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("G2S_LIBRARY", os.path.join(ROOT, "gap2seq_amd", "_prof", "libg2s_hip.so"))
os.environ["G2S_RESIDENT"] = "0"
os.environ["G2S_SEG_DUMP_BRIEF"] = "1"
import bench  # noqa: E402
from gap2seq_amd import lib as P  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
ngaps_override = int(sys.argv[2]) if len(sys.argv) > 2 else 0
genome_bp, k, ngaps, min_len, max_len, d_err, _ = bench.CONFIGS[cfg]
reads = P.G2S.synth_genome(genome_bp, 3, 20240101)
seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
gaps = bench.parse_gaps(P.G2S.synth_gaps(reads, k, 10, ngaps, min_len, max_len, 20240103), 10)
if ngaps_override:
    gaps = gaps[:ngaps_override]
dump = tempfile.mktemp()
os.environ["G2S_SEG_DUMP"] = dump
pg = P.Graph.from_seqs(seqs, k, 1)
sess = P.Session(pg, 0, d_err=d_err, randseed=1)
res, tm = sess.fill_batch([P.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gaps], True)
rows, cur = [], None
for ln in open(dump):
    p = ln.split()
    if p[0] == "gap":
        cur = dict(gap=int(p[1]), nseg=int(p[5]), rounds=int(p[11]), rounds_a=int(p[9]), entries=int(p[3]))
        rows.append(cur)
    elif p[0] == "PA":
        cur["pa"] = [int(x) for x in p[1:5]]
    elif p[0] == "P":
        cur["b"] = [int(x) for x in p[1:5]]       # scan + records | horizon barrier | selected + children | end barrier
        cur["tail"] = [int(x) for x in p[5:9]]    # hits + Q7 | D1 | D2 + recount | emission
        cur["sel"] = [int(x) for x in p[9:11]]    # selected events (length, segment) | children
rows = [r for r in rows if r.get("b") and r.get("pa") and r["rounds"] > 0]
print("%s: %d gaps through the eight-wave variant (of %d; seg tier %d)" % (cfg, len(rows), len(gaps), tm.seg_tier_gaps))
R = sum(r["rounds"] for r in rows)
tb = [sum(r["b"][i] for r in rows) for i in range(4)]
ts = [sum(r["sel"][i] for r in rows) for i in range(2)]
print("phase B: %d rounds, %d segments | cycles per round %.0f: scan + records %.0f, horizon barrier %.0f, selected events %.0f + children %.0f, end barrier %.0f" % (
    R, sum(r["nseg"] for r in rows), sum(tb) / R, tb[0] / R, tb[1] / R, ts[0] / R, ts[1] / R, tb[3] / R))
RA = sum(r["rounds_a"] for r in rows)
ta = [sum(r["pa"][i] for r in rows) for i in range(4)]
print("phase A: %d rounds, %d entries | cycles per round %.0f: queue + records %.0f, label + proposals %.0f, barriers %.0f; packing + sort + merge per gap %.0f" % (
    RA, sum(r["entries"] for r in rows), sum(ta[:3]) / RA, ta[0] / RA, ta[1] / RA, ta[2] / RA, ta[3] / len(rows)))
tt = [sum(r["tail"][i] for r in rows) for i in range(4)]
print("tail per gap: hits + Q7 %.0f, D1 %.0f, D2 + recount %.0f, emission %.0f" % tuple(x / len(rows) for x in tt))
for r in sorted(rows, key=lambda r: -(sum(r["b"]) + sum(r["pa"]) + sum(r["tail"])))[:6]:
    print("gap %d: A %d rounds %d entries %d cyc (+%d pack) | B %d rounds %d segments %d cyc = %.0f per round (scan %d, barrier %d, selected %d, children %d, barrier %d) | tail %d (hits+Q7 %d, D1 %d, D2 %d, emission %d)" % (
        r["gap"], r["rounds_a"], r["entries"], sum(r["pa"][:3]), r["pa"][3], r["rounds"], r["nseg"], sum(r["b"]), sum(r["b"]) / r["rounds"],
        r["b"][0], r["b"][1], r["sel"][0], r["sel"][1], r["b"][3], sum(r["tail"]), *r["tail"]))
if os.environ.get("SEGW_FINE"):  # library built with -DG2S_SEGW_PROFILE=2: the pass over the final events in eight sections
    f = [sum(r["pa"][i] for r in rows) for i in range(4)] + [sum(r["tail"][i] for r in rows) for i in range(4)]
    names = ["list + fields + segment ids", "length", "lead: free + segment", "child search", "insert: slot", "insert: probe", "insert: merge", "insert: Q7"]
    print("pass over the final events, cycles per round: " + ", ".join("%s %.0f" % (n, v / R) for n, v in zip(names, f)))
