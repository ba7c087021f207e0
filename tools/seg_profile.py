#!/usr/bin/env python3
"""GPU box, library built with -DG2S_SEG_PROFILE: where a phase B round of the segment tier spends its cycles
(records wait | horizon | lengths + hits | children), summed over the gaps of a bench configuration and for the
slowest gaps.  usage: python tools/seg_profile.py [C2|C3]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gap2seq_amd import lib as P  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
ngaps_override = int(sys.argv[2]) if len(sys.argv) > 2 else 0
genome_bp, k, ngaps, min_len, max_len, d_err, _ = bench.CONFIGS[cfg]
reads = P.G2S.synth_genome(genome_bp, 3, 20240101)
seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
gaps = bench.parse_gaps(P.G2S.synth_gaps(reads, k, 10, ngaps, min_len, max_len, 20240103), 10)
if ngaps_override:
    gaps = gaps[:ngaps_override]
if os.environ.get("G2S_PROFILE_ONLY"):  # these gaps alone (ids of the full list): each has a compute unit to itself
    only = [int(x) for x in os.environ["G2S_PROFILE_ONLY"].split(",")]
    gaps = [gaps[i] for i in only]
    print("gaps", only, "alone: gap i below is the i-th of these")
dump = tempfile.mktemp()
os.environ["G2S_SEG_DUMP"] = dump
pg = P.Graph.from_seqs(seqs, k, 1)
sess = P.Session(pg, 0, d_err=d_err, randseed=1)
res, tm = sess.fill_batch([P.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gaps], True)
rows = []
cur = None
for ln in open(dump):
    p = ln.split()
    if p[0] == "gap":
        cur = dict(gap=int(p[1]), nseg=int(p[5]), rounds=int(p[11]), rounds_a=int(p[9]), entries=int(p[3]), prof=None)
        rows.append(cur)
    elif p[0] == "PA":
        cur["pa"] = [int(x) for x in p[1:5]]
    elif p[0] == "P":
        cur["prof"] = [int(x) for x in p[1:5]]
        cur["tail"] = [int(x) for x in p[5:9]]
        cur["sync"] = [int(x) for x in p[9:11]] if len(p) >= 11 else [0, 0]
rows = [r for r in rows if r["prof"]]
tot = [sum(r["prof"][i] for r in rows) for i in range(4)]
allc = sum(tot)
print("gaps %d rounds %d segments %d | cycles per round %.0f: wait %.1f%% horizon %.1f%% lengths+hits %.1f%% children %.1f%%" % (
    len(rows), sum(r["rounds"] for r in rows), sum(r["nseg"] for r in rows), allc / max(1, sum(r["rounds"] for r in rows)),
    100.0 * tot[0] / allc, 100.0 * tot[1] / allc, 100.0 * tot[2] / allc, 100.0 * tot[3] / allc))
for r in sorted(rows, key=lambda r: -sum(r["prof"]))[:6]:
    t = sum(r["prof"])
    print("gap %d: rounds %d segments %d cycles %d (%.0f per round): wait %d horizon %d lengths+hits %d children %d" % (
        r["gap"], r["rounds"], r["nseg"], t, t / max(1, r["rounds"]), *r["prof"]))
tt = [sum(r["tail"][i] for r in rows) for i in range(4)]
print("tail (Q7 between segments + phase C | D1 | D2 | emission), all gaps: %s; share of tail %s" % (
    tt, ["%.1f%%" % (100.0 * x / max(1, sum(tt))) for x in tt]))
for r in sorted(rows, key=lambda r: -sum(r["tail"]))[:6]:
    print("gap %d: segments %d tail cycles %d: q7+C %d D1 %d D2 %d emission %d" % (r["gap"], r["nseg"], sum(r["tail"]), *r["tail"]))
ws = sum(r["sync"][0] for r in rows); ls = sum(r["sync"][1] for r in rows)
print("two waves: cycles waiting at the barrier for phase A %d, loading the right set behind it %d (all gaps)" % (ws, ls))
for r in sorted(rows, key=lambda r: -(sum(r["prof"]) + sum(r["tail"]) + sum(r["sync"])))[:6]:
    print("gap %d: B rounds %d, wait for A %d, load %d, tail %d" % (r["gap"], sum(r["prof"]), r["sync"][0], r["sync"][1], sum(r["tail"])))

pa = [r for r in rows if r.get("pa")]
if pa:  # (one wave per gap: G2S_SEG_WAVES=1)
    ta = [sum(r["pa"][i] for r in pa) for i in range(4)]
    ra = sum(r["rounds_a"] for r in pa)
    print("phase A: %d rounds, %d entries | cycles per round %.0f: records wait %.1f%% probes + atomics %.1f%% results, slow path, queue %.1f%% rest of the round %.1f%%" % (
        ra, sum(r["entries"] for r in pa), sum(ta) / max(1, ra), *[100.0 * x / max(1, sum(ta)) for x in ta]))
    for r in sorted(pa, key=lambda r: -sum(r["pa"]))[:6]:
        print("gap %d: A rounds %d entries %d cycles %d: wait %d probes %d results %d rest %d" % (r["gap"], r["rounds_a"], r["entries"], sum(r["pa"]), *r["pa"]))
