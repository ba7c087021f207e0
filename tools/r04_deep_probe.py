#!/usr/bin/env python3
"""GPU box: which gaps of a deep list outgrow the regular tier, by gap length — can the host tell in advance?
(instrumented library as for tools/segw_profile.py)  usage: python tools/r04_deep_probe.py [C5]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("G2S_LIBRARY", os.path.join(ROOT, "gap2seq_amd", "_prof", "libg2s_hip.so"))
os.environ["G2S_RESIDENT"] = "0"
os.environ["G2S_SEG_DUMP_BRIEF"] = "1"
import bench
from gap2seq_amd import lib as P
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
genome_bp, k, ngaps, min_len, max_len, d_err, _ = bench.CONFIGS[cfg]
reads = P.G2S.synth_genome(genome_bp, 3, 20240101)
seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
gaps = bench.parse_gaps(P.G2S.synth_gaps(reads, k, 10, ngaps, min_len, max_len, 20240103), 10)
dump = tempfile.mktemp()
os.environ["G2S_SEG_DUMP"] = dump
pg = P.Graph.from_seqs(seqs, k, 1)
sess = P.Session(pg, 0, d_err=d_err, randseed=1)
res, tm = sess.fill_batch([P.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gaps], True)
rows_, cur = [], None
for ln in open(dump):
    p = ln.split()
    if p[0] == "gap":
        cur = dict(gap=int(p[1]), nseg=int(p[5]), rounds=int(p[11]))
        rows_.append(cur)
    elif p[0] == "PA":
        cur["pa"] = 1
    elif p[0] == "P":
        cur["b"] = 1
big = {r["gap"]: r for r in rows_ if r.get("b") and r.get("pa") and r["rounds"] > 0}  # (as tools/segw_profile.py: the large variant's rows)
rows = sorted(range(len(gaps)), key=lambda i: gaps[i]["gap_len"])
print("%d gaps, %d through the large variant" % (len(gaps), len(big)))
step = len(rows) // 10
for d in range(10):
    part = rows[d * step:(d + 1) * step]
    nb = sum(1 for i in part if i in big)
    print("gap length %4d-%4d: %3d of %3d outgrow the regular tier; their rounds in the large variant: %s" % (
        gaps[part[0]]["gap_len"], gaps[part[-1]]["gap_len"], nb, len(part),
        sorted(big[i]["rounds"] for i in part if i in big)[-3:]))
