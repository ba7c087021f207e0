#!/usr/bin/env python3
"""GPU box: the segment tier's kernel against its executable model (tests/seg_model.py), structure
by structure — phase A's entries (node -> label) and phase B's segments (node, depth, length,
count) of every gap — so that a difference is located in the kernel, not only seen in the results.
Usage: python tools/seg_check.py [genome_bp variant ngaps min_len max_len d_err k]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from gap2seq_amd import lib as P  # noqa: E402
import seg_model as M  # noqa: E402
import test_seg_model as T  # noqa: E402
from test_gpu_parity import _parse_scaffolds  # noqa: E402

a = [int(x) for x in sys.argv[1:]] + [200000, 3, 80, 50, 600, 500, 31][len(sys.argv) - 1:]
genome, variant, ngaps, lo, hi, e, k = a
reads = P.G2S.synth_genome(genome, variant, 20240101)
seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
gaps = _parse_scaffolds(P.G2S.synth_gaps(reads, k, 10, ngaps, lo, hi, 20240103))
dump = tempfile.mktemp()
os.environ["G2S_SEG_DUMP"] = dump
pg = P.Graph.from_seqs(seqs, k, 1)
sess = P.Session(pg, 0, d_err=e, randseed=1)
res, tm = sess.fill_batch([P.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gaps], True)
print("seg tier gaps", tm.seg_tier_gaps, "of", len(gaps), "lds tier", tm.lds_tier_gaps, "segments", tm.seg_segments)
tb = M.Tables(P, pg)
cur = None
dev = {}
for ln in open(dump):
    p = ln.split()
    if p[0] == "gap":
        cur = dev[int(p[1])] = dict(A={}, S=[], flags=int(p[7], 16), c_count=int(p[13]), nseg=int(p[5]))
    elif p[0] == "A":
        cur["A"][int(p[1])] = int(p[2])
    elif p[0] == "S":
        cur["S"].append((int(p[1]), int(p[2]), int(p[3]), int(p[4])))
bad = 0
for gi, g in enumerate(gaps):
    mg = T._model_gap(pg, k, g, e)
    lab = M.right_entries(tb, mg)
    m = M.fill_model(tb, mg)
    d = dev.get(gi)
    if d is None:
        print("gap", gi, "missing in dump"); bad += 1; continue
    if d["flags"] & 0xC:
        print("gap", gi, "overflow flags %#x (model: %d entries, %d segments)" % (d["flags"], len(lab), m.n_seg)); continue
    msgs = []
    if d["A"] != lab:
        only_d = {x: d["A"][x] for x in d["A"] if lab.get(x) != d["A"][x]}
        only_m = {x: lab[x] for x in lab if d["A"].get(x) != lab[x]}
        msgs.append("A differs: device-only/changed %s model-only/changed %s" % (list(only_d.items())[:5], list(only_m.items())[:5]))
    ms = sorted((s[0], s[1], s[3], s[2]) for s in m.segs)
    ds = sorted(d["S"])
    if ms != ds:
        sm, sd = set(ms), set(ds)
        msgs.append("segments differ (%d model, %d device): model-only %s device-only %s" % (len(ms), len(ds), sorted(sm - sd)[:4], sorted(sd - sm)[:4]))
    if d["c_count"] != m.c_count:
        msgs.append("c_count device %d model %d" % (d["c_count"], m.c_count))
    r = res[gi]
    if (r.phaseC_count, r.lengths) != (m.c_count, m.lengths):
        msgs.append("phase C: device %s model %s" % ((r.phaseC_count, r.lengths), (m.c_count, m.lengths)))
    if msgs:
        bad += 1
        print("gap", gi, "g", g["gap_len"], "; ".join(msgs))
print("checked", len(gaps), "gaps,", bad, "differ")
sys.exit(1 if bad else 0)
