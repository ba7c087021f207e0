#!/usr/bin/env python3
"""GPU box: one failing configuration of the campaign, the gap the oracle flags Q7, through each kernel."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import cases, oracle_lib
import test_gpu_parity as tp
from gap2seq_amd import lib as P
cfg = json.loads(sys.argv[1])
k = cfg["k"]
seqs = cases.toy_genome(cfg["gseed"], cfg["length"], k, repeats=cfg["repeats"], tandem=cfg["tandem"], inverted=cfg["inverted"], snp_every=cfg["snp_every"])
gaps = cases.cut_gaps(cfg["cseed"], seqs[0], k, fuz=cfg["fuz"], ngaps=cfg["ngaps"], min_len=cfg["min_len"], max_len=cfg["max_len"], d_err=cfg["d_err"])
og = oracle_lib.OracleGraph(seqs, k, 1)
rng = oracle_lib.OracleRng(cfg["randseed"])
oq7 = []
for i, g in enumerate(gaps):
    o = oracle_lib.fill_gap(og, rng, g["left"], g["right"], g["gap_len"], cfg["d_err"], g["lmf"], g["rmf"], cfg["skip"], cfg["allp"])
    if o.info.q7:
        oq7.append(i)
print("oracle Q7 gaps:", oq7[:40], len(oq7))
pg = P.Graph.from_seqs(seqs, k, 1)
for name, env in (("seg", {}), ("segw", {"G2S_FORCE_SEGX": "1"}), ("segx1", {"G2S_FORCE_SEGX": "1", "G2S_SEGX_ONE_WAVE": "1"})):
    for kk in ("G2S_FORCE_SEGX", "G2S_SEGX_ONE_WAVE"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    os.environ["G2S_RESIDENT"] = "0"
    sess = P.Session(pg, 0, d_err=cfg["d_err"], skip_confident=cfg["skip"], all_paths=cfg["allp"], randseed=cfg["randseed"])
    res, tm = sess.fill_batch(tp._gaps(P, gaps), True)
    sess.destroy()
    missed = [i for i in oq7 if not (res[i].flags & P.G2S_GAP_Q7)]
    flagged = [i for i, r in enumerate(res) if r.flags & P.G2S_GAP_Q7]
    print(name, "flagged", len(flagged), "missed", missed[:20], "seg/segx gaps", tm.seg_tier_gaps, tm.segx_tier_gaps)
import tempfile
for name, env in (("seg", {}), ("segw", {"G2S_FORCE_SEGX": "1"})):
    for kk in ("G2S_FORCE_SEGX", "G2S_SEGX_ONE_WAVE"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    dump = tempfile.mktemp()
    os.environ["G2S_SEG_DUMP"] = dump
    os.environ["G2S_SEG_DUMP_BRIEF"] = "1"
    sess = P.Session(pg, 0, d_err=cfg["d_err"], skip_confident=cfg["skip"], all_paths=cfg["allp"], randseed=cfg["randseed"])
    res, tm = sess.fill_batch(tp._gaps(P, gaps), True)
    sess.destroy()
    rows = {}
    for ln in open(dump):
        p = ln.split()
        if p[0] == "gap":
            rows[int(p[1])] = ln.strip()
    for i in (47, 116, 133, 1, 7):
        print(name, rows.get(i), "| gap_len", gaps[i]["gap_len"], "lmf", gaps[i]["lmf"], "rmf", gaps[i]["rmf"])
os.environ.pop("G2S_SEG_DUMP_BRIEF", None)
for name, env in (("seg", {}), ("segw", {"G2S_FORCE_SEGX": "1"})):
    for kk in ("G2S_FORCE_SEGX", "G2S_SEGX_ONE_WAVE"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    dump = tempfile.mktemp()
    os.environ["G2S_SEG_DUMP"] = dump
    sess = P.Session(pg, 0, d_err=cfg["d_err"], skip_confident=cfg["skip"], all_paths=cfg["allp"], randseed=cfg["randseed"])
    res, tm = sess.fill_batch(tp._gaps(P, gaps[:48]), True)
    sess.destroy()
    cur = None
    segs = []
    for ln in open(dump):
        p = ln.split()
        if p[0] == "gap":
            cur = int(p[1])
        elif p[0] == "S" and cur == 47:
            segs.append((int(p[1]), int(p[2]), int(p[3]), int(p[4]), int(p[7])))
    print(name, "segments of gap 47 (node, depth, len, cnt, gen):")
    for s_ in sorted(segs, key=lambda x: (x[4], x[0])):
        print("   ", s_, "idx", s_[0] >> 1, "odd" if s_[0] & 1 else "even")
    # crossing test between every up and down segment; same-depth entries of both strands
    hits = []
    for a in segs:
        for b in segs:
            if (a[0] & 1) == 0 and (b[0] & 1) == 1:
                ia, da, la = a[0] >> 1, a[1], a[2]
                ib, db, lb = b[0] >> 1, b[1], b[2]
                for t1 in range(la):
                    t2 = ib - (ia + t1)
                    if 0 <= t2 < lb and da + t1 == db + t2:
                        hits.append((a, b, t1, t2))
    print(name, "crossings:", hits[:5])
