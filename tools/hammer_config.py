#!/usr/bin/env python3
"""GPU box: configurations of tools/fuzz_parity.py (one JSON object a line in FILE, as a campaign prints them behind
"FAIL") run N times each in ONE process against the oracle — for failures that do not show in a single replay: the
missing barrier in g2s_fill_segw's tail (round 5) lost a gap's traceback start once in a hundred runs of one list.
On a failure: the assertion and the call's timing counters (which path the list took).

  G2S_FORCE_SEGX=1 G2S_DEBUG_DRAWS=1 python tools/hammer_config.py FILE N
  python tools/hammer_config.py draw:SEED:COUNT:BIG N     (COUNT configurations drawn as a campaign with --seed SEED --big BIG would)
"""
import json, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_parity as F
if sys.argv[1].startswith("draw:"):
    _, seed, count, big = sys.argv[1].split(":")
    rng = F.cases.SplitMix(int(seed) * 1000003 + 17)
    cfgs = [F.draw(rng, float(big)) for _ in range(int(count))]
else:
    cfgs = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
n = int(sys.argv[2])
F.oracle_lib.lib(); F.product.load_library()
last = {}
def runner(sess, gs):
    res, tm = sess.fill_batch(gs, True)
    last["tm"] = tm
    return res, tm
bad = 0
for it in range(n):
    for cfg in cfgs:
        k = cfg["k"]
        seqs = F.cases.toy_genome(cfg["gseed"], cfg["length"], k, repeats=cfg["repeats"], tandem=cfg["tandem"], inverted=cfg["inverted"], snp_every=cfg["snp_every"])
        gaps = F.cases.cut_gaps(cfg["cseed"], seqs[0], k, fuz=cfg["fuz"], ngaps=cfg["ngaps"], min_len=cfg["min_len"], max_len=cfg["max_len"], d_err=cfg["d_err"])
        try:
            F.tp._check_batch(F.product, F.oracle_lib, seqs, k, gaps, cfg["d_err"], cfg["skip"], cfg["allp"], seed=cfg["randseed"], run_product=runner)
        except Exception as ex:
            bad += 1
            tm = last.get("tm")
            print("FAIL iter", it, "".join(traceback.format_exception_only(type(ex), ex)).strip()[:300], flush=True)
            if tm is not None:
                print("   timing:", {f: getattr(tm, f) for f in ("resident_launches", "resident_fallbacks", "host_finished_gaps", "seg_tier_gaps", "segx_tier_gaps", "watchdog_gaps", "seg_launches", "segx_launches") if hasattr(tm, f)}, flush=True)
print("runs", n * len(cfgs), "failures", bad)
