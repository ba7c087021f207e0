#!/usr/bin/env python3
"""Differential campaign on an MI355X: random graphs, gap lists and parameters, the HIP
fill path (through the C ABI) against the CPU oracle, gap by gap and bit-exact (the same
comparison as tests/test_gpu_parity.py::_check_batch).  Not part of the test suite: run
it on the GPU box for as long as the budget allows,

    python tools/fuzz_parity.py --seconds 600 --seed 1 > gpurun_out/fuzz.log

Every failing configuration is printed as one JSON line that reproduces it
(`--replay "$(cat cfg.json)"`); the exit code is the number of failing configurations (max 100).
"""
import argparse
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import cases  # noqa: E402
import oracle_lib  # noqa: E402
import test_gpu_parity as tp  # noqa: E402
from gap2seq_amd import lib as product  # noqa: E402

IN_FLIGHT = False  # (--in-flight)


def draw_big(rng):
    """Long gap lists and deep gaps: small LDS shares, the right-set spill pool, passes 1 and 2,
    gaps that start in a later pass, the HBM tier as the last resort."""
    length = rng.choice([300000, 1000000])
    units = length // 10000
    deep = rng.random() < 0.4
    return dict(
        k=rng.choice([31, 31, 21, 41]), length=length, repeats=units * rng.randint(0, 3), tandem=units * rng.randint(0, 1),
        inverted=0, snp_every=rng.choice([0, 211, 500, 1000]), fuz=rng.choice([10, 10, 20]),
        d_err=rng.choice([1000, 2000]) if deep else rng.choice([200, 500, 1000]),
        ngaps=rng.choice([200, 600]) if deep else rng.choice([1500, 4000]),
        min_len=rng.choice([1, 200, 1000]) if deep else rng.choice([1, 200]),
        max_len=rng.choice([2000, 5000]) if deep else rng.choice([300, 1000]),
        skip=rng.random() < 0.1, allp=rng.random() < 0.8, randseed=rng.randint(1, 1 << 20),
        hbm_tier=False, gseed=rng.randint(0, 1 << 30), cseed=rng.randint(0, 1 << 30),
    )


def draw(rng, big_share=0.08):
    if rng.random() < big_share:
        return draw_big(rng)
    k = rng.choice([9, 11, 13, 15, 21, 25, 31, 31, 31, 33, 41, 55, 63, 16, 32])
    length = rng.choice([2000, 5000, 20000, 60000, 200000])
    dens = rng.choice([0, 1, 1, 3, 8])  # structures per 10 kbp
    units = max(1, length // 10000)
    cfg = dict(
        k=k, length=length,
        repeats=dens * units * rng.randint(0, 2), tandem=dens * units * rng.randint(0, 1) // 2,
        inverted=rng.choice([0, 0, 0, 1, 2]) * (1 if dens else 0),
        snp_every=rng.choice([0, 0, 37, 83, 211, 500, 1000]),
        fuz=rng.choice([0, 1, 3, 10, 10, 10, 15, 20]),
        d_err=rng.choice([k, 50, 200, 500, 500, 1000]),
        ngaps=rng.choice([20, 100, 300, 600]),
        min_len=rng.choice([1, 20, 200]), max_len=rng.choice([60, 300, 1000, 1500]),
        skip=rng.random() < 0.15, allp=rng.random() < 0.75, randseed=rng.randint(1, 1 << 20),
        hbm_tier=rng.random() < 0.15, gseed=rng.randint(0, 1 << 30), cseed=rng.randint(0, 1 << 30),
    )
    if k <= 13:  # 4^k is small: the graph of a long genome is a tangle, every gap costs the oracle seconds
        cfg["length"] = min(length, 2000 if k == 9 else 5000)
        cfg["d_err"] = min(cfg["d_err"], 200)
        cfg["ngaps"] = min(cfg["ngaps"], 60)
        cfg["max_len"] = min(cfg["max_len"], 300)
        cfg["repeats"] = min(cfg["repeats"], 4)
        cfg["tandem"] = min(cfg["tandem"], 2)
        length = cfg["length"]
    cfg["max_len"] = max(cfg["max_len"], cfg["min_len"] + 1)
    cfg["max_len"] = min(cfg["max_len"], length // 4)
    cfg["min_len"] = min(cfg["min_len"], cfg["max_len"] - 1)
    return cfg


def run(cfg):
    if cfg.get("scaffold"):
        return run_scaffold(cfg)
    k = cfg["k"]
    seqs = cases.toy_genome(cfg["gseed"], cfg["length"], k, repeats=cfg["repeats"], tandem=cfg["tandem"],
                            inverted=cfg["inverted"], snp_every=cfg["snp_every"])
    gaps = cases.cut_gaps(cfg["cseed"], seqs[0], k, fuz=cfg["fuz"], ngaps=cfg["ngaps"], min_len=cfg["min_len"],
                          max_len=cfg["max_len"], d_err=cfg["d_err"])
    if cfg["hbm_tier"]:
        os.environ["G2S_NO_LDS_TIER"] = "1"
    else:
        os.environ.pop("G2S_NO_LDS_TIER", None)
    return tp._check_batch(product, oracle_lib, seqs, k, gaps, cfg["d_err"], cfg["skip"], cfg["allp"],
                           seed=cfg["randseed"], run_product=in_flight_runner(cfg) if IN_FLIGHT else None)


def in_flight_runner(cfg):
    """--in-flight: the configuration's gaps as three to six consecutive lists through g2s_fill_begin / g2s_fill_end, up
    to three of them in flight (one rand() stream from list to list — on the device when a list ends there, through
    the host when one falls back, which makes the lists behind it run again): the results in order are those of one
    list, which is what _check_batch compares with the oracle."""
    def runner(sess, gap_structs):
        rng = cases.SplitMix(cfg["cseed"] ^ 0x5EED)
        n = len(gap_structs)
        parts = min(n, rng.randint(3, 6))
        cuts = sorted({0, n} | {rng.randint(1, max(1, n - 1)) for _ in range(parts - 1)})
        lists = [gap_structs[a:b] for a, b in zip(cuts, cuts[1:]) if b > a]
        outs, tm = sess.fill_lists_overlapped(lists, pinned=rng.random() < 0.7, depth=rng.randint(2, 3))
        return [r for part in outs for r in part], tm
    return runner


def draw_scaffold(rng):
    k = rng.choice([5, 7, 11, 15, 21, 31])
    cfg = dict(
        scaffold=True, k=k, length=rng.choice([3000, 8000, 30000]), repeats=rng.randint(0, 6), tandem=rng.randint(0, 2),
        inverted=rng.choice([0, 0, 1]), snp_every=rng.choice([0, 97, 211]), fuz=rng.choice([1, 4, 8, 10, 12]),
        d_err=rng.choice([k, 20, 60, 200]), records=rng.choice([5, 25, 60]), max_gaps=rng.randint(1, 6),
        max_len=rng.choice([10, 40, 200]), mode=rng.choice(["default", "default", "unique", "best_only", "all_upper"]),
        randseed=rng.randint(1, 1 << 20), gseed=rng.randint(0, 1 << 30), cseed=rng.randint(0, 1 << 30),
    )
    if k <= 11:  # tangled graphs: keep the searches shallow (cf. draw)
        cfg["length"] = min(cfg["length"], 8000)
        cfg["d_err"] = min(cfg["d_err"], 30 if k <= 7 else 60)
        cfg["max_len"] = min(cfg["max_len"], 40)
    if k <= 7:
        cfg["length"] = min(cfg["length"], 600 if k == 5 else 2000)
        cfg["records"] = min(cfg["records"], 25)
    return cfg


def run_scaffold(cfg):
    """Whole-scaffold runs (the gap scan, the couplings between consecutive gaps of a record,
    the splice and the log text): FASTA, log and counters against the oracle's execute."""
    k, fuz, e = cfg["k"], cfg["fuz"], cfg["d_err"]
    seqs = cases.toy_genome(cfg["gseed"], cfg["length"], k, repeats=cfg["repeats"], tandem=cfg["tandem"],
                            inverted=cfg["inverted"], snp_every=cfg["snp_every"])
    g = seqs[0]
    rng = cases.SplitMix(cfg["cseed"])
    recs = []
    for r in range(cfg["records"]):
        span = cfg["max_gaps"] * (cfg["max_len"] + 3 * k + 2 * fuz + 10) + 2 * (k + fuz + 10)
        start = rng.randint(0, max(1, len(g) - span - 1))
        pos = start + k + fuz + 5
        triples = []
        for _ in range(rng.randint(1, cfg["max_gaps"])):
            ln = rng.randint(1, cfg["max_len"])
            triples.append((pos, ln, max(1, ln + k + rng.randint(-3, 3))))
            pos += ln + rng.choice([k + fuz - 1, k + fuz, k + fuz + 1, k + 2 * fuz, 3 * k + 2 * fuz + 7])
        rec = cases.scaffold_record(g, k, fuz, triples)
        if rng.random() < 0.2:
            rec = rec.replace("N", "n")
        recs.append(("rec%d extra words" % r, rec))
    text = "".join(">%s\n%s\n" % x for x in recs)
    kw = dict(default={}, unique=dict(unique_paths=True), best_only=dict(all_paths=False),
              all_upper=dict(skip_confident=True))[cfg["mode"]]
    og = oracle_lib.OracleGraph(seqs, k, 1)
    pg = product.Graph.from_seqs(seqs, k, 1)
    try:
        ofa, olog, sm = oracle_lib.execute_scaffolds(og, text, k, solid=1, d_err=e, max_fuz=fuz,
                                                     randseed=cfg["randseed"], **kw)
        if sm.q7_gaps:
            return 0, 0
        sess = product.Session(pg, 0, d_err=e, randseed=cfg["randseed"], **kw)
        try:
            if IN_FLIGHT:  # (the same records batch by batch, batches in flight where they are long enough)
                chunk = cases.SplitMix(cfg["cseed"] ^ 0xC4).choice([1, 3, 10, 40, 300])
                fas, lgs, gaps, filled = sess.execute_scaffolds_stream(text, k, chunk, solid=1, max_fuz=fuz)
                fa, lg = "".join(fas), "".join(lgs)
            else:
                fa, lg, gaps, filled = sess.execute_scaffolds(text, k, solid=1, max_fuz=fuz)
        finally:
            sess.destroy()
        assert fa == ofa, "FASTA differs"
        assert lg == olog, "log differs"
        assert (gaps, filled) == (sm.gaps, sm.filled), "counters differ"
        return sm.gaps, sm.filled
    finally:
        og.free()
        pg.free()


def explain(cfg):
    """The comparison of _check_batch again, printing every field that differs."""
    k = cfg["k"]
    seqs = cases.toy_genome(cfg["gseed"], cfg["length"], k, repeats=cfg["repeats"], tandem=cfg["tandem"],
                            inverted=cfg["inverted"], snp_every=cfg["snp_every"])
    gaps = cases.cut_gaps(cfg["cseed"], seqs[0], k, fuz=cfg["fuz"], ngaps=cfg["ngaps"], min_len=cfg["min_len"],
                          max_len=cfg["max_len"], d_err=cfg["d_err"])
    if cfg["hbm_tier"]:
        os.environ["G2S_NO_LDS_TIER"] = "1"
    if os.environ.get("G2S_FUZZ_ONLY"):  # indices of the gaps to keep (the rand() stream then differs, counts do not)
        gaps = [gaps[int(x)] for x in os.environ["G2S_FUZZ_ONLY"].split(",")]
    og = oracle_lib.OracleGraph(seqs, k, 1)
    pg = product.Graph.from_seqs(seqs, k, 1)
    sess = product.Session(pg, 0, d_err=cfg["d_err"], skip_confident=cfg["skip"], all_paths=cfg["allp"],
                           randseed=cfg["randseed"])
    res, _ = sess.fill_batch(tp._gaps(product, gaps), True)
    rng = oracle_lib.OracleRng(cfg["randseed"])
    used = 0
    for i, (g, r) in enumerate(zip(gaps, res)):
        o = oracle_lib.fill_gap(og, rng, g["left"], g["right"], g["gap_len"], cfg["d_err"], g["lmf"], g["rmf"],
                                cfg["skip"], cfg["allp"])
        used += r.draws
        q7 = bool(o.info.q7), bool(r.flags & product.G2S_GAP_Q7)
        diffs = []
        if q7[0] and not q7[1]:
            diffs.append("oracle q7, gpu not")
        if not q7[0] and r.count != -1:
            for name, a, b in (("count", r.count, o.count), ("phaseC", r.phaseC_count, o.info.phaseC_count),
                               ("lengths", r.lengths, o.lengths), ("draws", r.draws, o.info.draws)):
                if a != b:
                    diffs.append(f"{name}: gpu {a} oracle {b}")
            if o.phase_d:
                if (r.left_fuz, r.right_fuz) != (o.left_fuz, o.right_fuz):
                    diffs.append(f"fuz: gpu {(r.left_fuz, r.right_fuz)} oracle {(o.left_fuz, o.right_fuz)}")
                if r.fill != o.fill:
                    pos = next((x for x in range(min(len(r.fill), len(o.fill))) if r.fill[x] != o.fill[x]), -1)
                    diffs.append(f"fill differs at {pos} (len gpu {len(r.fill)} oracle {len(o.fill)}): "
                                 f"gpu ..{r.fill[max(0, pos - 5):pos + 10]} oracle ..{o.fill[max(0, pos - 5):pos + 10]}")
                if not cfg["skip"] and r.substats != o.substats:
                    diffs.append(f"substats: gpu {r.substats} oracle {o.substats}")
        if o.info.draws != r.draws:  # put the oracle's stream where the product's is
            rng = oracle_lib.OracleRng(cfg["randseed"], used)
        if diffs:
            print(f"gap {i} (g {g['gap_len']} lmf {g['lmf']} rmf {g['rmf']} flags {r.flags:#x}): " + "; ".join(diffs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", type=float, default=0.08, help="share of long-list / deep-gap configurations")
    ap.add_argument("--scaffold", type=float, default=0.1, help="share of whole-scaffold (execute) configurations")
    ap.add_argument("--replay", default=None)
    ap.add_argument("--in-flight", action="store_true", help="every configuration's gaps as several lists in flight")
    a = ap.parse_args()
    global IN_FLIGHT
    IN_FLIGHT = a.in_flight
    oracle_lib.lib()
    product.load_library()
    if a.replay:
        cfg = json.loads(a.replay)
        if cfg.get("scaffold"):
            print(run_scaffold(cfg))
        else:
            explain(cfg)
        return 0
    rng = cases.SplitMix(a.seed * 1000003 + 17)
    t_end = time.time() + a.seconds
    n = bad = compared = filled = 0
    while time.time() < t_end and bad < 100:
        cfg = draw_scaffold(rng) if rng.random() < a.scaffold else draw(rng, a.big)
        n += 1
        try:
            c, f = run(cfg)[:2]
            compared += c
            filled += f
        except Exception as ex:  # noqa: BLE001 (report and go on)
            bad += 1
            print("FAIL " + json.dumps(cfg), flush=True)
            print("     " + "".join(traceback.format_exception_only(type(ex), ex)).strip()[:600], flush=True)
        if n % 50 == 0:
            print(f"# {n} configurations, {compared} gaps compared ({filled} filled), {bad} failing", flush=True)
    print(f"# done: {n} configurations, {compared} gaps compared ({filled} filled), {bad} failing", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(main())
