"""tests/golden/make_golden.py — regenerates the golden vectors in this directory.

The reference (/root/reference) has no tests or fixtures and cannot be built in
this image, so these vectors are produced by the repo's C++ oracle
(oracle/g2s_oracle.cpp) and are committed ONLY after the independent Python
restatement (oracle/pyref.py) reproduced every field.  They pin the oracle against
silent drift and give the GPU tests inputs with known answers.  Data only: inputs
and expected outputs, no reference source text.

Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cases  # noqa: E402
import oracle_lib as O  # noqa: E402
import pyref  # noqa: E402


def one_set(name, seed, k, length, e, fuz, ngaps, skip, allp, min_len=1, max_len=80, **gk):
    seqs = cases.toy_genome(seed, length, k, **gk)
    gaps = cases.cut_gaps(seed, seqs[0], k, fuz, ngaps, min_len, max_len, e)
    og = O.OracleGraph(seqs, k, 1)
    pg = pyref.Graph(seqs, k, 1)
    rng, prng = O.OracleRng(11), pyref.GlibcRand(11)
    exp = []
    for g in gaps:
        o = O.fill_gap(og, rng, g["left"], g["right"], g["gap_len"], e, g["lmf"], g["rmf"], skip, allp)
        pi = pyref.Info()
        c2, lf2, rf2, fill2, sub2 = pyref.fill_gap(pg, prng, g["left"], g["right"], g["gap_len"], e, g["lmf"],
                                                   g["rmf"], skip, allp, True, pi)
        assert (o.count, o.left_fuz, o.right_fuz, o.info.draws, o.info.q7) == (c2, lf2, rf2, pi.draws, pi.q7), name
        if o.phase_d:
            assert o.fill == pyref.fill_string(fill2, g["lmf"] - lf2), name
            if sub2 is not None:
                assert o.substats == [sub2[x] for x in ("vertices", "edges", "nontrivial", "size_nontrivial",
                                                        "vertices_final", "edges_final")], name
        exp.append(dict(count=o.count, left_fuz=o.left_fuz, right_fuz=o.right_fuz, fill=o.fill, draws=o.info.draws,
                        q7=o.info.q7, phaseC_count=o.info.phaseC_count, lengths=o.lengths,
                        substats=o.substats if (o.phase_d and not skip) else None,
                        ctr=[int(x) for x in o.info.ctr]))
    og.free()
    return dict(name=name, k=k, solid=1, d_err=e, skip_confident=skip, all_paths=allp, randseed=11, seqs=seqs,
                gaps=gaps, expected=exp)


def scaffold_set():
    """Scaffold-mode fixture: multi-gap records (Q8), a gap too close to the record
    start (Q9), a truncated right flank (Q10/D2), close gaps (right_fuz coupling)."""
    k, fuz, e = 9, 4, 20
    seqs = cases.toy_genome(5, 1500, k, repeats=2, tandem=1, snp_every=131)
    g = seqs[0]
    recs = []
    recs.append(("two_gaps", cases.scaffold_record(g, k, fuz, [(100, 30, 30 + k), (300, 12, 12 + k)])))
    recs.append(("close_gaps", cases.scaffold_record(g, k, fuz, [(500, 10, 10 + k), (500 + 10 + k + fuz + 1, 8, 8 + k)])))
    recs.append(("start_gap", "NNNNN" + g[700:760]))
    recs.append(("short_right", g[800:860] + "NNNNNNNN" + g[868:872]))
    recs.append(("lower_n", g[900:960] + "nnnnNNNNnnnn" + g[972 - k:1040]))
    recs.append(("no_gap", g[1100:1160]))
    text = "".join(">%s\n%s\n" % r for r in recs)
    out = {}
    for mode, kw in (("default", {}), ("best_only", dict(all_paths=False)), ("all_upper", dict(skip_confident=True)),
                     ("unique", dict(unique_paths=True))):
        og = O.OracleGraph(seqs, k, 1)
        fa, lg, sm = O.execute_scaffolds(og, text, k, solid=1, d_err=e, max_fuz=fuz, randseed=3, **kw)
        pfa, plg, pf, pgaps = pyref.execute_scaffolds(pyref.Graph(seqs, k, 1), [(c, s) for c, s in recs], k, e, fuz, 3,
                                                      skip_confident=kw.get("skip_confident", False),
                                                      unique=kw.get("unique_paths", False),
                                                      all_paths=kw.get("all_paths", True))
        assert fa == pfa, mode
        body = "".join(ln + "\n" for ln in lg.splitlines()
                       if ln.startswith(("Scaffold:", "SubgraphStats:", "Unable")) or
                       (ln.startswith("Filled ") and ln[7].isdigit()))
        assert body == plg, (mode, body, plg)
        out[mode] = dict(fasta=fa, log=lg)
        og.free()
    return dict(k=k, solid=1, d_err=e, max_fuz=fuz, randseed=3, seqs=seqs, scaffolds=text, expected=out)


def main():
    sets = [
        one_set("linear_k9", 1, 9, 700, 9 + 12, 3, 12, False, True),
        one_set("repeats_k7", 2, 7, 700, 7 + 9, 2, 12, False, True, repeats=4),
        one_set("tandem_k11", 3, 11, 800, 11 + 20, 4, 12, False, True, tandem=3),
        one_set("bubbles_k13", 4, 13, 900, 13 + 31, 5, 12, False, True, snp_every=71),
        one_set("bubbles_best_only_k13", 4, 13, 900, 13 + 31, 5, 12, False, False, snp_every=71),
        one_set("all_upper_k9", 6, 9, 700, 9 + 12, 3, 12, True, True, repeats=2, snp_every=97),
        one_set("inverted_k15", 7, 15, 900, 15 + 12, 3, 12, False, True, inverted=3, repeats=1),
        one_set("even_k12", 8, 12, 900, 12 + 15, 3, 12, False, True, repeats=2, snp_every=89),
        one_set("wide_k33", 9, 33, 1600, 33 + 40, 6, 8, False, True, repeats=1, snp_every=150),
        # SURVEY 8(c) pin (5): the 128-bit k-mer codec (k = 63 is BASELINE config 4's)
        one_set("wide_k63", 10, 63, 2600, 63 + 60, 8, 8, False, True, repeats=1, snp_every=210),
        # -dist-error 2000 (BASELINE config 5's): thousands of DP levels, a tandem repeat and bubbles inside them
        one_set("deep_e2000", 13, 21, 9000, 2000, 10, 3, False, True, min_len=600, max_len=1500, repeats=2, tandem=2,
                snp_every=260),
    ]
    with open(os.path.join(HERE, "fill_gap_cases.json"), "w") as f:
        json.dump(sets, f, indent=0, sort_keys=True)
    with open(os.path.join(HERE, "scaffold_cases.json"), "w") as f:
        json.dump(scaffold_set(), f, indent=0, sort_keys=True)
    print("wrote", sum(len(s["gaps"]) for s in sets), "gap vectors")


if __name__ == "__main__":
    main()
