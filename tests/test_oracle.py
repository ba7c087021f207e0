"""CPU tests that pin the oracle (oracle/g2s_oracle.cpp).

The reference has no tests, golden vectors or fixtures for this path and cannot be
built here, so the pins are: glibc rand() known answers, an
independent Python restatement (oracle/pyref.py), brute-force walk counts on toy
graphs, hand-built quirk fixtures and the committed golden vectors.
"""
import ctypes
import json
import os

import pytest

import cases
import pyref

HERE = os.path.dirname(os.path.abspath(__file__))


def test_glibc_rand_known_answers(oracle):
    # /root/reference/src/Gap2Seq.cpp:178,1440,1513 use libc rand(); SURVEY §8c known answers
    r = oracle.OracleRng(1)
    assert [r.next() for _ in range(5)] == [1804289383, 846930886, 1681692777, 1714636915, 1957747793]
    r = oracle.OracleRng(42)
    assert [r.next() for _ in range(3)] == [71876166, 708592740, 1483128881]
    p = pyref.GlibcRand(1)
    assert [p.next() for _ in range(5)] == [1804289383, 846930886, 1681692777, 1714636915, 1957747793]


def test_glibc_rand_matches_this_libc(oracle):
    libc = ctypes.CDLL("libc.so.6")
    for seed in (1, 2, 7, 12345, 20240101):
        libc.srand(seed)
        r = oracle.OracleRng(seed)
        for _ in range(2000):
            assert libc.rand() == r.next()


def _compare_with_pyref(oracle, seqs, k, gaps, e, skip, allp):
    og = oracle.OracleGraph(seqs, k, 1)
    pg = pyref.Graph(seqs, k, 1)
    assert og.num_kmers == len(pg.kmers)
    r1, r2 = oracle.OracleRng(7), pyref.GlibcRand(7)
    filled = 0
    for g in gaps:
        o = oracle.fill_gap(og, r1, g["left"], g["right"], g["gap_len"], e, g["lmf"], g["rmf"], skip, allp)
        pi = pyref.Info()
        c2, lf2, rf2, fill2, sub2 = pyref.fill_gap(pg, r2, g["left"], g["right"], g["gap_len"], e, g["lmf"],
                                                   g["rmf"], skip, allp, True, pi)
        assert o.count == c2
        assert o.info.phaseC_count == pi.phaseC_count
        assert o.lengths == pi.lengths
        assert o.info.draws == pi.draws
        assert o.info.q7 == pi.q7
        assert (o.left_fuz, o.right_fuz) == (lf2, rf2)
        assert [int(x) for x in o.info.ctr] == [pi.ctr[x] for x in ("xA", "sA", "xB", "sB", "xD", "sD")]
        if o.phase_d:
            assert o.fill == pyref.fill_string(fill2, g["lmf"] - lf2)
            if sub2 is not None:
                assert o.substats == [sub2[x] for x in ("vertices", "edges", "nontrivial", "size_nontrivial",
                                                        "vertices_final", "edges_final")]
        filled += o.count > 0
    og.free()
    return filled


@pytest.mark.parametrize("seed", range(10))
def test_cpp_oracle_equals_python_restatement(oracle, seed):
    k = [5, 7, 9, 11, 13][seed % 5]
    seqs = cases.toy_genome(seed, 700, k, repeats=seed % 4, tandem=seed % 3, inverted=int(seed % 5 == 0),
                            snp_every=(0 if seed % 2 else 83))
    e = [0, 4, 9, 20, 31][seed % 5] + k
    gaps = cases.cut_gaps(seed, seqs[0], k, fuz=seed % 5 + 1, ngaps=14, min_len=1, max_len=50, d_err=e)
    total = 0
    for skip, allp in ((False, True), (False, False), (True, True)):
        total += _compare_with_pyref(oracle, seqs, k, gaps, e, skip, allp)
    assert total > 0


def test_wide_kmers_and_even_k(oracle):
    for k in (12, 32, 33, 47):
        seqs = cases.toy_genome(k, 1500, k, repeats=1, snp_every=160)
        gaps = cases.cut_gaps(k, seqs[0], k, fuz=4, ngaps=5, min_len=5, max_len=60, d_err=k + 10)
        assert _compare_with_pyref(oracle, seqs, k, gaps, k + 10, False, True) > 0


def test_phase_c_counts_equal_brute_force(oracle):
    """Walk counts from the DP equal plain level-by-level enumeration without pruning
    (even g and e so that Q1 cannot cut the longest walks)."""
    k = 7
    seqs = cases.toy_genome(21, 500, k, repeats=2, snp_every=41)
    pg = pyref.Graph(seqs, k, 1)
    og = oracle.OracleGraph(seqs, k, 1)
    rng = oracle.OracleRng(1)
    g = seqs[0]
    checked = 0
    for pos in range(40, 400, 23):
        true_len, fuz, e = 12, 2, 12
        left, right = g[pos - k - fuz:pos], g[pos + true_len:pos + true_len + k + fuz]
        claimed = true_len + k - 1  # odd offset so that err != 0 at the hit
        claimed += claimed % 2  # even
        o = oracle.fill_gap(og, rng, left, right, claimed, e, fuz, fuz, False, True)
        if o.info.q7 or not o.lengths:
            continue
        total = 0
        for L in o.lengths:
            total += pyref.brute_force_walks(pg, left, right, claimed, fuz, fuz, o.info.reached_fuz, L)
        assert o.info.phaseC_count == min(total, pyref.MAX_PATHS)
        checked += 1
    assert checked >= 5


# ---- quirk fixtures -------------------------------------------------

def _linear(seed, n):
    return cases.random_dna(cases.SplitMix(seed), n)


def test_q4_fuz0_never_fills_in_all_paths_mode(oracle):
    k, e, gl = 9, 20, 15
    g = _linear(3, 300)
    og = oracle.OracleGraph([g], k, 1)
    left, right = g[100 - k:100], g[100 + gl:100 + gl + k]
    args = (left, right, gl + k, e, 0, 0)
    o = oracle.fill_gap(og, oracle.OracleRng(1), *args, False, True)
    assert o.count == 0 and o.info.phaseC_count == 1 and o.info.draws > 0  # draws are still consumed
    o = oracle.fill_gap(og, oracle.OracleRng(1), *args, False, False)  # -best-only
    assert o.count == 1 and o.fill == g[100:100 + gl + k]
    o = oracle.fill_gap(og, oracle.OracleRng(1), *args, True, True)  # -all-upper
    assert o.count == 1 and o.fill == g[100:100 + gl + k]


def test_path_length_window_is_centred_on_g_not_g_plus_k(oracle):
    """Gap2Seq.cpp:1125-1126: lengths g+lmf+j+-err are tested while a true gap of g
    bases needs g+k+lmf steps, so an exact gap needs dist-error >= k."""
    k, gl, fuz = 9, 20, 2
    g = _linear(5, 300)
    og = oracle.OracleGraph([g], k, 1)
    left, right = g[100 - k - fuz:100], g[100 + gl:100 + gl + k + fuz]
    assert oracle.fill_gap(og, oracle.OracleRng(1), left, right, gl, k - 1, fuz, fuz).count == 0
    o = oracle.fill_gap(og, oracle.OracleRng(1), left, right, gl, k, fuz, fuz)
    assert o.count == 1 and o.fill == g[100:100 + gl + k] and o.fill.isupper()


def test_q1_odd_gap_and_odd_error_prune_one_level_early(oracle):
    """g and e both odd: right BFS depth uses ceil((g+e)/2) but the pruning threshold
    g/2+e/2 (Gap2Seq.cpp:1050, two int divisions) is one smaller, so a path of
    exactly g+e steps is lost; with g, e even the same path is found."""
    k, fuz = 9, 0
    g = _linear(9, 400)
    og = oracle.OracleGraph([g], k, 1)
    pg = pyref.Graph([g], k, 1)
    true_len = 31
    left, right = g[100 - k:100], g[100 + true_len:100 + true_len + k]
    need = true_len + k  # 40 steps from the left k-mer to the right k-mer
    odd = oracle.fill_gap(og, oracle.OracleRng(1), left, right, need - 5, 5, fuz, fuz, False, False)
    even = oracle.fill_gap(og, oracle.OracleRng(1), left, right, need - 4, 4, fuz, fuz, False, False)
    assert odd.count == 0 and even.count == 1
    assert pyref.fill_gap(pg, pyref.GlibcRand(1), left, right, need - 5, 5, fuz, fuz, False, False)[0] == 0
    assert pyref.fill_gap(pg, pyref.GlibcRand(1), left, right, need - 4, 4, fuz, fuz, False, False)[0] == 1


def test_bubble_chain_counts_and_saturation(oracle):
    """b bubbles -> 2^b paths; 2^31 paths saturate at MAX_PATHS (Gap2Seq.cpp:38)."""
    k = 11
    rng = cases.SplitMix(77)
    # a de Bruijn-safe chain: random sequence, haplotype 2 differs every 2k+1 bases
    for nb, expect in ((3, 8), (31, pyref.MAX_PATHS)):
        n = (2 * k + 3) * nb + 4 * k
        # retry seeds until both oracle implementations see a clean chain (no accidental repeats)
        for attempt in range(200):
            g = cases.random_dna(rng, n)
            h = list(g)
            for b in range(nb):
                p = 2 * k + b * (2 * k + 3)
                h[p] = "ACGT"[("ACGT".index(h[p]) + 1) % 4]
            h = "".join(h)
            pg = pyref.Graph([g, h], k, 1)
            if len(pg.kmers) != 2 * (n - k + 1) - (n - k + 1 - nb * k):
                continue
            og = oracle.OracleGraph([g, h], k, 1)
            gl = n - 2 * k
            # e = k + 1 keeps (g, e) from being both odd, which would lose the path to Q1
            o = oracle.fill_gap(og, oracle.OracleRng(1), g[:k], g[n - k:], gl, k + 1, 0, 0, False, False)
            og.free()
            if o.info.q7:
                continue
            assert o.count == expect
            break
        else:
            pytest.skip("no clean bubble chain found")


def test_golden_fill_gap_vectors(oracle):
    sets = json.load(open(os.path.join(HERE, "golden", "fill_gap_cases.json")))
    n = 0
    for s in sets:
        og = oracle.OracleGraph(s["seqs"], s["k"], s["solid"])
        rng = oracle.OracleRng(s["randseed"])
        for g, exp in zip(s["gaps"], s["expected"]):
            o = oracle.fill_gap(og, rng, g["left"], g["right"], g["gap_len"], s["d_err"], g["lmf"], g["rmf"],
                                s["skip_confident"], s["all_paths"])
            assert (o.count, o.left_fuz, o.right_fuz, o.fill, o.info.draws, o.info.q7) == \
                (exp["count"], exp["left_fuz"], exp["right_fuz"], exp["fill"], exp["draws"], exp["q7"]), s["name"]
            assert o.lengths == exp["lengths"] and o.info.phaseC_count == exp["phaseC_count"]
            if exp["substats"] is not None:
                assert o.substats == exp["substats"]
            assert [int(x) for x in o.info.ctr] == exp["ctr"]
            n += 1
        og.free()
    assert n >= 100


def test_golden_scaffold_mode(oracle):
    sc = json.load(open(os.path.join(HERE, "golden", "scaffold_cases.json")))
    og = oracle.OracleGraph(sc["seqs"], sc["k"], sc["solid"])
    modes = dict(default={}, best_only=dict(all_paths=False), all_upper=dict(skip_confident=True),
                 unique=dict(unique_paths=True))
    for mode, kw in modes.items():
        fa, lg, sm = oracle.execute_scaffolds(og, sc["scaffolds"], sc["k"], solid=sc["solid"], d_err=sc["d_err"],
                                              max_fuz=sc["max_fuz"], randseed=sc["randseed"], **kw)
        assert fa == sc["expected"][mode]["fasta"]
        assert lg == sc["expected"][mode]["log"]
    # Q8: the second fill of a record discards what precedes the previous gap end
    recs = dict(ln.split("\n")[:2] for ln in sc["expected"]["default"]["fasta"].split(">")[1:])
    src = dict(ln.split("\n")[:2] for ln in sc["scaffolds"].split(">")[1:])
    assert len(recs["two_gaps"]) < len(src["two_gaps"]) - 40
    # Q9 / D2: gaps without complete flanks keep their N's
    assert recs["start_gap"] == src["start_gap"] and recs["short_right"] == src["short_right"]
    assert recs["no_gap"] == src["no_gap"]
    og.free()


def test_oracle_cli_accepts_wrapper_argv(oracle, tmp_path):
    """The argv the reference wrapper builds (Gap2Seq.py:230-241)."""
    import subprocess
    k = 9
    seqs = cases.toy_genome(5, 800, k)
    reads = tmp_path / "reads.fa"
    reads.write_text(">r\n%s\n" % seqs[0])
    scaf = tmp_path / "scaf.fa"
    scaf.write_text(">s1\n%s\n" % cases.scaffold_record(seqs[0], k, 4, [(200, 20, 20 + k)]))
    out = tmp_path / "out.fa"
    cli = os.path.join(os.path.dirname(oracle.ORACLE_SO), "g2s_oracle_cli")
    res = subprocess.run([cli, "-k", str(k), "-fuz", "4", "-solid", "1", "-nb-cores", "1", "-dist-error", "20",
                          "-max-mem", "1", "-randseed", "1", "-reads", str(reads), "-filled", str(out),
                          "-scaffolds", str(scaf)], capture_output=True, text=True)
    assert res.returncode == 0
    assert "Filled 1 gaps out of 1" in res.stdout
    assert out.read_text().splitlines()[1].upper() == seqs[0][200 - k - 4 - 5:200 + 20 + k + 4 + 5].upper()


def test_committed_oracle_units_are_the_oracles(oracle, product):
    """profiles/oracle_units.json prices bench.py's rooflines (X and S are "counted by the CPU
    oracle"): the table's entry for the headline workload (BASELINE config 2) and for config 3's share of one
    GPU at 8 GPUs must be what the oracle counts on those lists now, and an unknown workload has no entry."""
    import bench
    for ngaps in (500, 1250):
        key = bench.units_key(3000000, 3, 31, ngaps, 200, 1000, 10, 500)
        u = bench.load_oracle_units()[key]
        reads = product.G2S.synth_genome(3000000, 3, bench.GENOME_SEED)
        seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
        gaps = bench.parse_gaps(product.G2S.synth_gaps(reads, 31, 10, ngaps, 200, 1000, bench.GAP_SEED), 10)
        og = oracle.OracleGraph(seqs, 31, 1)
        _, filled, c = oracle.time_fill_batch(og, gaps, 500, 4)
        og.free()
        assert [u["xA"], u["sA"], u["xB"], u["sB"], u["xD"], u["sD"]] == c and u["filled"] == filled == ngaps
        x, s, by = bench.oracle_units(key)
        assert (x, s) == (c[0] + c[2] + c[4], c[1] + c[3] + c[5]) and by.startswith("oracle")
    x, s, by = bench.oracle_units(bench.units_key(3000000, 3, 31, 499, 200, 1000, 10, 500))
    assert x is None and s is None and by.startswith("unavailable")
    assert bench.oracle_units("no such workload", lambda: [1, 2, 3, 4, 5, 6]) == (9, 12, "oracle (counted in this run over the whole list)")
