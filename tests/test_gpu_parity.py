"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP fill path, called
through the C ABI, against the CPU oracle on the same seeded inputs, against the
committed golden vectors, and — at BASELINE.json's full sizes — through properties
that need no oracle (cut -> fill -> original sequence round trip on a repeat-free
genome).  Bit-exact: counts, fuz values, rand() draw counts, fill strings including
the upper/lower-case safe/unsafe classification, subgraph statistics, FASTA and log
text.  Gaps flagged Q7 (both strands of a k-mer in one border; the reference's
outcome depends on libstdc++ hash-set order) are excluded from the
bit-exact claim but must be flagged by the GPU path whenever the oracle flags them.
"""
import json
import os
import subprocess
import sys

import pytest

import cases

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _gaps(product, gl):
    return [product.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gl]


def _parse_scaffolds(text, fuz=10):
    lines = text.splitlines()
    out = []
    for j in range(0, len(lines), 2):
        s = lines[j + 1]
        a = s.index("N")
        b = len(s) - s[::-1].index("N")
        out.append(dict(left=s[:a], right=s[b:], gap_len=b - a, lmf=fuz, rmf=fuz))
    return out


def _assert_gap_equal(product, r, o, skip, what=""):
    """One gap of the product against the oracle's fill_gap: every observable field."""
    assert r.count == o.count, "%s: count %r, the oracle's %r (flags %#x)" % (what, r.count, o.count, r.flags)
    assert r.phaseC_count == o.info.phaseC_count and r.lengths == o.lengths, "%s: phase C count %r lengths %r, the oracle's %r %r (flags %#x)" % (
        what, r.phaseC_count, r.lengths, o.info.phaseC_count, o.lengths, r.flags)
    assert r.draws == o.info.draws, "%s: %r draws, the oracle's %r (flags %#x)" % (what, r.draws, o.info.draws, r.flags)
    if o.phase_d:
        assert (r.left_fuz, r.right_fuz) == (o.left_fuz, o.right_fuz), "%s: fuz %r, the oracle's %r" % (what, (r.left_fuz, r.right_fuz), (o.left_fuz, o.right_fuz))
        if r.fill != o.fill:  # sequence AND case (safe/unsafe bases)
            at = next((x for x in range(min(len(r.fill), len(o.fill))) if r.fill[x] != o.fill[x]), min(len(r.fill), len(o.fill)))
            raise AssertionError("%s: fill text differs at %d of %d / %d (%r against the oracle's %r; case only: %r; flags %#x)" % (
                what, at, len(r.fill), len(o.fill), r.fill[at:at + 12], o.fill[at:at + 12], r.fill.upper() == o.fill.upper(), r.flags))
        if not skip:
            assert r.substats == o.substats, "%s: subgraph statistics %r, the oracle's %r" % (what, r.substats, o.substats)


def _compare_with_oracle_in_parallel(product, oracle, og, gaps, res, e, seed, skip=False, allp=True, threads=0):
    """Every gap of a finished list against the oracle, on all host cores: gap i's oracle run starts its own rand()
    stream at the product's cumulative draw count in front of gap i (what the sequential harness re-synchronises to
    anyway), so the gaps are independent.  Returns (compared, oracle_q7); the oracle's calls release the GIL."""
    import concurrent.futures
    import threading
    offs, used = [], 0
    for r in res:
        offs.append(used)
        used += r.draws
    lock = threading.Lock()
    tally = dict(compared=0, q7=0)

    def one(i):
        g, r = gaps[i], res[i]
        rng = oracle.OracleRng(seed, offs[i])
        try:
            o = oracle.fill_gap(og, rng, g["left"], g["right"], g["gap_len"], e, g["lmf"], g["rmf"], skip, allp)
        finally:
            rng.free()
        if o.info.q7:
            assert r.flags & product.G2S_GAP_Q7, "oracle saw a Q7 collision the GPU path did not flag (gap %d)" % i
            with lock:
                tally["q7"] += 1
            return
        if r.count == -1:
            return
        _assert_gap_equal(product, r, o, skip, "gap %d" % i)
        with lock:
            tally["compared"] += 1

    nthreads = threads or min(64, os.cpu_count() or 1)
    with concurrent.futures.ThreadPoolExecutor(nthreads) as ex:
        for f in [ex.submit(one, i) for i in range(len(gaps))]:
            f.result()  # (raises the first failed comparison)
    return tally["compared"], tally["q7"]


def _check_batch(product, oracle, seqs, k, gaps, e, skip=False, allp=True, seed=5, max_mem=20 << 30, run_product=None, host_threads=0):
    """The product's batch against the oracle gap by gap.  Only gaps on which the ORACLE
    sees a Q7 collision (both strands of a k-mer in one border: the reference's outcome then
    depends on libstdc++'s hash-set order) are outside the bit-exact claim; they must carry
    the product's Q7 flag.  A gap the product flags conservatively while the oracle sees no
    collision is compared like any other.  The oracle's rand() stream is re-synchronised
    with the product's after every gap whose draw counts differ (possible for oracle-Q7 gaps
    only), so every later gap is still compared.  Returns (compared, filled, timing, xB, sB)
    with compared == len(gaps) - oracle_q7 asserted.  run_product(sess, gap structs) -> (results, timing) replaces the
    one g2s_fill_batch call (tools/fuzz_parity.py: the same gaps as several lists in flight)."""
    og = oracle.OracleGraph(seqs, k, 1)
    pg = product.Graph.from_seqs(seqs, k, 1)
    sess = product.Session(pg, 0, d_err=e, skip_confident=skip, all_paths=allp, randseed=seed, max_mem=max_mem, host_threads=host_threads)
    res, tm = run_product(sess, _gaps(product, gaps)) if run_product else sess.fill_batch(_gaps(product, gaps), True)
    assert tm.watchdog_gaps == 0  # (a probe loop of the large variant ran past its bound: a defect)
    rng = oracle.OracleRng(seed)
    compared = filled = oracle_q7 = 0
    xb = sb = 0
    used = 0  # draws the product has consumed so far = where its next gap starts in the stream
    try:
        for i, (g, r) in enumerate(zip(gaps, res)):
            o = oracle.fill_gap(og, rng, g["left"], g["right"], g["gap_len"], e, g["lmf"], g["rmf"], skip, allp)
            used += r.draws
            if o.info.q7:
                assert r.flags & product.G2S_GAP_Q7, "oracle saw a Q7 collision the GPU path did not flag (gap %d)" % i
                oracle_q7 += 1
                if o.info.draws != r.draws:  # the two streams have parted: put the oracle's where the product's is
                    rng.free()
                    rng = oracle.OracleRng(seed, used)
                continue
            if r.count == -1:  # the -max-mem verdict is a device-budget analogue (D3), not compared
                if o.info.draws != r.draws:
                    rng.free()
                    rng = oracle.OracleRng(seed, used)
                continue
            _assert_gap_equal(product, r, o, skip, "gap %d" % i)
            xb += o.info.ctr[2]
            sb += o.info.ctr[3]
            compared += 1
            filled += o.count > 0
        mem = sum(1 for r in res if r.count == -1)
        assert compared == len(gaps) - oracle_q7 - mem
    finally:
        rng.free()
        sess.destroy()
        pg.free()
        og.free()
    return compared, filled, tm, xb, sb


@pytest.fixture(params=["seg", "res", "segx", "lds", "hbm"])
def tier(request, monkeypatch):
    """All kernel tiers: the segment tier (default: the search over unitig segments) with phase D on the host
    ("seg") and with the whole list finished on the device ("res": resident mode, d3_device.hip), its large
    variant (what a gap takes when it outgrows the LDS-resident capacities), the LDS tier (level by
    level, round 1's kernel: the fallback behind both) and the general tier with per-gap tables in
    HBM (the last resort)."""
    monkeypatch.delenv("G2S_NO_LDS_TIER", raising=False)
    monkeypatch.delenv("G2S_NO_SEG_TIER", raising=False)
    monkeypatch.delenv("G2S_FORCE_SEGX", raising=False)
    monkeypatch.setenv("G2S_RESIDENT", "1" if request.param == "res" else "0")
    if request.param == "segx":
        monkeypatch.setenv("G2S_FORCE_SEGX", "1")
    elif request.param == "hbm":
        monkeypatch.setenv("G2S_NO_LDS_TIER", "1")
    elif request.param == "lds":
        monkeypatch.setenv("G2S_NO_SEG_TIER", "1")
    return request.param


@pytest.mark.parametrize("seed", range(12))
def test_toy_graphs_all_modes(product, oracle, seed, tier):
    k = [5, 7, 9, 11, 13, 15][seed % 6]
    seqs = cases.toy_genome(seed, 900, k, repeats=seed % 4, tandem=seed % 3, inverted=int(seed % 5 == 0),
                            snp_every=(0 if seed % 2 else 83))
    e = [0, 4, 9, 20, 31][seed % 5] + k
    gaps = cases.cut_gaps(seed, seqs[0], k, fuz=seed % 5 + 1, ngaps=40, min_len=1, max_len=60, d_err=e)
    total = 0
    for skip, allp in ((False, True), (False, False), (True, True)):
        c, f, _, _, _ = _check_batch(product, oracle, seqs, k, gaps, e, skip, allp)  # asserts c == gaps - oracle Q7
        total += c
    # (k = 5, 7: nearly every gap of a 900 bp genome meets both strands of some k-mer: _check_batch has
    # asserted that each of those carries the product's Q7 flag and that every other gap is equal;
    # test_small_k_without_strand_collisions is where gaps at these k are compared bit for bit)
    assert total >= (90 if k >= 11 else 30 if k >= 9 else 0)


@pytest.mark.parametrize("k,length", [(5, 60), (5, 100), (7, 200)])
def test_small_k_without_strand_collisions(product, oracle, k, length, tier):
    """k = 5 and 7 on genomes over {A, C} (cases.one_strand_genome): no k-mer meets its reverse strand, so the
    oracle sees no Q7 case and EVERY gap is compared bit for bit — dense graphs, closures made of cycles,
    counts that saturate (on the toy graphs above nearly every gap at these k is a Q7 case and only its flag
    is checked)."""
    for seed in range(4):
        seqs = [cases.one_strand_genome(seed, length)]
        e = [0, 4, 9, 20, 31][seed % 5] + k
        gaps = cases.cut_gaps(seed, seqs[0], k, fuz=seed % 3 + 1, ngaps=40, min_len=1, max_len=length // 4, d_err=e)
        filled = 0
        for skip, allp in ((False, True), (False, False), (True, True)):
            c, f, _, _, _ = _check_batch(product, oracle, seqs, k, gaps, e, skip, allp)
            assert c == len(gaps)
            filled += f
        assert filled >= 100


def test_golden_vectors_on_gpu(product):
    """The committed vectors.  Expected-Q7 gaps only have to be flagged; when such a gap's draw
    count differs, the rest of the set is run again from the expected stream position
    (srand + skip), so that every other gap of every set is compared."""
    sets = json.load(open(os.path.join(HERE, "golden", "fill_gap_cases.json")))
    n = nq7 = 0
    for s in sets:
        pg = product.Graph.from_seqs(s["seqs"], s["k"], s["solid"])
        sess = product.Session(pg, 0, d_err=s["d_err"], skip_confident=s["skip_confident"], all_paths=s["all_paths"],
                               randseed=s["randseed"])
        start = 0
        exp_used = 0  # draws the expected results consume before gap `start`
        while start < len(s["gaps"]):
            sess.srand(s["randseed"], exp_used)
            res = sess.fill_batch(_gaps(product, s["gaps"][start:]))
            nxt = len(s["gaps"])
            for i, (r, exp) in enumerate(zip(res, s["expected"][start:])):
                exp_used += exp["draws"]
                if exp["q7"]:
                    assert r.flags & product.G2S_GAP_Q7
                    nq7 += 1
                    if exp["draws"] != r.draws:
                        nxt = start + i + 1
                        break
                    continue
                assert (r.count, r.draws, r.phaseC_count, r.lengths) == (exp["count"], exp["draws"], exp["phaseC_count"],
                                                                        exp["lengths"]), s["name"]
                if exp["fill"]:
                    assert (r.left_fuz, r.right_fuz, r.fill) == (exp["left_fuz"], exp["right_fuz"], exp["fill"]), s["name"]
                if exp["substats"] is not None and exp["phaseC_count"] > 0:
                    assert r.substats == exp["substats"], s["name"]
                n += 1
            start = nxt
        sess.destroy()
        pg.free()
    assert n + nq7 == sum(len(s["gaps"]) for s in sets) and n >= 90


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_k31_default_parameters(product, oracle, variant, tier):
    """k=31, -fuz 10, -dist-error 500 on a 200 kbp genome: V0 plain, V1 repeats,
    V2 bubbles, V3 both; device work counters equal the oracle's."""
    reads = product.G2S.synth_genome(200000, variant, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 80, 50, 600, 20240103))
    c, f, tm, xb, sb = _check_batch(product, oracle, seqs, 31, gaps, 500)
    assert c == 80 and f >= 70
    assert (tm.xB, tm.sB) == (xb, sb)


def test_wide_kmers_k63_and_even_k(product, oracle):
    """128-bit k-mer encoding (BASELINE config 4 uses k=63) and an even k (palindromes)."""
    for k, variant in ((63, 3), (33, 1), (32, 2), (12, 0)):
        reads = product.G2S.synth_genome(60000, variant, 7)
        seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
        fuz = 10 if k > 12 else 3
        gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, k, fuz, 30, 40, 300, 11), fuz)
        c, f, _, _, _ = _check_batch(product, oracle, seqs, k, gaps, 200 if k > 12 else 40)
        assert c >= 25 and f >= 15


def test_deep_dp_dist_error_2000(product, oracle, tier):
    """BASELINE config 5 shape (wide rows): -dist-error 2000, gaps of 2-5 kbp, few gaps."""
    reads = product.G2S.synth_genome(300000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 16, 2000, 5000, 5))
    c, f, _, _, _ = _check_batch(product, oracle, seqs, 31, gaps, 2000)
    assert c == 16 and f >= 12


@pytest.mark.parametrize("lmf,rmf", [(31, 31), (40, 40), (3, 25), (25, 0)])
def test_fuz_extremes(product, oracle, lmf, rmf):
    """Flank fuzz at and beyond what the LDS tier holds (32 target k-mers): -fuz 31 stays in the
    LDS tier, -fuz 40 takes the HBM tier; asymmetric values as `execute` produces them near
    record ends (:349,360)."""
    reads = product.G2S.synth_genome(120000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    g = seqs[0]
    k = 31
    gaps = []
    for i in range(40):
        p = 500 + i * 2500
        ln = 60 + 13 * i
        gaps.append(dict(left=g[p - k - lmf:p], right=g[p + ln:p + ln + k + rmf], gap_len=ln, lmf=lmf, rmf=rmf))
    c, f, _, _, _ = _check_batch(product, oracle, seqs, k, gaps, 200)
    assert c >= 35 and (f >= 30 or rmf == 0)


def test_unfillable_and_ragged_inputs(product, oracle):
    k = 15
    seqs = cases.toy_genome(4, 3000, k, repeats=2)
    g = seqs[0]
    other = cases.random_dna(cases.SplitMix(99), 200)
    gaps = [
        dict(left=other[:k + 3], right=g[500:500 + k + 3], gap_len=30, lmf=3, rmf=3),   # left flank not in graph
        dict(left=g[100:100 + k + 3], right=other[50:50 + k + 3], gap_len=30, lmf=3, rmf=3),  # right not in graph
        dict(left=g[100:100 + k], right=g[140:140 + k], gap_len=40, lmf=0, rmf=0),      # fuz 0 (Q4)
        dict(left=g[100:100 + k + 2], right=g[2000:2000 + k + 2], gap_len=50, lmf=2, rmf=2),  # too far apart
        dict(left=g[200:200 + k + 3], right=g[200 + k + 3 + 1:200 + 2 * k + 7], gap_len=1, lmf=3, rmf=3),  # 1 base
    ]
    c, f, _, _, _ = _check_batch(product, oracle, seqs, k, gaps, k + 10)
    assert c >= 4
    # empty batch and a flank shorter than k + fuz (the reference would throw; D2)
    pg = product.Graph.from_seqs(seqs, k, 1)
    sess = product.Session(pg, 0, d_err=30)
    assert sess.fill_batch([]) == []
    r = sess.fill_batch([product.Gap(g[:k - 1], g[100:100 + k], 10, 0, 0)])[0]
    assert r.count == 0 and (r.flags & product.G2S_GAP_BAD_FLANK)
    sess.destroy()
    pg.free()


def test_table_overflow_retry_and_memory_verdict(product, oracle):
    """Repeat-rich graph: small first-tier tables overflow and are retried (same
    results); with a tiny -max-mem the gap ends as -1 'Memory limit exceeded'."""
    reads = product.G2S.synth_genome(200000, 1, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 120, 200, 1000, 3))
    c, f, tm, _, _ = _check_batch(product, oracle, seqs, 31, gaps, 500)
    assert c == 120
    os.environ["G2S_NO_LDS_TIER"] = "1"
    try:
        c, f, tm, _, _ = _check_batch(product, oracle, seqs, 31, gaps, 500)
    finally:
        del os.environ["G2S_NO_LDS_TIER"]
    assert c == 120
    pg = product.Graph.from_seqs(seqs, 31, 1)
    sess = product.Session(pg, 0, d_err=500, max_mem=1 << 16)  # 1024 states per gap
    res = sess.fill_batch(_gaps(product, gaps))
    assert any(r.count == -1 and (r.flags & product.G2S_GAP_MEM_EXCEEDED) for r in res)
    assert all(r.count == -1 or r.count >= 0 for r in res)
    sess.destroy()
    pg.free()


def test_scaffold_mode_fasta_and_log_identical(product, oracle):
    sc = json.load(open(os.path.join(HERE, "golden", "scaffold_cases.json")))
    pg = product.Graph.from_seqs(sc["seqs"], sc["k"], sc["solid"])
    modes = dict(default={}, best_only=dict(all_paths=False), all_upper=dict(skip_confident=True),
                 unique=dict(unique_paths=True))
    for mode, kw in modes.items():
        sess = product.Session(pg, 0, d_err=sc["d_err"], randseed=sc["randseed"], **kw)
        fa, lg, gaps, filled = sess.execute_scaffolds(sc["scaffolds"], sc["k"], solid=sc["solid"],
                                                      max_fuz=sc["max_fuz"])
        assert fa == sc["expected"][mode]["fasta"], mode
        assert lg == sc["expected"][mode]["log"], mode
        sess.destroy()
    pg.free()


def test_scaffold_mode_many_records_vs_oracle(product, oracle, tier):
    """Multi-gap scaffolds with close gaps (right_fuz coupling), k < fuz (left_max_fuz
    coupling -> batch barrier), lower-case n runs."""
    for k, fuz, e in ((11, 4, 30), (5, 8, 20), (21, 10, 60)):
        seqs = cases.toy_genome(k, 6000, k, repeats=3, tandem=1, snp_every=211)
        g = seqs[0]
        rng = cases.SplitMix(k)
        recs = []
        for r in range(25):
            start = rng.randint(0, 4000)
            pos = start + k + fuz + 5
            triples = []
            for _ in range(rng.randint(1, 4)):
                ln = rng.randint(1, 40)
                triples.append((pos, ln, max(1, ln + k + rng.randint(-3, 3))))
                pos += ln + rng.choice([k + fuz - 1, k + fuz, k + fuz + 1, k + 2 * fuz, 3 * k + 2 * fuz + 7])
            recs.append(("rec%d extra words" % r, cases.scaffold_record(g, k, fuz, triples)))
        text = "".join(">%s\n%s\n" % x for x in recs)
        og = oracle.OracleGraph(seqs, k, 1)
        pg = product.Graph.from_seqs(seqs, k, 1)
        for kw in ({}, dict(unique_paths=True), dict(all_paths=False)):
            ofa, olog, sm = oracle.execute_scaffolds(og, text, k, solid=1, d_err=e, max_fuz=fuz, randseed=9, **kw)
            if sm.q7_gaps:
                continue
            sess = product.Session(pg, 0, d_err=e, randseed=9, **kw)
            fa, lg, gaps, filled = sess.execute_scaffolds(text, k, solid=1, max_fuz=fuz)
            sess.destroy()
            assert fa == ofa
            assert lg == olog
            assert (gaps, filled) == (sm.gaps, sm.filled)
        og.free()
        pg.free()


def test_single_gap_mode(product, oracle):
    k = 13
    seqs = cases.toy_genome(8, 2000, k, snp_every=97)
    g = seqs[0]
    og = oracle.OracleGraph(seqs, k, 1)
    pg = product.Graph.from_seqs(seqs, k, 1)
    for left, right, length in ((g[300:330], g[360:395], 30 + k), (g[300:300 + k], g[340:340 + k], 27 + k),
                                (g[300:310], g[340:380], 30)):
        ofa, olog = oracle.execute_single(og, left, right, length, k, solid=1, d_err=40, max_fuz=6, randseed=2)
        sess = product.Session(pg, 0, d_err=40, randseed=2)
        fa, lg = sess.execute_single(left, right, length, k, solid=1, max_fuz=6)
        sess.destroy()
        assert (fa, lg) == (ofa, olog)
    og.free()
    pg.free()


def test_bench_workload_c2_vs_oracle(product, oracle):
    """The exact bench.py workload (BASELINE config 2: 3 Mbp genome V3 = repeats + bubbles,
    k=31, 500 gaps of 200-1000 bp, -fuz 10 -dist-error 500): every gap bit-exact against
    the oracle, FASTA and log of the scaffold run identical, device work counters equal."""
    reads = product.G2S.synth_genome(3000000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    scaf = product.G2S.synth_gaps(reads, 31, 10, 500, 200, 1000, 20240103)
    gaps = _parse_scaffolds(scaf)
    c, f, tm, xb, sb = _check_batch(product, oracle, seqs, 31, gaps, 500, seed=1)
    assert (c, f) == (500, 500)
    assert (tm.xB, tm.sB) == (xb, sb)
    assert tm.seg_tier_gaps == 500 and tm.lds_tier_gaps == 0 and tm.retried_gaps == 0
    og = oracle.OracleGraph(seqs, 31, 1)
    pg = product.Graph.from_seqs(seqs, 31, 1)
    ofa, olog, sm = oracle.execute_scaffolds(og, scaf, 31, solid=1, d_err=500, max_fuz=10, randseed=1)
    sess = product.Session(pg, 0, d_err=500, randseed=1)
    fa, lg, ngaps, nfilled = sess.execute_scaffolds(scaf, 31, solid=1, max_fuz=10)
    sess.destroy()
    assert fa == ofa and lg == olog and (ngaps, nfilled) == (500, 500)
    assert any(ch.islower() for ch in fa.replace(">", "")), "expected some unsafe (lower-case) bases in V3"
    og.free()
    pg.free()


def test_log_keeps_the_references_max_mem_line_whatever_budget_applies(product, oracle):
    """`Max mem:` is -max-mem / execution units in the reference (Gap2Seq.cpp:302-303).  The session applies its own
    per-gap budget (g2s_params.max_mem: the whole -max-mem when -nb-cores is omitted on the command line); the log
    line keeps the reference's formula either way."""
    reads = product.G2S.synth_genome(60000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    scaf = product.G2S.synth_gaps(reads, 31, 10, 20, 100, 400, 20240103)
    og = oracle.OracleGraph(seqs, 31, 1)
    pg = product.Graph.from_seqs(seqs, 31, 1)
    try:
        for cores in (1, 7, 64):
            ofa, olog, sm = oracle.execute_scaffolds(og, scaf, 31, solid=1, d_err=500, max_fuz=10, randseed=1, nb_cores=cores)
            sess = product.Session(pg, 0, d_err=500, randseed=1)  # (budget: the whole 20 GB per gap, not divided)
            fa, lg, ngaps, nfilled = sess.execute_scaffolds(scaf, 31, solid=1, max_fuz=10, nb_cores=cores)
            sess.destroy()
            assert "Max mem: %d\n" % ((20 << 30) // cores) in lg
            assert fa == ofa and lg == olog
    finally:
        og.free()
        pg.free()


@pytest.mark.parametrize("which", ["res", "seg", "lds"])
def test_c3_gap_list_on_the_branching_genome_vs_oracle(product, oracle, which, monkeypatch):
    """BASELINE config 3's list length (10 000 gaps, 3 Mbp, k=31, -fuz 10, -dist-error 500) on
    the V3 genome (repeats + bubbles), gap by gap against the oracle.  Segment tier: every gap
    of this list fits its capacities (one launch, nothing left for the other tiers).  LDS tier
    alone (G2S_NO_SEG_TIER): at this list length the LDS share per gap is at its smallest,
    right sets spill to the launch's pool in HBM and the few repeat-rich gaps whose state log
    outgrows its slice move to the log pool; both must have happened, in a single launch."""
    if which == "lds":
        monkeypatch.setenv("G2S_NO_SEG_TIER", "1")
    monkeypatch.setenv("G2S_RESIDENT", "1" if which == "res" else "0")
    reads = product.G2S.synth_genome(3000000, 3, 20240101)
    scaff = product.G2S.synth_gaps(reads, 31, 10, 10000, 200, 1000, 20240103)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = _parse_scaffolds(scaff)
    assert len(gaps) == 10000
    c, f, tm, _, _ = _check_batch(product, oracle, seqs, 31, gaps, 500, seed=1)
    assert c > 9900 and f > 9900
    if which == "res":  # the whole list on the device: fill kernel + phase D3, nothing left for the host path
        assert tm.resident_launches == 1 and tm.resident_fallbacks == 0 and tm.seg_tier_gaps == 10000
        # (the 70 closures with a k-mer at several depths: analysed by g2s_d2_* on its own stream — d2_device.hip —, none
        # left to the host's threads)
        assert tm.host_finished_gaps <= 5
        assert tm.draw_dependent_gaps > 100
    if which in ("seg", "res"):
        assert tm.seg_tier_gaps == 10000 and tm.seg_launches == 1 and tm.lds_launches == 0
        assert 100000 < tm.seg_segments < 300000  # ~17 segments per gap for ~1000 DP states
    else:
        assert tm.lds_tier_gaps == 10000 and tm.lds_launches == 1
        assert tm.rs_pool_gaps > 0 and tm.log_pool_gaps > 0


def test_multi_rank_bench_path(product, tmp_path):
    """bench.py as the driver starts it for N>1 (torch.distributed.run, one rank per GPU; here 2
    ranks, both "GPUs" being device 0): rank 0's process drives the N sessions of the dispatcher
    on one gap list, asserts the result equal to the one-session result, and prints one JSON line
    with strong scaling; the other rank only joins the barriers."""
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--genome", "300000", "--gaps", "600", "--share-device", "--group", "300"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["equals_one_gpu_result"] is True and out["one_gpu_same_list"]["value"] > 0
    assert out["config"]["gaps"] == 600 and out["config"]["group"] == 300
    # (groups of 300 on ONE device: the sessions share it, so one wave per gap; two on devices of their own)
    assert out["cpu_baseline"] is None and out["roofline"]["kernel"] in ("g2s_fill_seg", "g2s_fill_seg2")
    assert out["resident"]["lists_finished_on_the_device"] == 1 and out["resident"]["team_groups"] == 2
    assert out["roofline"]["launches_per_step"] == 2.0


def test_bench_refuses_more_gpus_than_there_are(product):
    """`bench.py --gpus N` drives N devices itself; with fewer usable devices it must fail, not fall back to one."""
    import sys
    n = product.G2S.device_count()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0",
                          "--genome", "100000", "--gaps", "50", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert "--gpus %d but only %d" % (n + 1, n) in (res.stdout + res.stderr)
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_bench_one_gpu_line(product):
    """bench.py's default shape on a small graph: the timed region is the whole ABI call, the
    JSON line carries roofline and cpu_baseline, and the workload label names what ran."""
    import sys
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--genome",
                          "300000", "--gaps", "200"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["config"]["config"] == "C2-custom" and "custom" in out["config"]["workload"]
    assert "g2s_fill_batch" in out["config"]["timed_region"]
    b = out["breakdown_ms_per_step"]
    assert 0 < b["prepare_flank_lookup_and_upload"] < b["wall_inside_the_abi_call"] <= out["ms_per_step"] * 1.05
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] == 1
    assert out["roofline"]["units_counted_by"].startswith("oracle (counted in this run") and 0 < out["roofline"]["frac"] < 1
    assert out["filled"] >= 190


def _result_tuple(r):
    return (r.count, r.left_fuz, r.right_fuz, r.flags, r.draws, r.fill, tuple(r.substats), r.phaseC_count,
            tuple(r.lengths))


@pytest.mark.parametrize("nsess,group", [(1, 37), (2, 64), (3, 50)])
def test_team_of_sessions_equals_one_session(product, oracle, nsess, group):
    """g2s_team_fill (the dispatcher: sessions pulling groups of gaps from one list, D3 in
    gap order on the lead's rand() stream) must reproduce a single session's batch bit for
    bit, and therefore the oracle: V3 genome (repeats + bubbles), 300 gaps."""
    reads = product.G2S.synth_genome(300000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gl = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 300, 50, 700, 99))
    gaps = _gaps(product, gl)
    pg = product.Graph.from_seqs(seqs, 31, 1)
    og = oracle.OracleGraph(seqs, 31, 1)
    single = product.Session(pg, 0, d_err=500, randseed=17)
    team = [product.Session(pg, 0, d_err=500, randseed=17) for _ in range(nsess)]
    try:
        want = [_result_tuple(r) for r in single.fill_batch(gaps)]
        got, tm = product.team_fill(team, gaps, group_size=group, want_timing=True)
        assert [_result_tuple(r) for r in got] == want
        assert tm.seg_launches >= (300 + group - 1) // group
        # and a second list on the same team continues the lead's rand() stream like the single session
        want2 = [_result_tuple(r) for r in single.fill_batch(gaps[:120])]
        got2 = product.team_fill(team, gaps[:120], group_size=group)
        assert [_result_tuple(r) for r in got2] == want2
        rng = oracle.OracleRng(17)
        n = nq7 = used = 0
        for g, r in zip(gl, got):
            o = oracle.fill_gap(og, rng, g["left"], g["right"], g["gap_len"], 500, g["lmf"], g["rmf"], False, True)
            used += r.draws
            if o.info.q7:
                assert r.flags & product.G2S_GAP_Q7
                nq7 += 1
                if o.info.draws != r.draws:
                    rng = oracle.OracleRng(17, used)
                continue
            _assert_gap_equal(product, r, o, False)
            n += 1
        assert n == 300 - nq7 and n > 250
    finally:
        for t in team:
            t.destroy()
        single.destroy()
        pg.free()
        og.free()


def test_long_list_goes_through_the_group_pipeline(product):
    """g2s_fill_batch cuts lists of more than 16 384 gaps into groups (bounded HBM for the
    state logs) even on a single session; same results as one prepared batch."""
    reads = product.G2S.synth_genome(400000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gl = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 17000, 30, 120, 77))
    gaps = _gaps(product, gl)
    pg = product.Graph.from_seqs(seqs, 31, 1)
    a = product.Session(pg, 0, d_err=100, randseed=21)
    b = product.Session(pg, 0, d_err=100, randseed=21)
    try:
        one = [_result_tuple(r) for r in a.fill_batch(gaps)]
        grouped = [_result_tuple(r) for r in b.fill_batch_onecall(gaps)]
        assert grouped == one
        assert sum(1 for r in one if r[0] > 0) > 16000
    finally:
        a.destroy()
        b.destroy()
        pg.free()


def test_execute_stream_is_the_one_batch_text_for_every_chunk_size(product):
    """g2s_execute_scaffolds_stream hands the log and the records over batch by batch; the concatenation is the
    text of the one-batch run whatever the batch size (the rand() stream runs on across batches, the couplings
    between consecutive gaps never cross a record boundary)."""
    k = 21
    seqs = cases.toy_genome(11, 60000, k, repeats=6, snp_every=400)
    text = _stream_scaffolds(seqs[0], k)
    pg = product.Graph.from_seqs(seqs, k, 1)
    try:
        want = None
        for chunk in (0, 1, 7, 40, 100000):
            sess = product.Session(pg, 0, d_err=100, randseed=4)
            try:
                if chunk == 0:
                    fa, lg, ngaps, nfilled = sess.execute_scaffolds(text, k, solid=1)
                    want = (fa, lg, ngaps, nfilled)
                    assert ngaps >= 60 and nfilled >= 20
                else:
                    fas, lgs, ngaps, nfilled = sess.execute_scaffolds_stream(text, k, chunk, solid=1)
                    assert ("".join(fas), "".join(lgs), ngaps, nfilled) == want, chunk
                    if chunk == 1:
                        assert len(lgs) >= 20  # really handed over in pieces: a batch per record
            finally:
                sess.destroy()
    finally:
        pg.free()


def test_execute_stream_with_lists_in_flight(product):
    """The same with batches long enough to be finished on the device (resident mode) and, behind every cut at a
    record boundary, TWO of them in flight (g2s_fill_begin / g2s_fill_end: the next batch is scanned and begun before
    this one is ended and replayed): the concatenated text is the one-batch text for every batch size."""
    import test_gpu_fullsize
    k = 31
    reads = product.G2S.synth_genome(1300000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    text = test_gpu_fullsize._simulated_scaffolds(seqs[0], k, 10, 5, 840, 1500)
    pg = product.Graph.from_seqs(seqs, k, 1)
    try:
        sess = product.Session(pg, 0, d_err=500, randseed=4)
        want = sess.execute_scaffolds(text, k, solid=1)
        sess.destroy()
        assert want[2] >= 1200 and want[3] >= 700
        for chunk in (260, 300, 700):
            sess = product.Session(pg, 0, d_err=500, randseed=4)
            fas, lgs, ngaps, nfilled = sess.execute_scaffolds_stream(text, k, chunk, solid=1)
            sess.destroy()
            assert ("".join(fas), "".join(lgs), ngaps, nfilled) == want, chunk
            assert len(lgs) >= 2
    finally:
        pg.free()


def _stream_scaffolds(genome, k):
    """multi-gap scaffold records over a toy genome (the generator of the full-size C1 stand-in)"""
    import test_gpu_fullsize
    return test_gpu_fullsize._simulated_scaffolds(genome, k, 10, 77, 40, 1500)


def test_session_team_drives_execute_scaffolds(product, oracle):
    """g2s_session_set_team: execute() on the lead spreads the record list's gaps over the
    helpers; FASTA and log equal the oracle's execute()."""
    k = 21
    seqs = cases.toy_genome(5, 30000, k, repeats=3, snp_every=401)
    g = seqs[0]
    recs = []
    for r in range(60):
        p = 100 + r * 450
        recs.append((">sc%d\n" % r) + cases.scaffold_record(g, k, 10, [(p, 30 + r % 40, 30 + r % 40 + k), (p + 200, 20, 20 + k)]))
    text = "\n".join(recs) + "\n"
    pg = product.Graph.from_seqs(seqs, k, 1)
    og = oracle.OracleGraph(seqs, k, 1)
    lead = product.Session(pg, 0, d_err=100, randseed=4)
    helpers = [product.Session(pg, 0, d_err=100, randseed=4) for _ in range(2)]
    try:
        lead.set_team(helpers, group_size=16)
        fa, lg, ngaps, nfilled = lead.execute_scaffolds(text, k, solid=1)
        ofa, olg, sm = oracle.execute_scaffolds(og, text, k, solid=1, d_err=100, max_fuz=10, randseed=4)
        assert ngaps == 120
        if not sm.q7_gaps:
            assert (ngaps, nfilled) == (sm.gaps, sm.filled)
            assert fa == ofa
            assert lg == olg
        lead.set_team([])
        lead.srand(4)
        fa1, lg1, _, _ = lead.execute_scaffolds(text, k, solid=1)
        assert (fa1, lg1) == (fa, lg)
    finally:
        lead.destroy()
        for h in helpers:
            h.destroy()
        pg.free()
        og.free()


def test_gpu_unitig_numbering_equals_host_walk(product, monkeypatch):
    """The graph build numbers k-mers along unitigs by list ranking on the GPU (dbg_gpu.hip);
    G2S_HOST_BUILD=1 keeps the sequential host walk.  Same unitigs, same adjacency (compared
    through k-mer strings, node ids differ), same fills.  V3 genome + a circular sequence
    (a unitig without a head, which the GPU leaves to the host walk)."""
    reads = product.G2S.synth_genome(200000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    import random
    rr = random.Random(11)
    ring = "".join(rr.choice("ACGT") for _ in range(400))  # not in the genome: an isolated circular unitig
    seqs.append(ring + ring[:40])  # k-1 = 30 bases of overlap close the circle (and 10 more)
    gl = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 60, 50, 500, 5))
    monkeypatch.setenv("G2S_HOST_BUILD", "1")
    gh = product.Graph.from_seqs(seqs, 31, 1)
    monkeypatch.delenv("G2S_HOST_BUILD")
    gg = product.Graph.from_seqs(seqs, 31, 1)
    try:
        assert (gg.num_kmers, gg.num_unitigs) == (gh.num_kmers, gh.num_unitigs)
        assert gg.validate() == (0, "") and gh.validate() == (0, "")
        rnd = random.Random(3)
        for s in (seqs[0], seqs[1], seqs[-1]):
            for _ in range(300):
                p = rnd.randrange(0, len(s) - 31)
                km = s[p:p + 31]
                a, b = gg.node(km), gh.node(km)
                assert a != 0xFFFFFFFF and b != 0xFFFFFFFF
                assert gg.node_string(a) == km and gh.node_string(b) == km
                # same neighbours, in the same (GATB) order
                assert [gg.node_string(x) for x in gg.successors(a)] == [gh.node_string(x) for x in gh.successors(b)]
                assert [gg.node_string(x) for x in gg.predecessors(a)] == [gh.node_string(x) for x in gh.predecessors(b)]
        sg = product.Session(gg, 0, d_err=500, randseed=3)
        sh = product.Session(gh, 0, d_err=500, randseed=3)
        try:
            rg = [_result_tuple(r) for r in sg.fill_batch(_gaps(product, gl))]
            rh = [_result_tuple(r) for r in sh.fill_batch(_gaps(product, gl))]
            assert rg == rh
        finally:
            sg.destroy()
            sh.destroy()
    finally:
        gg.free()
        gh.free()


@pytest.mark.parametrize("k", [15, 31, 47, 63])
def test_gpu_graph_build_solid_threshold_and_wide_kmers(product, monkeypatch, k):
    """GPU graph build (k-mer extraction + radix sort + run lengths, dbg_gpu.hip) against the
    host build: reads with N's, lower case, overlaps and `-solid 2`; 64- and 128-bit k-mers."""
    import random
    rr = random.Random(k)
    genome = "".join(rr.choice("ACGT") for _ in range(30000))
    reads = []
    for i in range(0, len(genome) - 400, 150):  # 400 bp reads every 150 bp: coverage 2-3
        r = genome[i:i + 400]
        if i % 900 == 0:
            r = r[:200] + "N" + r[201:]
        if i % 1350 == 0:
            r = r.lower()
        reads.append(r)
    reads.append(genome[:k - 1])  # shorter than k: contributes nothing
    graphs = {}
    for mode in ("host", "gpu"):
        if mode == "host":
            monkeypatch.setenv("G2S_HOST_BUILD", "1")
        else:
            monkeypatch.delenv("G2S_HOST_BUILD", raising=False)
        graphs[mode] = product.Graph.from_seqs(reads, k, 2)
    gh, gg = graphs["host"], graphs["gpu"]
    try:
        assert gg.num_kmers == gh.num_kmers and gg.num_kmers > 20000
        assert gg.num_unitigs == gh.num_unitigs
        assert gg.validate() == (0, "")
        for _ in range(500):
            p = rr.randrange(0, len(genome) - k)
            km = genome[p:p + k]
            a, b = gg.node(km), gh.node(km)
            assert (a == 0xFFFFFFFF) == (b == 0xFFFFFFFF)
            if a != 0xFFFFFFFF:
                assert gg.node_string(a) == km
                assert [gg.node_string(x) for x in gg.successors(a)] == [gh.node_string(x) for x in gh.successors(b)]
    finally:
        gg.free()
        gh.free()


def test_full_size_round_trip_c3(product):
    """BASELINE config 3 size (3 Mbp DBG, 10 000 gaps, k=31, -fuz 10, -dist-error 500) on
    the repeat-free V0 genome: every gap has exactly one path, so cut -> fill must give
    back the original bases, all upper case (safe), with 1 path, and the device state
    count has the closed form of an unbranched chain."""
    reads = product.G2S.synth_genome(3000000, 0, 20240101)
    genome = reads.splitlines()[1]
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 10000, 200, 1000, 20240103))
    pg = product.Graph.from_seqs([genome], 31, 1)
    assert pg.num_unitigs == 1
    sess = product.Session(pg, 0, d_err=500, randseed=1)
    res, tm = sess.fill_batch(_gaps(product, gaps), True)
    for g, r in zip(gaps, res):
        assert r.count == 1 and r.left_fuz == 0 and r.right_fuz == 0 and r.flags == product.G2S_GAP_PHASE_D
        pos = genome.index(g["left"]) + len(g["left"])
        assert r.fill == genome[pos:pos + g["gap_len"] + 31]
        assert r.draws == 1 + g["gap_len"] + 31  # 1 + (L - depth_at_stop), SURVEY A.3
    assert tm.retried_gaps == 0
    sess.destroy()
    pg.free()


def test_cli_binary_is_a_drop_in(product, oracle, tmp_path):
    """gap2seq_amd/Gap2Seq-core with the argv the reference wrapper builds
    (Gap2Seq.py:230-241) against the oracle CLI: identical FASTA and stdout."""
    k = 21
    seqs = cases.toy_genome(12, 20000, k, repeats=4, snp_every=301)
    reads = tmp_path / "reads.fa"
    reads.write_text("".join(">r%d\n%s\n" % (i, s) for i, s in enumerate(seqs)))
    g = seqs[0]
    recs = []
    for r in range(20):
        p = 100 + r * 900
        recs.append((">sc%d\n" % r) + cases.scaffold_record(g, k, 10, [(p, 50 + r, 50 + r + k), (p + 300, 20, 20 + k)]))
    scaf = tmp_path / "scaf.fa"
    scaf.write_text("\n".join(recs) + "\n")
    outs = {}
    for name, exe in (("gpu", os.path.join(ROOT, "gap2seq_amd", "Gap2Seq-core")),
                      ("cpu", os.path.join(os.path.dirname(oracle.ORACLE_SO), "g2s_oracle_cli"))):
        out = tmp_path / ("out_%s.fa" % name)
        res = subprocess.run([exe, "-k", str(k), "-fuz", "10", "-solid", "1", "-nb-cores", "1", "-dist-error", "100",
                              "-max-mem", "20", "-randseed", "4", "-reads", str(reads), "-filled", str(out),
                              "-scaffolds", str(scaf)], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stdout + res.stderr
        outs[name] = (out.read_text(), res.stdout.replace(str(out), "OUT"))
    assert outs["gpu"][0] == outs["cpu"][0]
    assert outs["gpu"][1] == outs["cpu"][1]
    assert "Filled 40 gaps out of 40" in outs["gpu"][1] or "Filled" in outs["gpu"][1]
    # the dispatcher: several sessions (here three on device 0, as -devices 0,0 -streams 2 minus
    # the lead) share the gap list; FASTA and stdout must not change
    out = tmp_path / "out_team.fa"
    res = subprocess.run([os.path.join(ROOT, "gap2seq_amd", "Gap2Seq-core"), "-k", str(k), "-fuz", "10", "-solid", "1",
                          "-nb-cores", "1", "-dist-error", "100", "-max-mem", "20", "-randseed", "4", "-reads", str(reads),
                          "-filled", str(out), "-scaffolds", str(scaf), "-devices", "0,0", "-streams", "2"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert out.read_text() == outs["cpu"][0]
    assert res.stdout.replace(str(out), "OUT") == outs["cpu"][1]
    # -fasta-width 70 (GATB's BankFasta line length, as recalled) and small batches (-stream-gaps): the same
    # records, data lines of at most 70 characters; the log is the same text
    out = tmp_path / "out_wrapped.fa"
    res = subprocess.run([os.path.join(ROOT, "gap2seq_amd", "Gap2Seq-core"), "-k", str(k), "-fuz", "10", "-solid", "1",
                          "-nb-cores", "1", "-dist-error", "100", "-max-mem", "20", "-randseed", "4", "-reads", str(reads),
                          "-filled", str(out), "-scaffolds", str(scaf), "-fasta-width", "70", "-stream-gaps", "6"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.replace(str(out), "OUT") == outs["cpu"][1]
    wrapped = out.read_text()
    assert all(len(ln) <= 70 for ln in wrapped.splitlines() if not ln.startswith(">"))
    unwrapped, cur = [], []
    for ln in wrapped.splitlines():
        if ln.startswith(">"):
            if cur:
                unwrapped.append("".join(cur))
            unwrapped.append(ln)
            cur = []
        else:
            cur.append(ln)
    unwrapped.append("".join(cur))
    assert "\n".join(unwrapped) + "\n" == outs["cpu"][0]


def test_per_gap_flow_with_filtered_reads(product, oracle, tmp_path):
    """The wrapper's per-gap flow (Gap2Seq.py:133-218): ReadFilter on the library's BAM for ONE gap, then
    Gap2Seq-core -left/-right/-length on the reads it extracted.  The filter's output is checked against its
    restatement, the fill against the oracle CLI on the same reads, and the filled bases against the genome
    the reads were simulated from."""
    import random
    import bamwriter as BW
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import readfilter_ref as RF
    rng = random.Random(77)
    # (-mean is the distance BETWEEN the two reads of a pair: the windows of ReadFilter.cpp:384-385 put the mate
    # on the breakpoint for that reading.  Only the left-hand window works in the reference (:388-389), so the gap
    # must be short enough for the left-hand mates and the flank reads to cover it.)
    k, rl, frag, sd = 31, 100, 400, 20
    mean = frag - 2 * rl
    glen, gap_at, gap_len, flank = 8000, 4000, 60, 41  # (flanks of k + fuz bases, as GapCutter cuts them)
    genome = "".join(rng.choice("ACGT") for _ in range(glen))
    recs, uid = [], 0
    for _ in range(2400):  # 60x
        ins = max(2 * rl, int(rng.gauss(frag, sd)))
        a = rng.randrange(0, glen - ins)
        ends = [(a, False), (a + ins - rl, True)]
        mapped = [not (st + rl > gap_at and st < gap_at + gap_len) for st, _ in ends]
        name = "p%05d" % uid
        uid += 1
        for i, (st, rev) in enumerate(ends):
            o = 1 - i
            flag = 1 | (64 if i == 0 else 128) | (0 if mapped[i] else 4) | (0 if mapped[o] else 8)
            if mapped[i] and rev:
                flag |= 16
            seq = genome[st:st + rl]
            if mapped[i]:
                pos, cig = st, "%dM" % rl
                stored = BW.revcomp(seq) if rev else seq
            elif mapped[o]:
                pos, cig, stored = ends[o][0], "", seq
            else:
                pos, cig, stored = -1, "", seq
            tid = 0 if pos >= 0 else -1
            recs.append(((tid if tid >= 0 else 1 << 30, pos, len(recs)), BW.record(name, flag, tid, pos, cig, stored, tid, ends[o][0] if mapped[o] else pos)))
    recs.sort(key=lambda r: r[0])
    bam = BW.bam_bytes([("scaffold1", glen)], [r[1] for r in recs], block=20000)
    bam_path = tmp_path / "lib.bam"
    bam_path.write_bytes(bam)
    filt = tmp_path / "tmp.reads.1.0"
    res = subprocess.run([os.path.join(ROOT, "gap2seq_amd", "ReadFilter"), "-reads", str(filt), "-scaffold", "scaffold1",
                          "-breakpoint", str(gap_at), "-flank-length", str(flank), "-gap-length", str(gap_len), "-bam", str(bam_path),
                          "-mean", str(mean), "-std-dev", str(sd)], capture_output=True, text=True, timeout=120)
    want = RF.read_filter(bam, mean, sd, "scaffold1", gap_at, gap_len, flank)
    assert res.returncode == 0 and res.stdout == want[1] and filt.read_text() == want[0]
    n_reads = filt.read_text().count(">")
    assert 20 < n_reads < len(recs) // 4  # a small part of the library
    left, right = genome[gap_at - flank:gap_at], genome[gap_at + gap_len:gap_at + gap_len + flank]
    outs = {}
    for name, exe in (("gpu", os.path.join(ROOT, "gap2seq_amd", "Gap2Seq-core")),
                      ("cpu", os.path.join(os.path.dirname(oracle.ORACLE_SO), "g2s_oracle_cli"))):
        out = tmp_path / ("tmp.filled.%s" % name)
        r = subprocess.run([exe, "-k", str(k), "-fuz", "10", "-solid", "2", "-nb-cores", "1", "-dist-error", "500", "-max-mem", "20",
                            "-randseed", "3", "-reads", str(filt), "-filled", str(out), "-left", left, "-right", right,
                            "-length", str(gap_len)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[name] = (out.read_text(), r.stdout.replace(str(out), "OUT"))
    assert outs["gpu"] == outs["cpu"]
    fill = "".join(ln for ln in outs["gpu"][0].splitlines() if not ln.startswith(">"))
    # single-gap mode writes the left flank, the fill and the right k-mer (Gap2Seq.cpp:262-266)
    assert fill.upper() == genome[gap_at - flank:gap_at + gap_len + k]


def _fuzz_regressions():
    return json.load(open(os.path.join(HERE, "golden", "fuzz_regressions.json")))


@pytest.mark.parametrize("idx", range(len(_fuzz_regressions())))
def test_fuzz_regressions(product, oracle, idx, monkeypatch):
    """Configurations on which tools/fuzz_parity.py (random graphs and parameters, GPU path
    against the oracle) once found a difference; `why` in the fixture names the cause."""
    cfg = _fuzz_regressions()[idx]
    k = cfg["k"]
    seqs = cases.toy_genome(cfg["gseed"], cfg["length"], k, repeats=cfg["repeats"], tandem=cfg["tandem"],
                            inverted=cfg["inverted"], snp_every=cfg["snp_every"])
    gaps = cases.cut_gaps(cfg["cseed"], seqs[0], k, fuz=cfg["fuz"], ngaps=cfg["ngaps"], min_len=cfg["min_len"],
                          max_len=cfg["max_len"], d_err=cfg["d_err"])
    if cfg["hbm_tier"]:
        monkeypatch.setenv("G2S_NO_LDS_TIER", "1")
    if cfg.get("force_segx"):  # (every gap through the large variant of the segment tier, on the host path)
        monkeypatch.setenv("G2S_FORCE_SEGX", "1")
        monkeypatch.setenv("G2S_RESIDENT", "0")
    c, f, _, _, _ = _check_batch(product, oracle, seqs, k, gaps, cfg["d_err"], cfg["skip"], cfg["allp"],
                                 seed=cfg["randseed"])
    assert c > 0 and f > 0


def test_large_variant_loses_no_traceback_start_over_many_runs(product, oracle, monkeypatch):
    """A deep list through g2s_fill_segw (G2S_FORCE_SEGX=1, -dist-error 2000, -all-upper) forty times in one process: the
    kernel's eight waves share the words that say where a gap's traceback starts — round 5's campaign found thread 0
    resetting them without a barrier in front of the pass that sets them: one run in a hundred of this very list lost a
    gap's start and the host's traceback stopped at its first state (draws 1).  Every gap against the oracle, every run."""
    monkeypatch.setenv("G2S_FORCE_SEGX", "1")
    k = 21
    seqs = cases.toy_genome(327597752, 1000000, k, repeats=200, tandem=0, inverted=0, snp_every=500)
    gaps = cases.cut_gaps(858877549, seqs[0], k, fuz=10, ngaps=200, min_len=1000, max_len=5000, d_err=2000)
    for _ in range(40):
        c, f, tm, _, _ = _check_batch(product, oracle, seqs, k, gaps, 2000, skip=True, allp=True, seed=67666)
        assert c >= 190 and tm.segx_tier_gaps == 200
