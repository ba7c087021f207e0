"""The algorithm of the segment tier (tests/seg_model.py) against the CPU oracle, without a GPU:
DP table, work counters, phase C outcome, Q7 and — through the host half of phase D
(g2s_test_post_closure) — fill, fuz values, draw counts, safe/unsafe case and subgraph statistics."""
import pytest

import cases
import seg_model as M


def _model_gap(pg, k, g, e, allp=True, skip=False):
    lmf, rmf = g["lmf"], g["rmf"]
    left, right = g["left"], g["right"]
    lseeds = [pg.node(left[d:d + k]) for d in range(lmf + 1)]
    rseeds = [pg.node(right[len(right) - k - d:len(right) - d]) for d in range(rmf + 1)]
    targets = [pg.node(right[d:d + k]) for d in range(rmf + 1)]
    return M.Gap(g["gap_len"], e, lmf, rmf, lseeds, rseeds, targets, all_paths=allp, skip_confident=skip)


STATS = {}


def _parent_sets(records, xp):
    extra = {}
    for x in xp:
        extra.setdefault(x >> 32, set()).add(x & 0xFFFFFFFF)
    out = []
    for i, (_, _, _, pred) in enumerate(records):
        out.append(set() if pred < 0 else ({pred & M.SUB_MORE - 1} | extra.get(i, set())))
    return out


def check_config(product, oracle, seqs, k, gaps, e, allp=True, skip=False, seed=5, stats=None):
    pg = product.Graph.from_seqs(seqs, k, 1)
    og = oracle.OracleGraph(seqs, k, 1)
    tb = M.Tables(product, pg)
    params = product.make_params(d_err=e, skip_confident=skip, all_paths=allp, randseed=seed)
    compared = nq7 = 0
    nseg_path = STATS.setdefault("on_segments", [0])
    try:
        for gi, g in enumerate(gaps):
            rng = oracle.OracleRng(seed)
            o = oracle.fill_gap(og, rng, g["left"], g["right"], g["gap_len"], e, g["lmf"], g["rmf"], skip, allp, dump=True)
            rng.free()
            mg = _model_gap(pg, k, g, e, allp, skip)
            m = M.fill_model(tb, mg, stats=stats)
            if o.info.q7:
                nq7 += 1
                continue
            what = "gap %d" % gi
            want = sorted((s, d, c) for s, d, c in o.states)
            got = sorted((pg.node_string(v), d, c) for v, d, c in M.expand_states(m))
            assert got == want, what
            assert (m.xB, m.sB) == (o.info.ctr[2], o.info.ctr[3]), what
            assert not m.q7, what + ": the model flags Q7, the oracle does not"
            assert m.c_count == o.info.phaseC_count and m.lengths == o.lengths, what
            if o.phase_d:
                assert m.reached_j == o.info.reached_fuz and m.final_d == o.info.final_d, what
                pgap = product.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"])
                # the host's expansion of the closure segments = the model's per-state records (the side
                # list as a set per state: the kernel lists a state's parents in arrival order)
                recs, xps = product.test_seg_expand(pg, params, pgap, m.compact, m.lengths, m.reached_j, len(m.records),
                                                    len(m.xp))
                assert [x[:3] for x in recs] == [x[:3] for x in m.records], what
                assert _parent_sets(recs, xps) == _parent_sets(m.records, sorted(m.xp)), what
                r = product.test_post_closure(pg, params, pgap, recs, xps, m.c_count, m.lengths, m.reached_j, m.final_d,
                                              seed, 0)
                # and the analysis + traceback run on the segments themselves (what the batch path does
                # whenever no k-mer occurs at two depths of the closure)
                r2, on_segments = product.test_post_segments(pg, params, pgap, m.compact, m.c_count, m.lengths, m.reached_j,
                                                             m.final_d, seed, 0)
                if on_segments:
                    nseg_path[0] += 1
                    assert (r2.count, r2.left_fuz, r2.right_fuz, r2.draws, r2.fill, r2.flags) == \
                        (r.count, r.left_fuz, r.right_fuz, r.draws, r.fill, r.flags), what
                    if not skip:
                        assert r2.substats == r.substats, what
                assert r.count == o.count, what
                assert (r.left_fuz, r.right_fuz, r.draws) == (o.left_fuz, o.right_fuz, o.info.draws), what
                assert r.fill == o.fill, what
                if not skip:
                    assert r.substats == o.substats, what
            compared += 1
    finally:
        pg.free()
        og.free()
    return compared, nq7


@pytest.mark.parametrize("seed", range(12))
def test_model_toy_graphs_all_modes(product, oracle, seed):
    k = [5, 7, 9, 11, 13, 15][seed % 6]
    seqs = cases.toy_genome(seed, 900, k, repeats=seed % 4, tandem=seed % 3, inverted=int(seed % 5 == 0),
                            snp_every=(0 if seed % 2 else 83))
    e = [0, 4, 9, 20, 31][seed % 5] + k
    # (the Python model is slow on the tangled graphs of k <= 9: fewer gaps there)
    gaps = cases.cut_gaps(seed, seqs[0], k, fuz=seed % 5 + 1, ngaps=40 if k >= 11 else 10, min_len=1, max_len=60, d_err=e)
    total = 0
    for skip, allp in ((False, True), (False, False), (True, True)):
        c, q = check_config(product, oracle, seqs, k, gaps, e, allp, skip)
        assert c + q == len(gaps)
        total += c
    assert total >= (90 if k >= 11 else 0)


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_model_k31_default_parameters(product, oracle, variant):
    reads = product.G2S.synth_genome(200000, variant, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    from test_gpu_parity import _parse_scaffolds
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 40, 50, 400, 20240103))
    stats = []
    before = STATS.setdefault("on_segments", [0])[0]
    c, q = check_config(product, oracle, seqs, 31, gaps, 500, stats=stats)
    assert c == 40
    assert STATS["on_segments"][0] - before >= 36  # nearly every closure has each k-mer at one depth only


def test_model_repeated_kmers_are_analysed_on_runs(product, oracle):
    """Closures in which a k-mer occurs at several depths (tandem repeats, bubbles of unequal length): the
    host analyses them on runs of k-mers (post.cpp: seg_analyze_runs).  check_config compares fill, flags,
    draws and the subgraph statistics of that analysis with the per-state one for every closure; here nearly
    every gap's closure stays on segments (with the run analysis switched off, G2S_STATE_D2=1, a third of
    this set falls back to per-state records)."""
    k = 11
    seqs = cases.toy_genome(21, 3000, k, repeats=4, tandem=6, snp_every=97)
    gaps = cases.cut_gaps(10, seqs[0], k, fuz=7, ngaps=40, min_len=1, max_len=80, d_err=30)
    before = STATS.setdefault("on_segments", [0])[0]
    c, q = check_config(product, oracle, seqs, k, gaps, 30)
    assert c + q == 40
    assert STATS["on_segments"][0] - before >= 25


def test_model_tandem_flanks_and_fuz_extremes(product, oracle):
    """Q6 territory (seed states reached by the DP as well) and fuz 0 / large fuz."""
    k = 11
    seqs = cases.toy_genome(21, 3000, k, repeats=4, tandem=6, snp_every=97)
    for fuz in (0, 1, 7, 15):
        gaps = cases.cut_gaps(3 + fuz, seqs[0], k, fuz=fuz, ngaps=40, min_len=1, max_len=80, d_err=30)
        c, q = check_config(product, oracle, seqs, k, gaps, 30)
        assert c + q == 40 and c >= 20


@pytest.mark.parametrize("k,length", [(5, 60), (5, 100), (7, 200)])
def test_model_small_k_without_strand_collisions(product, oracle, k, length):
    """k = 5 and 7 on genomes over {A, C}: no gap is a Q7 case, so every one is compared — dense graphs,
    closures full of cycles, saturating counts (the toy graphs above leave almost nothing to compare at these k)."""
    for seed in range(3):
        seqs = [cases.one_strand_genome(seed, length)]
        e = [0, 4, 9, 20, 31][seed % 5] + k
        gaps = cases.cut_gaps(seed, seqs[0], k, fuz=seed % 3 + 1, ngaps=16, min_len=1, max_len=length // 4, d_err=e)
        for skip, allp in ((False, True), (False, False), (True, True)):
            c, q = check_config(product, oracle, seqs, k, gaps, e, allp, skip)
            assert (c, q) == (len(gaps), 0)
