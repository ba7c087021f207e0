"""CPU tests of the product's host side: the C ABI loads and exports every symbol
include/g2s.h declares, the host graph builder agrees with the independent Python
restatement, host phase D agrees with the oracle on oracle-supplied DP tables, and
the fill path refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re
import zlib

import pytest

import cases
import pyref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol(product):
    names = set()
    for h in ("g2s.h", "g2s_test.h"):  # the reference-facing interface, and the unit tests' hooks
        hdr = open(os.path.join(ROOT, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        found = set(re.findall(r"\b(g2s_[a-z0-9_]+)\s*\(", hdr))
        assert all(n.startswith("g2s_test_") for n in found) == (h == "g2s_test.h") or h == "g2s.h"
        if h == "g2s.h":
            assert not any(n.startswith("g2s_test_") for n in found), "test hooks belong in include/g2s_test.h"
        names |= found
    assert len(names) >= 30
    so = ctypes.CDLL(product.library_path())
    for n in sorted(names):
        assert hasattr(so, n), "include/*.h declares %s but libg2s_hip.so does not export it" % n
    assert set(product._SIGS) == names  # the ctypes binding covers the whole header
    assert so.g2s_abi_version() == 6


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "gap2seq_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(d, f), errors="replace").read()
                assert "oracle" not in text.lower().replace("no oracle", ""), os.path.join(d, f)


@pytest.mark.parametrize("k", [4, 5, 8, 11, 21, 31, 32, 33, 63])
def test_graph_builder_matches_python_restatement(product, k):
    rng = cases.SplitMix(k)
    n = 300 if k < 12 else 1500
    seqs = [cases.random_dna(rng, n) for _ in range(2)]
    seqs[1] = seqs[1][:60] + "N" + seqs[0][40:200] + "n" + seqs[1][220:]
    for solid in (1, 2):
        g = product.Graph.from_seqs(seqs, k, solid, nthreads=3)
        p = pyref.Graph(seqs, k, solid)
        assert g.num_kmers == len(p.kmers)
        for s in seqs:
            for i in range(0, len(s) - k, 5):
                x = pyref.norm(s[i:i + k])
                v = g.node(s[i:i + k])
                if not p.contains(x):
                    assert v == product.G2S_INVALID_NODE
                    continue
                assert g.node_string(v) == x
                if pyref.revcomp(x) != x:  # orientation bit is unitig-relative: only v^1 == revcomp is promised
                    assert g.node(pyref.revcomp(x)) == v ^ 1
                for y in (x, pyref.revcomp(x)):
                    w = g.node(y)
                    assert [g.node_string(t) for t in g.successors(w)] == p.succ(y)
                    assert [g.node_string(t) for t in g.predecessors(w)] == p.pred(y)  # GATB order T,G,A,C
        g.free()


def test_unitig_order_numbers_a_chain_consecutively(product):
    g = cases.random_dna(cases.SplitMix(3), 5000)
    gr = product.Graph.from_seqs([g], 21, 1)
    assert gr.num_unitigs == 1
    ids = [gr.node(g[i:i + 21]) for i in range(len(g) - 20)]
    step = ids[1] - ids[0]
    # oriented ids advance by +2 (even orientation) or -2 (odd) along the whole unitig
    assert step in (2, -2) and all(b - a == step for a, b in zip(ids, ids[1:]))
    assert (ids[0] & 1) == (0 if step == 2 else 1)
    gr.free()


@pytest.mark.parametrize("k", [9, 15, 31, 33, 63])
def test_graph_tables_are_consistent(product, k):
    """g2s_graph_validate: the unitig-start bitmap, the successor table and the last-base table
    agree on tangled graphs (dispersed, tandem and inverted repeats, a second haplotype) — the
    invariants the kernels walk unitigs by — and the check survives the cache round trip."""
    for seed in range(6):
        seqs = cases.toy_genome(seed * 5 + k, 4000, k, repeats=seed, tandem=seed % 3, inverted=seed % 2,
                                snp_every=(0 if seed % 2 else 97))
        g = product.Graph.from_seqs(seqs, k, 1)
        assert g.validate() == (0, ""), (k, seed)
        g.free()


def test_graph_cache_roundtrip(product, tmp_path):
    seqs = cases.toy_genome(3, 2000, 15, repeats=3, snp_every=101)
    a = product.Graph.from_seqs(seqs, 15, 1)
    path = str(tmp_path / "g.g2s")
    a.save(path)
    b = product.Graph.load(path)
    assert (a.num_kmers, a.num_unitigs, a.k) == (b.num_kmers, b.num_unitigs, b.k)
    # the cache records -solid as well as -k: Gap2Seq-core reuses "<reads>.g2s" only when both are this run's
    lib = product.load_library()
    assert lib.g2s_graph_solid(a.h) == 1 and lib.g2s_graph_solid(b.h) == 1
    c = product.Graph.from_seqs(seqs + seqs, 15, 2)
    c.save(path)
    d = product.Graph.load(path)
    assert lib.g2s_graph_solid(d.h) == 2 and d.num_kmers == c.num_kmers
    c.free()
    d.free()
    for i in range(0, 1900, 7):
        km = seqs[0][i:i + 15]
        assert a.node(km) == b.node(km)
        assert a.successors(a.node(km)) == b.successors(b.node(km))
    a.free()
    b.free()


def test_synthetic_generator_is_deterministic(product):
    a = product.G2S.synth_genome(50000, 3, 20240101)
    b = product.G2S.synth_genome(50000, 3, 20240101)
    assert a == b and a.count(">") == 2
    assert zlib.crc32(a.encode()) == zlib.crc32(b.encode())
    s = product.G2S.synth_gaps(a, 31, 10, 20, 200, 1000, 20240103)
    assert s == product.G2S.synth_gaps(a, 31, 10, 20, 200, 1000, 20240103)
    lines = s.splitlines()
    assert len(lines) == 40
    genome = a.splitlines()[1]
    for hdr, seq in zip(lines[::2], lines[1::2]):
        glen = int(hdr.split("len=")[1])
        assert seq.count("N") == glen and len(seq) == glen + 2 * 41
        assert seq[:41] in genome and seq[-41:] in genome
    assert product.G2S.synth_genome(50000, 0, 1) != product.G2S.synth_genome(50000, 0, 2)


@pytest.mark.parametrize("seed", range(8))
def test_host_phase_d_matches_oracle_on_oracle_tables(product, oracle, seed):
    """post.cpp (D1 subgraph, D2 SCC/branch rule, D3 traceback) fed with the oracle's
    DP table through the g2s_test_post_gap hook."""
    k = [9, 11, 13, 15][seed % 4]
    seqs = cases.toy_genome(seed, 800, k, repeats=seed % 4, tandem=seed % 3, snp_every=(0 if seed % 2 else 83))
    e = [0, 4, 9, 20, 31][seed % 5] + k
    compared = 0
    for skip, allp in ((False, True), (False, False), (True, True)):
        og = oracle.OracleGraph(seqs, k, 1)
        pg = product.Graph.from_seqs(seqs, k, 1)
        rng = oracle.OracleRng(5)
        used = 0
        params = product.make_params(d_err=e, skip_confident=skip, all_paths=allp)
        for gp in cases.cut_gaps(seed, seqs[0], k, fuz=seed % 5 + 1, ngaps=12, min_len=1, max_len=50, d_err=e):
            o = oracle.fill_gap(og, rng, gp["left"], gp["right"], gp["gap_len"], e, gp["lmf"], gp["rmf"], skip, allp,
                                dump=True)
            states = [(pg.node(s[0]), s[1], s[2]) for s in o.states]
            r = product.test_post_gap(pg, params, product.Gap(gp["left"], gp["right"], gp["gap_len"], gp["lmf"],
                                                              gp["rmf"]), states, o.info.phaseC_count, o.lengths,
                                      o.info.reached_fuz, o.info.final_d, 5, used)
            used += o.info.draws
            if o.info.q7:
                continue
            assert (r.count, r.draws) == (o.count, o.info.draws)
            if o.phase_d:
                assert (r.left_fuz, r.right_fuz, r.fill) == (o.left_fuz, o.right_fuz, o.fill)
                if not skip:
                    assert r.substats == o.substats
            compared += 1
        og.free()
        pg.free()
    assert compared > 10


def test_worker_pool_runs_every_task_once(product):
    """The host pool behind the per-gap analysis/tracebacks: thousands of short parallel-for
    rounds (some with fewer tasks than threads, so idle workers arrive after the round is
    over) must run every task exactly once and never stall."""
    product.test_worker_pool(8, 3000, 40)
    product.test_worker_pool(16, 500, 1000)
    product.test_worker_pool(1, 10, 100)
    product.test_worker_pool(16, 2000, 10)  # fewer tasks than workers: only as many workers are woken as there are tasks


def test_product_rand_stream_matches_libc(product):
    """The flat glibc TYPE_3 stream the tracebacks read at precomputed offsets
    (Gap2Seq.cpp:178,1440,1513 use libc rand()), incl. the buffer compaction path."""
    libc = ctypes.CDLL("libc.so.6")
    for seed, skip in ((1, 0), (42, 1000), (20240101, 5000000)):
        libc.srand(seed)
        for _ in range(skip):
            libc.rand()
        got = product.test_rand_stream(seed, skip, 3000)
        assert got == [libc.rand() for _ in range(3000)]
    assert product.test_rand_stream(1, 0, 3) == [1804289383, 846930886, 1681692777]


def test_fill_path_fails_loudly_without_a_gpu(product):
    if product.G2S.device_count() > 0:
        pytest.skip("a GPU is present")
    g = product.Graph.from_seqs([cases.random_dna(cases.SplitMix(1), 500)], 11, 1)
    with pytest.raises(product.G2SError) as ei:
        product.Session(g, 0)
    assert ei.value.code == product.G2S_ERR_NO_DEVICE
    g.free()
