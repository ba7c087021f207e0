"""GPU tests of RESIDENT MODE (run with `-m gpu` on an MI355X): lists finished entirely on the device — the
segment tier's kernel, then phase D3 (rand() stream, stream offsets, tracebacks: gap2seq_amd/csrc/d3_device.hip)
on the same stream — against the CPU oracle and, field by field, against the host path of the same library
(g2s_api.hip: batch_stage1 + batches_stage2), which the reference parity suite has pinned since round 1.
Everything goes through the C ABI."""
import pytest

import cases
from test_gpu_parity import _check_batch, _gaps, _parse_scaffolds

pytestmark = pytest.mark.gpu


def _key(r):
    return (r.count, r.left_fuz, r.right_fuz, r.flags, r.draws, r.fill, r.fill_len, tuple(r.substats), r.phaseC_count,
            tuple(r.lengths), r.backtrace_msg)


@pytest.mark.parametrize("seed,skip", [(1, 0), (42, 5), (20240101, 4095), (7, 4096 * 300 + 17), (1, 3000000)])
def test_device_rand_stream_is_glibcs(product, seed, skip):
    """g2s_rand_fill (every block of 4096 values reached from the host's state with three jump polynomials,
    then the recurrence per lane) against the host's generator, which the CPU suite pins to libc's rand():
    70 000 values behind `skip` consumed ones, i.e. blocks in the first and, for the last case, the third 2^20."""
    n = 70000
    dev = product.test_device_rand(0, seed, skip, n)
    host = product.test_rand_stream(seed, skip, n)
    assert dev == host


def _run(product, monkeypatch, resident, seqs, k, gaps, e, pinned=False, seed=3, **kw):
    monkeypatch.setenv("G2S_RESIDENT", "1" if resident else "0")
    pg = product.Graph.from_seqs(seqs, k, 1)
    sess = product.Session(pg, 0, d_err=e, randseed=seed, **kw)
    try:
        # two lists in a row on one session: the second starts where the first left the rand() stream
        r1, t1 = sess.fill_batch(_gaps(product, gaps), True, pinned=pinned)
        r2, t2 = sess.fill_batch(_gaps(product, gaps[: max(1, len(gaps) // 3)]), True, pinned=pinned)
    finally:
        sess.destroy()
        pg.free()
    return [_key(r) for r in r1], [_key(r) for r in r2], t1, t2


@pytest.mark.parametrize("mode", ["default", "best_only", "all_upper", "unique"])
@pytest.mark.parametrize("pinned", [False, True])
def test_resident_mode_equals_the_host_path(product, monkeypatch, mode, pinned):
    """A branching genome (repeats + second haplotype): every field of every result, the fill text included, and
    the position the rand() stream is left at, with the caller's buffers pinned (written by the kernels) and
    not (staged)."""
    kw = dict(default={}, best_only=dict(all_paths=False), all_upper=dict(skip_confident=True), unique=dict(unique_paths=True))[mode]
    reads = product.G2S.synth_genome(200000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 700, 100, 900, 20240103))
    h1, h2, th, _ = _run(product, monkeypatch, False, seqs, 31, gaps, 500, **kw)
    d1, d2, td, td2 = _run(product, monkeypatch, True, seqs, 31, gaps, 500, pinned=pinned, **kw)
    assert th.resident_launches == 0 and td.resident_launches == 1 and td.resident_fallbacks == 0 and td2.resident_launches == 1
    assert d1 == h1
    assert d2 == h2  # (the second list began at the right place in the stream)
    assert (td.xB, td.sB, td.seg_tier_gaps, td.seg_segments, td.fill_bytes) == (th.xB, th.sB, th.seg_tier_gaps, th.seg_segments, th.fill_bytes)
    assert td.draw_dependent_gaps > 0 and td.d3_table_entries >= td.draw_dependent_gaps
    # a few closures per thousand hold a k-mer at two depths: on a list this short the host's threads finish those under
    # the trace kernel (lists that fill the chip, and deep ones, have them analysed by g2s_d2_*: the next test); with
    # -all-upper nothing is analysed at all
    assert (td.host_finished_gaps > 0) == (mode != "all_upper")
    assert sum(1 for r in d1 if r[0] > 0) > 600


@pytest.mark.parametrize("how", ["default", "host", "small", "large", "no_chains", "beside"])
def test_phase_d2_on_the_device_equals_the_hosts(product, monkeypatch, how):
    """The closures the fill kernels do not analyse themselves (a k-mer at several depths, more than 192 segments):
    analysed by g2s_d2_small / g2s_d2_big on a stream of their own and traced by the trace kernel from their runs (the
    default from 3 072 gaps on; "small": forced on a shorter list; "large": every one of them through the large
    instantiation, G2S_D2_BIG=2; "no_chains": the whole graph of runs through the component search; "beside": part of
    them taken by workgroups that poll the list while the fill kernel runs), or handed to the
    host's threads (G2S_DEVICE_D2=0: round 4's way, post.cpp) — every field of every result and the subgraph statistics
    equal the host path's either way."""
    reads = product.G2S.synth_genome(200000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 3300 if how in ("default", "host") else 1500, 100, 900, 20240103))
    for k in ("G2S_DEVICE_D2", "G2S_D2_BIG", "G2S_D2_NO_CHAINS", "G2S_D2_POLL"):
        monkeypatch.delenv(k, raising=False)
    for mode_kw in ({}, dict(all_paths=False)):
        h1, h2, th, _ = _run(product, monkeypatch, False, seqs, 31, gaps, 500, **mode_kw)
        if how == "host":
            monkeypatch.setenv("G2S_DEVICE_D2", "0")
        elif how != "default":
            monkeypatch.setenv("G2S_DEVICE_D2", "1")
        if how == "large":
            monkeypatch.setenv("G2S_D2_BIG", "2")
        if how == "no_chains":
            monkeypatch.setenv("G2S_D2_NO_CHAINS", "1")
        if how == "beside":  # (a few workgroups of g2s_d2_small beside the fill kernel, polling the list: G2S_D2_POLL=1)
            monkeypatch.setenv("G2S_D2_POLL", "1")
        d1, d2, td, td2 = _run(product, monkeypatch, True, seqs, 31, gaps, 500, pinned=True, **mode_kw)
        assert td.resident_launches == 1 and td.resident_fallbacks == 0
        assert d1 == h1 and d2 == h2
        assert (td.host_finished_gaps > 5) if how == "host" else (td.host_finished_gaps == 0)


@pytest.mark.parametrize("waves,ngaps", [("1", 600), ("4", 600), ("1", 1500), ("4", 1500)])
def test_trace_kernel_on_one_and_on_four_waves_per_gap(product, monkeypatch, waves, ngaps):
    """g2s_d3_trace gives a gap one wave or four (by default four on lists of up to 768 gaps: the kernel of a short list is
    its slowest gap): both instantiations on both sides of that limit (G2S_TRACE_WAVES), closures of the host's threads
    and of g2s_d2_* among them, every field of every result equal to the host path's."""
    reads = product.G2S.synth_genome(200000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, ngaps, 100, 900, 20240103))
    for k in ("G2S_DEVICE_D2", "G2S_TRACE_WAVES"):
        monkeypatch.delenv(k, raising=False)
    h1, h2, th, _ = _run(product, monkeypatch, False, seqs, 31, gaps, 500)
    monkeypatch.setenv("G2S_TRACE_WAVES", waves)
    for d2 in ("0", "1"):
        monkeypatch.setenv("G2S_DEVICE_D2", d2)
        d1, d2r, td, _ = _run(product, monkeypatch, True, seqs, 31, gaps, 500, pinned=True)
        assert td.resident_launches == 1 and td.resident_fallbacks == 0
        assert d1 == h1 and d2r == h2


def test_the_same_list_a_hundred_times_says_the_same(product, monkeypatch):
    """One list of 600 gaps a hundred times through one session (srand before every call): every field and the fill text of
    every gap as in the first call — what a race between the waves of a kernel shows as when it strikes once in many runs
    (tools/self_consistency.py does this with the bench lists, profiles/r05_self_consistency.txt)."""
    reads = product.G2S.synth_genome(200000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = _gaps(product, _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 600, 100, 900, 20240103)))
    pg = product.Graph.from_seqs(seqs, 31, 1)
    sess = product.Session(pg, 0, d_err=500, randseed=1)
    first = None
    for it in range(100):
        sess.srand(1)
        res, tm = sess.fill_batch(gaps, True)
        now = [(r.count, r.left_fuz, r.right_fuz, r.flags, r.draws, r.fill, tuple(r.substats), r.phaseC_count, tuple(r.lengths)) for r in res]
        if first is None:
            first = now
            assert tm.resident_launches == 1 and sum(1 for r in res if r.count > 0) > 400
        assert now == first, "call %d differs from the first" % it


def test_list_sizes_around_the_switch_points(product, monkeypatch):
    """The default choice of path by list size: below 256 gaps the host path; from 256 on the device, with the
    single-workgroup kernels of short lists up to 3 072 gaps and the multi-workgroup ones beyond.  Lists of 255, 256,
    3 072 and 3 073 gaps (prefixes of one list) against the host path, one session each so that every list starts
    at the same place in the stream; then the four in a row on one session (what one list cleans up behind itself is
    what the next one finds)."""
    reads = product.G2S.synth_genome(300000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    allg = _gaps(product, _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 3073, 60, 300, 20240103)))
    pg = product.Graph.from_seqs(seqs, 31, 1)
    sizes = (255, 256, 3072, 3073)
    try:
        want = {}
        monkeypatch.setenv("G2S_RESIDENT", "0")
        for n in sizes:
            s = product.Session(pg, 0, d_err=300, randseed=5)
            want[n] = [_key(r) for r in s.fill_batch(allg[:n])]
            s.destroy()
        chain = product.Session(pg, 0, d_err=300, randseed=5)
        want_chain = [[_key(r) for r in chain.fill_batch(allg[:n])] for n in sizes + (256,)]
        chain.destroy()
        monkeypatch.delenv("G2S_RESIDENT")
        for n in sizes:
            s = product.Session(pg, 0, d_err=300, randseed=5)
            res, tm = s.fill_batch(allg[:n], True, pinned=True)
            assert [_key(r) for r in res] == want[n], n
            assert tm.resident_launches == (1 if n >= 256 else 0) and tm.resident_fallbacks == 0, n
            s.destroy()
        chain = product.Session(pg, 0, d_err=300, randseed=5)
        got_chain = [[_key(r) for r in chain.fill_batch(allg[:n], pinned=True)] for n in sizes + (256,)]
        chain.destroy()
        assert got_chain == want_chain
    finally:
        pg.free()


def test_resident_mode_vs_oracle_on_the_bench_workload(product, oracle, monkeypatch):
    """BASELINE config 2's list (500 gaps) forced through resident mode, gap by gap against the oracle."""
    monkeypatch.setenv("G2S_RESIDENT", "1")
    reads = product.G2S.synth_genome(3000000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 500, 200, 1000, 20240103))
    c, f, tm, xb, sb = _check_batch(product, oracle, seqs, 31, gaps, 500, seed=1)
    assert (c, f) == (500, 500) and (tm.xB, tm.sB) == (xb, sb)
    assert tm.resident_launches == 1 and tm.resident_fallbacks == 0 and tm.seg_tier_gaps == 500


def test_one_deep_gap_does_not_send_a_list_to_the_host_path(product, oracle, monkeypatch):
    """A C2-shaped list of 2 000 gaps with ONE gap planted in its middle that outgrows the regular tier's capacities
    (6 kbp: more than 512 segments): the gap runs again in the large variant behind the fill kernel on the stream and
    rejoins the list's phase D3 — the list is finished on the device, nothing is given back, and every gap is the
    oracle's.  (Until round 3 one such gap discarded the whole attempt.)"""
    monkeypatch.delenv("G2S_RESIDENT", raising=False)
    reads = product.G2S.synth_genome(1000000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 2000, 200, 1000, 20240103))
    deep = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 3, 6000, 6000, 77))
    gaps = gaps[:1000] + deep[:1] + gaps[1000:1999]
    c, f, tm, xb, sb = _check_batch(product, oracle, seqs, 31, gaps, 500, seed=1)
    assert c >= 1995 and f >= 1900 and (tm.xB, tm.sB) == (xb, sb)
    assert tm.resident_launches == 1 and tm.resident_fallbacks == 0
    assert tm.segx_tier_gaps >= 1 and tm.seg_tier_gaps + tm.segx_tier_gaps == 2000 and tm.watchdog_gaps == 0


def test_two_lists_in_flight_equal_list_by_list(product, monkeypatch):
    """g2s_fill_begin / g2s_fill_end: seven lists of different lengths (resident and not, one empty), three (and two) in
    flight — the younger ones' kernels run while the oldest one's results cross the link, on twins of the session, their
    rand() streams continued from list to list on the device — against g2s_fill_batch list by list: every field, the
    fill text, and the one rand() stream."""
    reads = product.G2S.synth_genome(300000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    allg = _gaps(product, _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 4000, 100, 900, 20240103)))
    lists = [allg[:1500], allg[1500:1800], allg[1800:1810], [], allg[1810:3500], allg[3500:], allg[:700]]
    pg = product.Graph.from_seqs(seqs, 31, 1)
    try:
        monkeypatch.delenv("G2S_RESIDENT", raising=False)
        s = product.Session(pg, 0, d_err=500, randseed=11)
        want = [[_key(r) for r in s.fill_batch(L, pinned=True)] if L else [] for L in lists]
        s.destroy()
        for pinned, depth in ((True, 3), (False, 3), (True, 2)):
            s = product.Session(pg, 0, d_err=500, randseed=11)
            got, tm = s.fill_lists_overlapped(lists, pinned=pinned, depth=depth)
            tail = [_key(r) for r in s.fill_batch(allg[:300], pinned=True)]  # (the session itself goes on behind them)
            s.destroy()
            assert [[_key(r) for r in L] for L in got] == want
            assert tm.resident_launches == 1 and tm.resident_fallbacks == 0
        s = product.Session(pg, 0, d_err=500, randseed=11)
        for L in lists:
            if L:
                s.fill_batch(L, pinned=True)
        assert [_key(r) for r in s.fill_batch(allg[:300], pinned=True)] == tail
        s.destroy()
    finally:
        pg.free()


def test_other_entry_points_refuse_while_lists_are_in_flight(product, monkeypatch):
    """Between g2s_fill_begin and the matching g2s_fill_end the session's buffers and rand() stream belong to the lists in
    flight (include/g2s.h, ABI 5): every other entry point that would queue work on the session or move its stream
    returns G2S_ERR_STATE and changes nothing — the lists then end with the results of g2s_fill_batch list by list; and a
    session destroyed with lists begun and never ended waits for their kernels."""
    import ctypes as C
    reads = product.G2S.synth_genome(200000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    allg = _gaps(product, _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 1200, 100, 900, 20240103)))
    lists = [allg[:500], allg[500:900], allg[900:]]
    lib = product.load_library()
    pg = product.Graph.from_seqs(seqs, 31, 1)
    try:
        monkeypatch.delenv("G2S_RESIDENT", raising=False)
        s = product.Session(pg, 0, d_err=500, randseed=7)
        want = [[_key(r) for r in s.fill_batch(L, pinned=True)] for L in lists]
        s.destroy()
        s = product.Session(pg, 0, d_err=500, randseed=7)
        helper = product.Session(pg, 0, d_err=500, randseed=7)
        ctx = []
        for L in lists:
            arr, keep = product._gap_array(L)
            nbytes = lib.g2s_team_arena_bytes(s.h, arr, len(L))
            arena, rbuf = product.HostBuffer(max(1, nbytes)), product.HostBuffer(C.sizeof(product.g2s_result) * len(L))
            ctx.append((arr, keep, nbytes, arena, rbuf, rbuf.array(product.g2s_result, len(L))))
        for arr, keep, nbytes, arena, rbuf, res in ctx[:2]:
            product._check(lib.g2s_fill_begin(s.h, arr, len(res), res, C.cast(arena.p, C.c_void_p), nbytes))
        arr, keep, nbytes, arena, rbuf, res = ctx[2]
        ap = C.cast(arena.p, C.c_char_p)
        b = product._VP()
        team = (product._VP * 2)(s.h, helper.h)
        refused = [
            lib.g2s_fill_batch(s.h, arr, len(res), res, ap, nbytes),
            lib.g2s_batch_prepare(s.h, arr, len(res), C.byref(b)),
            lib.g2s_team_fill(team, 2, arr, len(res), 0, res, ap, nbytes, None),
            lib.g2s_session_set_team(s.h, (product._VP * 1)(helper.h), 1, 0),
            lib.g2s_session_srand(s.h, 99),
            lib.g2s_session_skip_draws(s.h, 5),
        ]
        assert refused == [product.G2S_ERR_STATE] * len(refused)
        assert b"in flight" in lib.g2s_last_error()
        assert lib.g2s_fill_in_flight(s.h) == 2
        product._check(lib.g2s_fill_begin(s.h, arr, len(res), res, C.cast(arena.p, C.c_void_p), nbytes))
        while lib.g2s_fill_in_flight(s.h) > 0:
            product._check(lib.g2s_fill_end(s.h))
        got = [[_key(product.FillResult(c[5][i], c[3].raw)) for i in range(len(c[5]))] for c in ctx]
        assert got == want
        s.srand(7)  # (no list in flight any more)
        # a session destroyed with lists begun and never ended: their kernels are waited for, the buffers go afterwards
        for arr, keep, nbytes, arena, rbuf, res in ctx[:2]:
            product._check(lib.g2s_fill_begin(s.h, arr, len(res), res, C.cast(arena.p, C.c_void_p), nbytes))
        s.destroy()
        helper.destroy()
        for c in ctx:
            c[3].free()
            c[4].free()
    finally:
        pg.free()


def test_lists_in_flight_when_one_does_not_end_on_the_device(product, monkeypatch):
    """Lists in flight draw from ONE rand() stream, and the list behind another has its phase D3 queued before the one
    in front has ended: its stream is generated on the device from the state the older list's kernels leave there.
    When the older list then does not end on the device (here: the K-th wait of the process is told to give up,
    G2S_RESIDENT_TEST_FALLBACK=nth:K, and the list takes the host path), what the younger one's kernels wrote is
    dropped and it runs again from the host's generator — every list as g2s_fill_batch list by list gives it, for
    every position of the failure; and the same with the device chain switched off."""
    reads = product.G2S.synth_genome(200000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    allg = _gaps(product, _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 2400, 100, 900, 20240103)))
    lists = [allg[:600], allg[600:1100], allg[1100:1500], allg[1500:2100], allg[2100:], allg[:400]]
    pg = product.Graph.from_seqs(seqs, 31, 1)
    try:
        monkeypatch.delenv("G2S_RESIDENT", raising=False)
        s = product.Session(pg, 0, d_err=500, randseed=5)
        want = [[_key(r) for r in s.fill_batch(L, pinned=True)] for L in lists]
        tail = [_key(r) for r in s.fill_batch(allg[:200], pinned=True)]
        s.destroy()
        for env in ({}, {"G2S_NO_DEVICE_CHAIN": "1"}, {"G2S_RESIDENT_TEST_FALLBACK": "nth"}):
            # (the wait counter is the process's: the K-th wait from here on = K + the waits done so far; "nth" alone
            # is filled in per position below)
            positions = range(len(lists) + 1) if "G2S_RESIDENT_TEST_FALLBACK" in env else [None]
            for pos in positions:
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                if pos is not None:
                    monkeypatch.setenv("G2S_RESIDENT_TEST_FALLBACK", "rel:%d" % pos)
                s = product.Session(pg, 0, d_err=500, randseed=5)
                got, tm = s.fill_lists_overlapped(lists, pinned=True)
                monkeypatch.delenv("G2S_RESIDENT_TEST_FALLBACK", raising=False)
                assert [_key(r) for r in s.fill_batch(allg[:200], pinned=True)] == tail, (env, pos)
                s.destroy()
                for k in env:
                    monkeypatch.delenv(k, raising=False)
                assert [[_key(r) for r in L] for L in got] == want, (env, pos)
    finally:
        pg.free()


def test_lists_on_one_session_long_short_long(product, monkeypatch):
    """The ready words of the host-finished gaps across lists of different length on ONE session: a long
    list sizes the side buffer, a short list runs, then a long list with more host-finished gaps than the short one
    has gaps — every list against the host path."""
    reads = product.G2S.synth_genome(200000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    allg = _gaps(product, _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 3000, 100, 900, 20240103)))
    pg = product.Graph.from_seqs(seqs, 31, 1)
    lists = [allg, allg[:5], allg, allg[:300], allg]
    try:
        monkeypatch.setenv("G2S_RESIDENT", "0")
        s = product.Session(pg, 0, d_err=500, randseed=9)
        want = [[_key(r) for r in s.fill_batch(L)] for L in lists]
        s.destroy()
        monkeypatch.setenv("G2S_RESIDENT", "1")
        monkeypatch.setenv("G2S_DEVICE_D2", "0")  # (the hand-over to the host's threads is what this test is about)
        s = product.Session(pg, 0, d_err=500, randseed=9)
        got, hosts = [], []
        for L in lists:
            res, tm = s.fill_batch(L, True, pinned=True)
            got.append([_key(r) for r in res])
            hosts.append(tm.host_finished_gaps)
            assert tm.resident_launches == 1 and tm.resident_fallbacks == 0
        s.destroy()
        assert got == want
        assert hosts[0] > 5  # (more host-finished gaps in the long list than the short list has gaps)
    finally:
        pg.free()


@pytest.mark.parametrize("seed", range(12))
def test_resident_mode_on_toy_graphs(product, oracle, monkeypatch, seed):
    """Small k, tandem repeats, inverted repeats: most lists hold closures the device leaves to the host's analysis
    (a k-mer at two depths: finished by the host under the trace kernel), gaps with both strands of a k-mer, gaps
    that outgrow the regular tier (they rerun in the large variant on the stream) — and every list is FINISHED ON THE
    DEVICE all the same (until round 3 most of these attempts were discarded); the results are the oracle's."""
    monkeypatch.setenv("G2S_RESIDENT", "1")
    monkeypatch.setenv("G2S_DEVICE_D2", "1")  # (short lists: by default their few such closures are the host threads')
    k = [9, 11, 13, 15, 17, 21][seed % 6]
    seqs = cases.toy_genome(seed, 1500, k, repeats=seed % 3, tandem=seed % 2, inverted=int(seed % 4 == 0), snp_every=(0 if seed % 2 else 97))
    e = [0, 9, 20, 31][seed % 4] + k
    gaps = cases.cut_gaps(seed, seqs[0], k, fuz=seed % 5 + 1, ngaps=60, min_len=1, max_len=80, d_err=e)
    for skip, allp in ((False, True), (False, False), (True, True)):
        c, f, tm, _, _ = _check_batch(product, oracle, seqs, k, gaps, e, skip, allp)
        assert tm.resident_launches == 1 and tm.resident_fallbacks == 0
        # (cyclic closures — tandem repeats — included: strong components and the safe-vertex rule ran on the device)
        assert tm.host_finished_gaps == 0


@pytest.mark.parametrize("seed", range(6))
def test_lists_in_flight_on_toy_graphs_vs_oracle(product, oracle, monkeypatch, seed):
    """The same kind of lists cut into three to five consecutive lists that are in flight together (g2s_fill_begin /
    g2s_fill_end, pinned and pageable buffers): the results in order are the oracle's for the one list — the rand()
    stream goes from list to list, on the device or through the host, whatever each list's way to its end was
    (tools/fuzz_parity.py --in-flight runs this comparison over random configurations)."""
    monkeypatch.setenv("G2S_RESIDENT", "1")
    k = [11, 15, 21, 31, 13, 17][seed]
    seqs = cases.toy_genome(100 + seed, 4000, k, repeats=seed % 3, tandem=seed % 2, inverted=int(seed % 3 == 0), snp_every=(0 if seed % 2 else 97))
    e = [20, 40, 9][seed % 3] + k
    gaps = cases.cut_gaps(100 + seed, seqs[0], k, fuz=seed % 5 + 1, ngaps=150, min_len=1, max_len=120, d_err=e)

    def in_flight(sess, structs):
        rng = cases.SplitMix(seed)
        n = len(structs)
        cuts = sorted({0, n} | {rng.randint(1, n - 1) for _ in range(2 + seed % 3)})
        lists = [structs[a:b] for a, b in zip(cuts, cuts[1:])]
        outs, tm = sess.fill_lists_overlapped(lists, pinned=seed % 2 == 0, depth=2 + seed % 2)
        return [r for part in outs for r in part], tm

    c, f, _, _, _ = _check_batch(product, oracle, seqs, k, gaps, e, False, True, run_product=in_flight)
    assert c > 100 and f > 20


def test_resident_mode_gives_a_list_back(product, monkeypatch):
    """G2S_RESIDENT_TEST_FALLBACK: the device's attempt is discarded after it ran; the host path must then
    produce the same results from the same stream position (nothing of the attempt may have stuck)."""
    reads = product.G2S.synth_genome(150000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 300, 100, 700, 20240103))
    h1, h2, _, _ = _run(product, monkeypatch, False, seqs, 31, gaps, 500)
    monkeypatch.setenv("G2S_RESIDENT_TEST_FALLBACK", "1")
    d1, d2, td, _ = _run(product, monkeypatch, True, seqs, 31, gaps, 500)
    assert td.resident_launches == 0 and td.resident_fallbacks == 1
    assert d1 == h1 and d2 == h2


def test_resident_mode_scaffold_records_and_the_skip_rule(product, oracle, monkeypatch):
    """Multi-gap records through g2s_execute_scaffolds: consecutive gaps of a record are coupled (a gap is not
    attempted when the previous one was filled past it, Gap2Seq.cpp:369,402) — the device's scan applies the rule;
    FASTA and log text against the oracle's execute()."""
    k = 21
    seqs = cases.toy_genome(11, 60000, k, repeats=4, tandem=0, inverted=0, snp_every=0)
    recs = []
    for r in range(40):
        start = 200 + r * 1400
        holes = [(300, 10, 40), (420, 15, 8), (460, 30, 3), (700, 80, 25)]
        recs.append(">rec%d\n%s" % (r, cases.scaffold_record(seqs[0][start:start + 1300], k, 6, holes)))
    scaf = "\n".join(recs) + "\n"
    og = oracle.OracleGraph(seqs, k, 1)
    ofa, olog, sm = oracle.execute_scaffolds(og, scaf, k, solid=1, d_err=60, max_fuz=6, randseed=1)
    og.free()
    out = {}
    for resident in ("0", "1"):
        monkeypatch.setenv("G2S_RESIDENT", resident)
        pg = product.Graph.from_seqs(seqs, k, 1)
        sess = product.Session(pg, 0, d_err=60, randseed=1)
        out[resident] = sess.execute_scaffolds(scaf, k, solid=1, max_fuz=6)
        sess.destroy()
        pg.free()
    assert out["1"] == out["0"]
    assert out["1"][0] == ofa and out["1"][1] == olog


def test_kernel_timing_is_sampled(product, monkeypatch):
    """Resident mode brackets one launch in N with HIP events (default eight; G2S_KERNEL_TIMING=all|off|sample:N):
    g2s_timing says which launches carry a duration; the results never depend on it."""
    reads = product.G2S.synth_genome(120000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = _gaps(product, _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 300, 100, 600, 20240103)))
    pg = product.Graph.from_seqs(seqs, 31, 1)
    try:
        seen = {}
        for mode, want in (("sample:2", [1, 0, 1, 0]), ("all", [1, 1, 1, 1]), ("off", [0, 0, 0, 0]), (None, [1, 0, 0, 0])):
            if mode is None:
                monkeypatch.delenv("G2S_KERNEL_TIMING", raising=False)
            else:
                monkeypatch.setenv("G2S_KERNEL_TIMING", mode)
            s = product.Session(pg, 0, d_err=500, randseed=3)
            got = []
            for _ in range(4):
                s.srand(3)
                res, tm = s.fill_batch(gaps, True)
                assert tm.resident_launches == 1 and tm.seg_launches == 1
                assert (tm.ms_fill_seg > 0) == (tm.seg_timed_launches == 1) and (tm.ms_d3 > 0) == (tm.seg_timed_launches == 1)
                got.append(tm.seg_timed_launches)
                seen.setdefault("res", [_key(r) for r in res])
                assert [_key(r) for r in res] == seen["res"]
            assert got == want, (mode, got)
            s.destroy()
    finally:
        pg.free()


@pytest.mark.parametrize("nsess,group,timing", [(1, 400, None), (2, 700, "all"), (3, 300, "off"), (4, 97, None), (4, 5000, "all")])
def test_team_on_the_devices_equals_one_session(product, monkeypatch, nsess, group, timing):
    """g2s_team_fill with the list finished on the devices: every session runs the fill kernel of the groups it
    pulls, the groups' records and closures are gathered on the lead's device and phase D3 runs once over the
    whole list (the stream offsets chain through the groups).  Same results as one session on the host path,
    same stream position afterwards; every group was handed out, in the list's order."""
    reads = product.G2S.synth_genome(200000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gl = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 1400, 100, 900, 20240103))
    gaps = _gaps(product, gl)
    monkeypatch.setenv("G2S_RESIDENT", "0")
    pg = product.Graph.from_seqs(seqs, 31, 1)
    solo = product.Session(pg, 0, d_err=500, randseed=9)
    want = [_key(r) for r in solo.fill_batch(gaps)]
    want2 = [_key(r) for r in solo.fill_batch(gaps[:200])]
    solo.destroy()
    monkeypatch.delenv("G2S_RESIDENT")
    if timing:  # HIP events around every resident launch, or none (default: one launch in eight); one group of
        monkeypatch.setenv("G2S_KERNEL_TIMING", timing)  # four sessions: the lead may not launch a fill kernel at all
    team = [product.Session(pg, 0, d_err=500, randseed=9) for _ in range(nsess)]
    try:
        got, tm = product.team_fill(team, gaps, group_size=group, want_timing=True)
        assert [_key(r) for r in got] == want
        assert tm.resident_launches == 1 and tm.resident_fallbacks == 0
        assert tm.seg_timed_launches == (0 if timing == "off" else tm.seg_launches if timing == "all" else tm.seg_timed_launches)
        assert (tm.ms_fill_seg > 0) == (tm.seg_timed_launches > 0)
        ngroups = -(-len(gaps) // group)
        assert tm.team_groups == ngroups and tm.team_sessions == nsess and tm.seg_launches == ngroups
        assert sum(tm.team_groups_by_session[i] for i in range(16)) == ngroups
        assert tm.seg_tier_gaps == len(gaps)
        got2 = product.team_fill(team, gaps[:200], group_size=group)  # (short list: the host path; the stream goes on)
        assert [_key(r) for r in got2] == want2
    finally:
        for s in team:
            s.destroy()
        pg.free()


@pytest.mark.parametrize("nsess", [2, 3, 4, 8])
def test_team_with_phase_d3_sharded_equals_one_session(product, monkeypatch, nsess):
    """g2s_team_fill, one group per session, the caller's buffers page-locked: every group stays on the device that
    filled it — its session classifies, builds the tables of, traces and writes the results of its own gaps; the
    sessions meet twice to place the groups in the one rand() stream (totals, then the composed group functions).
    Same results as one session on the host path, same stream position afterwards.  (The sessions share device 0 here:
    the logic is the one of N GPUs, the timing is not.)"""
    reads = product.G2S.synth_genome(200000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gl = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 1400, 100, 900, 20240103))
    deep = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 2, 6000, 6000, 77))
    gl = gl[:500] + deep[:1] + gl[500:1399]  # (one long gap among them)
    gaps = _gaps(product, gl)
    monkeypatch.setenv("G2S_RESIDENT", "0")
    pg = product.Graph.from_seqs(seqs, 31, 1)
    solo = product.Session(pg, 0, d_err=500, randseed=9)
    want = [_key(r) for r in solo.fill_batch(gaps)]
    want2 = [_key(r) for r in solo.fill_batch(gaps[:200])]
    solo.destroy()
    monkeypatch.delenv("G2S_RESIDENT")
    team = [product.Session(pg, 0, d_err=500, randseed=9) for _ in range(nsess)]
    try:
        group = -(-len(gaps) // nsess)
        got, tm = product.team_fill(team, gaps, group_size=group, want_timing=True, pinned=True)
        assert [_key(r) for r in got] == want
        assert tm.team_d3_sharded == 1 and tm.resident_launches == 1 and tm.resident_fallbacks == 0
        assert tm.team_groups == nsess and tm.team_sessions == nsess
        assert tm.seg_tier_gaps + tm.segx_tier_gaps == len(gaps)
        got2 = product.team_fill(team, gaps[:200], group_size=group)  # (short list: the host path; the stream goes on)
        assert [_key(r) for r in got2] == want2
        # the same list with the groups gathered on the lead's device (pageable buffers): the other form, same results
        team[0].srand(9)
        got3, tm3 = product.team_fill(team, gaps, group_size=group, want_timing=True)
        assert [_key(r) for r in got3] == want and tm3.team_d3_sharded == 0
    finally:
        for s in team:
            s.destroy()
        pg.free()


@pytest.mark.parametrize("nsess", [2, 4])
def test_stream_of_lists_over_a_team_equals_one_session(product, monkeypatch, nsess):
    """A caller that streams lists over several GPUs (Gap2Seq-core -devices, bench.py --gpus N --stream-lists K):
    g2s_session_set_team(lead, helpers, G2S_GROUP_PER_SESSION) and then g2s_fill_batch list after list — every list of at
    least 512 gaps per session is cut into ONE share per session (each GPU fills, traces and writes a whole share;
    phase D3 sharded), shorter ones run on the lead alone, and the one rand() stream runs on from share to share and
    from list to list: the results are those of one session, list by list.  (The sessions share device 0 here.)"""
    reads = product.G2S.synth_genome(200000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    allg = _gaps(product, _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 4200, 100, 900, 20240103)))
    lists = [allg[:2100], allg[2100:4200], allg[:300], allg[1000:3300]]
    monkeypatch.delenv("G2S_RESIDENT", raising=False)
    pg = product.Graph.from_seqs(seqs, 31, 1)
    solo = product.Session(pg, 0, d_err=500, randseed=13)
    want = [[_key(r) for r in solo.fill_batch_onecall(L, pinned=True)] for L in lists]
    solo.destroy()
    team = [product.Session(pg, 0, d_err=500, randseed=13) for _ in range(nsess)]
    try:
        team[0].set_team(team[1:], product.G2S_GROUP_PER_SESSION)
        for L, w in zip(lists, want):
            got, tm = team[0].fill_batch_onecall(L, pinned=True, want_timing=True)
            assert [_key(r) for r in got] == w
            if len(L) >= 512 * nsess:
                assert tm.team_d3_sharded == 1 and tm.team_groups == nsess and tm.resident_fallbacks == 0
            else:
                assert tm.team_groups == 0  # (too short for the team: the lead alone)
        team[0].set_team([], 0)
    finally:
        for s in team:
            s.destroy()
        pg.free()


def test_long_lists_go_slice_by_slice(product, monkeypatch):
    """A list beyond 20 480 gaps is filled in slices of about 16 384 (the draw-count tables of phase D3 grow with the
    square of the gaps they chain through); the rand() stream runs on from slice to slice, a slice ends where a
    record ends.  36 000 gaps through g2s_fill_batch, and through a team of two sessions, against the host path."""
    reads = product.G2S.synth_genome(400000, 3, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gl = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 36000, 60, 400, 20240103))
    gaps = _gaps(product, gl)
    for i in range(1, len(gaps), 3):  # records of three gaps: the second and third carry a skip rule
        gaps[i].skip_dep = 5
        if i + 1 < len(gaps):
            gaps[i + 1].skip_dep = 0
    pg = product.Graph.from_seqs(seqs, 31, 1)
    try:
        monkeypatch.setenv("G2S_RESIDENT", "0")
        s0 = product.Session(pg, 0, d_err=300, randseed=4)
        want = [_key(r) for r in s0.fill_batch_onecall(gaps)]
        s0.destroy()
        monkeypatch.delenv("G2S_RESIDENT")
        s1 = product.Session(pg, 0, d_err=300, randseed=4)
        got = [_key(r) for r in s1.fill_batch_onecall(gaps)]
        tm = product.g2s_timing()
        product._check(product.load_library().g2s_session_last_timing(s1.h, tm))
        s1.destroy()
        assert got == want
        assert tm.resident_launches == 2 and tm.resident_fallbacks == 0 and tm.seg_tier_gaps == len(gaps)
        team = [product.Session(pg, 0, d_err=300, randseed=4) for _ in range(2)]
        got2, tm2 = product.team_fill(team, gaps, group_size=18000, want_timing=True)
        for s in team:
            s.destroy()
        assert [_key(r) for r in got2] == want and tm2.resident_launches == 2 and tm2.team_groups == 4
        assert sum(1 for r in want if r[3] & product.G2S_GAP_SKIPPED) >= 1  # (a right fuz beyond 0 is rare on this genome)
    finally:
        pg.free()


# ---- race hunting (round 6): the kernels' synchronisation is hand-written — wave-local waits, workgroup barriers,
# spin-waits across streams — and a missing barrier shows only when the waves' timing happens to expose it (round 5's
# in g2s_fill_segw's tail: once in a hundred runs, green suites for two rounds).  Two instrumented builds of the same
# sources (csrc/sync_debug.h; built by __graft_entry__.build()) move that timing: a pseudo-random sleep at every
# synchronisation point, and every wait widened to everything outstanding with every barrier taken twice.  Every
# kernel path, dozens of calls per build: no call may differ from its run's first, and the three builds must agree.
def _race_hunt(path, runs, library):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "gap2seq_amd", library, "libg2s_hip.so") if library else os.path.join(root, "gap2seq_amd", "libg2s_hip.so")
    if not os.path.exists(so):
        pytest.fail("%s is missing: __graft_entry__.build() makes it" % so)
    env = dict(os.environ, G2S_LIBRARY=so)
    for v in ("G2S_RESIDENT", "G2S_DEVICE_D2", "G2S_FORCE_SEGX", "G2S_NO_SEG_TIER"):
        env.pop(v, None)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "race_hunt.py"), path, str(runs)], env=env, capture_output=True,
                         text=True, timeout=900)
    rows = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert rows, "race_hunt.py %s with %s: %s" % (path, library or "the product build", out.stderr[-2000:])
    return rows[-1]


@pytest.mark.parametrize("path,runs", [("seg2", 60), ("seg", 30), ("segw", 50), ("d2", 50)])
def test_race_hunting_builds_give_the_product_builds_results(path, runs):
    """tools/race_hunt.py: one list `runs` times on one kernel path (seg2: two waves per gap + four-wave trace kernel +
    closures handed to the host early; seg: one wave per gap, g2s_d2_small, the long list's phase D3; segw: the
    eight-wave kernel on a -dist-error 2000 list, closures on the host's threads; d2: the same list through
    g2s_d2_small / g2s_d2_big) with the product build, the jitter build and the paranoid build."""
    rows = [_race_hunt(path, runs, lib) for lib in ("", "_jit", "_par")]
    for r in rows:
        assert r["differ"] == 0, "%s, %s: %d of %d calls differ from the first" % (path, r["library"], r["differ"], r["runs"])
        assert r["resident_launches"] == 1 and r["fallbacks"] == 0 and r["filled"] == r["gaps"], r
    assert len({r["digest"] for r in rows}) == 1, "the builds disagree on %s: %r" % (path, [(r["library"], r["digest"]) for r in rows])
    if path == "d2":
        assert rows[0]["host_finished"] <= 1 and rows[0]["segx_tier_gaps"] > 0
    if path == "segw":
        assert rows[0]["host_finished"] > 0 and rows[0]["segx_tier_gaps"] > 0


@pytest.mark.parametrize("variant,n", [(0, 600), (1, 600), (3, 2500), (0, 3500)])
def test_single_path_gaps_traced_by_the_fill_kernel(product, monkeypatch, variant, n):
    """Round 6: a gap whose traceback has no choice to make (one path length, no entry of the traceback closure with
    several parents) is traced by its fill kernel's own wave — text, case, fuz values, draws — while the rest of the
    list is still being searched (fill_seg.hip, G2S_DEVA_TRACED); phase D3 only counts its draws.  On a genome without
    bubbles that is nearly every gap.  Against the same lists with the switch off (every gap through g2s_d3_trace),
    field by field and in the position the rand() stream is left at; short lists (two waves per gap) and long ones."""
    reads = product.G2S.synth_genome(300000, variant, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, n, 100, 900, 20240103))
    monkeypatch.setenv("G2S_TRACE_IN_FILL", "0")
    a1, a2, ta, _ = _run(product, monkeypatch, True, seqs, 31, gaps, 500, pinned=True)
    monkeypatch.setenv("G2S_TRACE_IN_FILL", "1")
    b1, b2, tb, _ = _run(product, monkeypatch, True, seqs, 31, gaps, 500, pinned=True)
    c1, c2, tc, _ = _run(product, monkeypatch, True, seqs, 31, gaps, 500, pinned=False)
    assert ta.resident_launches == 1 and tb.resident_launches == 1 and tb.resident_fallbacks == 0
    assert ta.traced_in_fill_gaps == 0
    assert tb.traced_in_fill_gaps > (len(gaps) * 8 // 10 if variant == 0 else 0)
    assert tc.traced_in_fill_gaps == tb.traced_in_fill_gaps
    assert b1 == a1 and b2 == a2
    assert c1 == a1 and c2 == a2
    assert tb.fill_bytes == ta.fill_bytes


@pytest.mark.parametrize("variant,n", [(3, 600), (2, 1500), (3, 2048)])
def test_short_lists_guess_the_tracebacks_of_their_early_gaps(product, monkeypatch, variant, n):
    """Round 6: on a short list (two waves per gap: the launch lasts as long as its slowest gap) a traceback WITH choices
    is guessed by its fill wave — first length, first parent — only when the gap's search ends early: among the first
    87 % of the list in the order the searches end (G2S_GUESS_PERCENT) and within 100 k cycles; the trace kernel sends
    what differs.  Which gaps guess depends on the waves' timing, the results must not: against the same list with
    guesses off, with every gap guessing and with a third of them, three calls each, field by field and in the
    position the rand() stream is left at."""
    reads = product.G2S.synth_genome(300000, variant, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, n, 100, 900, 20240103))
    monkeypatch.setenv("G2S_TRACE_GUESS", "0")
    a1, a2, ta, _ = _run(product, monkeypatch, True, seqs, 31, gaps, 500, pinned=True)
    assert ta.resident_launches == 1 and ta.guessed_in_fill_gaps == 0
    monkeypatch.delenv("G2S_TRACE_GUESS")
    guessed = {}
    for pct in ("", "100", "33"):
        if pct:
            monkeypatch.setenv("G2S_GUESS_PERCENT", pct)
        for rep in range(3):
            b1, b2, tb, _ = _run(product, monkeypatch, True, seqs, 31, gaps, 500, pinned=True)
            assert tb.resident_launches == 1 and tb.resident_fallbacks == 0
            assert b1 == a1 and b2 == a2, "guesses (G2S_GUESS_PERCENT=%s, call %d) changed a result" % (pct or "default", rep)
            assert tb.fill_bytes == ta.fill_bytes
            guessed[pct] = tb.guessed_in_fill_gaps
            assert tb.guessed_groups >= tb.guessed_groups_resent
    monkeypatch.delenv("G2S_GUESS_PERCENT")
    assert 0 < guessed[""] <= len(gaps) * 87 // 100 + 1
    assert guessed["33"] <= len(gaps) * 33 // 100 + 1
    assert guessed["100"] >= guessed["33"]


# ---- round 6: one list over several PROCESSES, a GPU and a share each (g2s_share_*, gap2seq_amd/shard.py) -----------
class _ThreadComm:
    """shard.fill_share's communicator over the threads of one process (a barrier and a shared table)."""

    def __init__(self, rank, world, table, barrier):
        self.rank, self.world, self._t, self._b = rank, world, table, barrier

    def all_gather(self, values, maxlen=None):
        assert maxlen is None or len(values) <= maxlen
        self._t[self.rank] = list(values)
        self._b.wait()
        out = [list(v) for v in self._t]
        self._b.wait()
        return out


@pytest.mark.parametrize("world,variant", [(2, 3), (3, 3), (4, 1)])
def test_shares_of_a_list_on_ranks_of_their_own_equal_one_session(product, monkeypatch, world, variant):
    """Every 'rank' a session of its own (here: threads of one process on one device; bench.py --rank-per-gpu runs the
    same protocol over gloo) with the same seed: share by share the results of g2s_fill_batch on the whole list, and every
    rank's generator behind the list where the one session's is — a second list says so."""
    import ctypes as C
    import threading
    from gap2seq_amd import shard
    monkeypatch.delenv("G2S_RESIDENT", raising=False)
    reads = product.G2S.synth_genome(300000, variant, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = _gaps(product, _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 2400, 100, 900, 20240103)))
    second = gaps[:900]
    pg = product.Graph.from_seqs(seqs, 31, 1)
    solo = product.Session(pg, 0, d_err=500, randseed=5)
    want1 = [_key(r) for r in solo.fill_batch(gaps, pinned=True)]
    want2 = [_key(r) for r in solo.fill_batch(second, pinned=True)]
    solo.destroy()
    lib = product.load_library()
    sessions = [product.Session(pg, 0, d_err=500, randseed=5) for _ in range(world)]
    table, barrier = [None] * world, threading.Barrier(world)
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            comm = _ThreadComm(r, world, table, barrier)
            got = []
            for lst in (gaps, second):
                lo, hi = shard.share_bounds(len(lst), world)[r]
                arr, keep = product._gap_array(lst[lo:hi])
                n = hi - lo
                nbytes = lib.g2s_team_arena_bytes(sessions[r].h, arr, n)
                abuf, rbuf = product.HostBuffer(max(1, nbytes)), product.HostBuffer(C.sizeof(product.g2s_result) * max(1, n))
                res = rbuf.array(product.g2s_result, max(1, n))
                draws = shard.fill_share(product, sessions[r], comm, arr, n, res, C.cast(abuf.p, C.c_void_p), nbytes)
                assert draws is not None and draws > 0
                raw = abuf.raw
                got.append((lo, hi, draws, [_key(product.FillResult(res[i], raw)) for i in range(n)]))
                abuf.free(); rbuf.free()
            out[r] = got
        except BaseException as e:  # noqa: B902 (a failing rank must not leave the others at the barrier)
            errs.append((r, repr(e)))
            barrier.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for s_ in sessions:
        s_.destroy()
    pg.free()
    assert not errs, errs
    for li, want in enumerate((want1, want2)):
        assert len({out[r][li][2] for r in range(world)}) == 1  # every rank knows what the list drew
        got = []
        for r in range(world):
            got += out[r][li][3]
        assert got == want, "list %d" % li


def test_bench_one_rank_per_gpu_under_torchrun_two_ranks_on_one_device():
    """The launch the driver makes for N = 2 when the launcher pins a device per rank: two ranks, each seeing ONE GPU
    (here the same one), bench.py falls back to one rank per GPU by itself, checks the shares against the one-GPU
    results and prints one line."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="0")
    for v in ("G2S_RESIDENT", "G2S_DEVICE_D2"):
        env.pop(v, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--gaps", "3000"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["config"]["mode"] == "rank_per_gpu" and out["n_gpus"] == 2 and out["equals_one_gpu_result"] is True
    assert out["filled"] == 3000 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["config"]["devices_seen_by_rank"] == [1, 1]
