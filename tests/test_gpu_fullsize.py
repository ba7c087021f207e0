"""GPU parity at BASELINE.json's STATED sizes (run with `-m gpu` on an MI355X): every config
of BASELINE.json that test_gpu_parity.py only covers on a smaller graph or list is run here
at full size — C1's stand-in (2.9 Mbp, multi-gap simulated scaffolds through the two command
lines), C4 (60 Mbp, k=63, 2 000 gaps) and C5 (3 Mbp, -dist-error 2000, 1 000 gaps of 2-5 kbp).
Bit-exact against the CPU oracle wherever the oracle finishes in about a minute, and through
size-independent properties (cut -> fill -> original bases on a repeat-free genome) beyond.
"""
import os
import subprocess

import pytest

import cases
from test_gpu_parity import _assert_gap_equal, _check_batch, _compare_with_oracle_in_parallel, _gaps, _parse_scaffolds

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _seqs(reads):
    return [ln for ln in reads.splitlines() if not ln.startswith(">")]


def test_c4_full_size_round_trip_k63_60mbp(product):
    """BASELINE config 4 at its stated size on the repeat-free V0 genome: 60 Mbp, k=63 (128-bit
    k-mers, the 110 M oriented-id space, the two-pass radix sort of the graph build), 2 000 gaps
    of 200-1000 bp.  Every gap has exactly one path: the fill is the genome slice, all upper
    case, 1 path, draws = 1 + g + k."""
    k = 63
    reads = product.G2S.synth_genome(60000000, 0, 20240101)
    genome = reads.splitlines()[1]
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, k, 10, 2000, 200, 1000, 20240103))
    assert len(gaps) == 2000
    pg = product.Graph.from_seqs([genome], k, 1)
    try:
        assert pg.num_kmers == 60000000 - k + 1 and pg.num_unitigs == 1
        sess = product.Session(pg, 0, d_err=500, randseed=1)
        res, tm = sess.fill_batch(_gaps(product, gaps), True)
        sess.destroy()
        for i, (g, r) in enumerate(zip(gaps, res)):
            assert r.count == 1 and r.left_fuz == 0 and r.right_fuz == 0 and r.flags == product.G2S_GAP_PHASE_D
            assert r.draws == 1 + g["gap_len"] + k and len(r.fill) == g["gap_len"] + k and r.fill.isupper()
            if i % 4 == 0:  # (locating a flank in 60 Mbp of Python string takes 50 ms: every fourth gap)
                pos = genome.find(g["left"]) + len(g["left"])
                assert r.fill == genome[pos:pos + g["gap_len"] + k]
        assert tm.retried_gaps == 0 and tm.seg_tier_gaps == 2000
    finally:
        pg.free()


def test_c4_full_size_vs_oracle_k63_60mbp(product, oracle):
    """BASELINE config 4 at its stated size on the branching V3 genome (planted repeats + second
    haplotype: 54.8 M k-mers, 1.75 GB of successor table): the GPU path fills all 2 000 gaps and
    every one of them is compared with the oracle, every field — on all host cores, each gap's oracle run started at
    the product's cumulative draw count (the oracle's own graph of 55 M 128-bit k-mers takes most of this test's
    minute)."""
    k = 63
    reads = product.G2S.synth_genome(60000000, 3, 20240101)
    seqs = _seqs(reads)
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, k, 10, 2000, 200, 1000, 20240103))
    pg = product.Graph.from_seqs(seqs, k, 1)
    og = None
    try:
        assert pg.num_kmers > 54000000
        sess = product.Session(pg, 0, d_err=500, randseed=1)
        res, tm = sess.fill_batch(_gaps(product, gaps), True)
        sess.destroy()
        assert tm.seg_tier_gaps + tm.lds_tier_gaps + tm.retried_gaps >= 2000 and tm.seg_tier_gaps >= 1900 and sum(1 for r in res if r.count > 0) >= 1990
        og = oracle.OracleGraph(seqs, k, 1)
        assert og.num_kmers == pg.num_kmers
        n, nq7 = _compare_with_oracle_in_parallel(product, oracle, og, gaps, res, 500, 1)
        assert n == 2000 - nq7 and n >= 1980
    finally:
        pg.free()
        if og is not None:
            og.free()


def test_c5_full_size_vs_oracle(product, oracle):
    """BASELINE config 5 at its stated size: the 3 Mbp V3 graph, -dist-error 2000, 1 000 gaps of
    2-5 kbp (D = 4-7 k levels), every gap against the oracle.  This list holds the gaps whose
    segments, pending events and right-set entries outgrow the segment tier's LDS-resident
    capacities: they run in its large variant (g2s_fill_segx)."""
    reads = product.G2S.synth_genome(3000000, 3, 20240101)
    seqs = _seqs(reads)
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 1000, 2000, 5000, 20240103))
    assert len(gaps) == 1000
    c, f, tm, xb, sb = _check_batch(product, oracle, seqs, 31, gaps, 2000, seed=1)
    assert c >= 995 and f >= 990
    assert (tm.xB, tm.sB) == (xb, sb)
    # the segment tier holds the whole list: the regular tier finishes 58 % of the gaps it is given — the longest 45 %
    # of a deep list go to the large variant at once (resident mode), the rest of what outgrows the tier behind it
    assert tm.seg_tier_gaps >= 300 and tm.seg_tier_gaps + tm.segx_tier_gaps == 1000
    assert tm.lds_tier_gaps == 0 and tm.retried_gaps == 0 and tm.watchdog_gaps == 0
    # ...and the list stays on the device (resident mode per gap): the gaps that outgrow the regular tier run again in
    # the large variant behind it on the stream, closures the device does not analyse are finished by the host under
    # the trace kernel — nothing is given back to the host path
    assert tm.resident_launches == 1 and tm.resident_fallbacks == 0 and tm.segx_tier_gaps >= 300


def test_c5_full_size_with_phase_d2_on_the_device_vs_oracle(product, oracle, monkeypatch):
    """The same list with the closures the fill kernels do not analyse (412 of the 1 000: more than 192 segments, a k-mer
    at several depths) analysed by g2s_d2_small / g2s_d2_big instead of the host's threads (G2S_DEVICE_D2=1; not the
    default for deep lists: DESIGN 3.6 says what it costs) and traced by the trace kernel from their runs, closures of
    thousands of segments walked where they lie: every gap against the oracle, at most a handful left to the host
    (closures beyond the large instantiation's capacities)."""
    monkeypatch.setenv("G2S_DEVICE_D2", "1")
    reads = product.G2S.synth_genome(3000000, 3, 20240101)
    seqs = _seqs(reads)
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 1000, 2000, 5000, 20240103))
    c, f, tm, xb, sb = _check_batch(product, oracle, seqs, 31, gaps, 2000, seed=1)
    assert c >= 995 and f >= 990
    assert (tm.xB, tm.sB) == (xb, sb)
    assert tm.resident_launches == 1 and tm.resident_fallbacks == 0 and tm.watchdog_gaps == 0
    assert tm.host_finished_gaps <= 10


def test_deep_list_on_a_host_short_of_threads_takes_phase_d2_to_the_device(product, oracle, monkeypatch):
    """The library's own choice (G2S_DEVICE_D2 unset) for a deep list: the host's pool with its default threads, the
    device when the session has fewer than 6 (g2s_params.host_threads = 2 here; the ranks of a launcher and the sessions of
    a team divide the host's CPUs the same way) — profiles/r05_c5_host_threads.txt is the measurement behind the threshold.
    The first 400 gaps of config 5's list, every gap against the oracle both ways."""
    monkeypatch.delenv("G2S_DEVICE_D2", raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    reads = product.G2S.synth_genome(3000000, 3, 20240101)
    seqs = _seqs(reads)
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 1000, 2000, 5000, 20240103))[:400]
    c, f, tm, xb, sb = _check_batch(product, oracle, seqs, 31, gaps, 2000, seed=1, host_threads=2)
    assert c >= 398 and (tm.xB, tm.sB) == (xb, sb)
    assert tm.resident_launches == 1 and tm.resident_fallbacks == 0 and tm.host_finished_gaps <= 5
    c2, f2, tm2, xb2, sb2 = _check_batch(product, oracle, seqs, 31, gaps, 2000, seed=1)
    assert (c2, f2, xb2, sb2) == (c, f, xb, sb)
    assert tm2.resident_launches == 1
    # (the default pool: the CPUs this process may use — affinity mask, container quota —, at most 32; a box that gives
    # the process fewer than 6 takes the device here too)
    cpus = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cpus = min(cpus, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    if cpus >= 8:
        assert tm2.host_finished_gaps >= 50
    elif cpus < 6:
        assert tm2.host_finished_gaps <= 5


def test_c5_on_the_host_path_vs_oracle(product, oracle, monkeypatch):
    """The same list with resident mode off (G2S_RESIDENT=0): the host path of round 2 — closures into pinned host
    memory, analysis while the kernels run, in-order offsets, tracebacks on the pool — stays the fallback."""
    monkeypatch.setenv("G2S_RESIDENT", "0")
    reads = product.G2S.synth_genome(3000000, 3, 20240101)
    seqs = _seqs(reads)
    gaps = _parse_scaffolds(product.G2S.synth_gaps(reads, 31, 10, 1000, 2000, 5000, 20240103))[:400]
    c, f, tm, xb, sb = _check_batch(product, oracle, seqs, 31, gaps, 2000, seed=1)
    assert c >= 398 and (tm.xB, tm.sB) == (xb, sb)
    assert tm.resident_launches == 0 and tm.segx_tier_gaps >= 100 and tm.watchdog_gaps == 0


def _simulated_scaffolds(genome, k, fuz, seed, nrec, rec_len):
    """Scaffold records as an assembler leaves them: consecutive slices of the genome, each with
    1-6 N runs whose lengths are the true gap lengths plus an estimate error, some runs closer
    together than k+fuz (the couplings between consecutive gaps, Gap2Seq.cpp:349,402), some in
    lower case, one next to a record end."""
    rng = cases.SplitMix(seed)
    recs = []
    for r in range(nrec):
        lo = r * rec_len
        s = genome[lo:lo + rec_len]
        pos = rng.randint(k + fuz, 400)
        out, cur = [], 0
        for _ in range(rng.randint(1, 6)):
            true_len = rng.choice([rng.randint(1, 60), rng.randint(100, 700), rng.randint(700, 1500)])
            if pos + true_len + k + fuz + 5 >= len(s):
                break
            est = max(1, true_len + rng.choice([0, 0, 0, 0, -3, 7, -40, 60, -150, 200]))
            out.append(s[cur:pos])
            out.append(("n" if rng.random() < 0.1 else "N") * est)
            cur = pos + true_len
            pos = cur + rng.choice([k + fuz - 1, k + fuz, k + fuz + 1, k + 2 * fuz, rng.randint(200, 3000)])
        out.append(s[cur:] if r % 37 else s[cur:cur + k + 3])  # now and then the record ends right behind a gap
        recs.append(">scaffold%d simulated\n%s\n" % (r, "".join(out)))
    return "".join(recs)


def test_c1_stand_in_multi_gap_scaffolds_through_the_cli(product, oracle, tmp_path):
    """BASELINE config 1's plumbing on the stand-in SURVEY 8(d) names (GAGE S. aureus cannot be
    downloaded): a 2.9 Mbp synthetic genome (V3), ~300 simulated multi-gap scaffolds, reads = the
    two haplotypes, run through gap2seq_amd/Gap2Seq-core and through the oracle's command line
    with the argv the reference wrapper builds (Gap2Seq.py:230-241): FASTA and stdout identical."""
    k, fuz = 31, 10
    reads_text = product.G2S.synth_genome(2900000, 3, 20240105)
    genome = _seqs(reads_text)[0]
    reads = tmp_path / "reads.fa"
    reads.write_text(reads_text)
    scaf = tmp_path / "scaffolds.fa"
    scaf.write_text(_simulated_scaffolds(genome, k, fuz, 7, 300, 9600))
    outs = {}
    for name, exe in (("gpu", os.path.join(ROOT, "gap2seq_amd", "Gap2Seq-core")),
                      ("cpu", os.path.join(os.path.dirname(oracle.ORACLE_SO), "g2s_oracle_cli"))):
        out = tmp_path / ("filled_%s.fa" % name)
        res = subprocess.run([exe, "-k", str(k), "-fuz", str(fuz), "-solid", "1", "-nb-cores", "1", "-dist-error", "500",
                              "-max-mem", "20", "-randseed", "1", "-reads", str(reads), "-filled", str(out),
                              "-scaffolds", str(scaf)], capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        outs[name] = (out.read_text(), res.stdout.replace(str(out), "OUT"), res.stderr)
    assert "# oracle: q7_gaps 0" in outs["cpu"][2], outs["cpu"][2][-300:]
    assert outs["gpu"][1] == outs["cpu"][1]
    assert outs["gpu"][0] == outs["cpu"][0]
    last = outs["gpu"][1].strip().splitlines()[-1].split()  # "Filled X gaps out of Y"
    assert last[0] == "Filled" and int(last[5]) >= 600 and int(last[1]) >= int(last[5]) // 2


def test_c1_wrapper_flow_cut_fill_merge(product, oracle, tmp_path):
    """The reference wrapper's flow without read filtering (Gap2Seq.py:448-454: GapCutter ->
    Gap2Seq-core on the one-gap records -> GapMerger) on the C1 stand-in, with this repository's
    three command lines: (a) multi-gap scaffolds: the merged scaffolds equal those of the same flow
    with the oracle's core in the middle; (b) one gap per scaffold: they also equal what
    `Gap2Seq-core -scaffolds` makes of the uncut scaffolds (with several gaps per record the direct
    run drops the sequence in front of a later filled gap; the cut records do not)."""
    k, fuz = 31, 10
    reads_text = product.G2S.synth_genome(2900000, 3, 20240105)
    genome = _seqs(reads_text)[0]
    reads = tmp_path / "reads.fa"
    reads.write_text(reads_text)
    bins = os.path.join(ROOT, "gap2seq_amd")
    core = {"gpu": os.path.join(bins, "Gap2Seq-core"), "cpu": os.path.join(os.path.dirname(oracle.ORACLE_SO), "g2s_oracle_cli")}

    def run(cmd):
        res = subprocess.run([str(c) for c in cmd], capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        return res

    def flow(which, scaf, tag):
        gaps, contigs, bed = (tmp_path / (tag + n) for n in (".gaps", ".contigs", ".bed"))
        run([os.path.join(bins, "GapCutter"), "-k", k, "-fuz", fuz, "-scaffolds", scaf, "-gaps", gaps, "-contigs", contigs, "-bed", bed])
        filled = tmp_path / (tag + "." + which + ".filled")
        res = run([core[which], "-k", k, "-fuz", fuz, "-solid", 1, "-nb-cores", 1, "-dist-error", 500, "-max-mem", 20,
                   "-randseed", 1, "-reads", reads, "-filled", filled, "-scaffolds", gaps])
        if which == "cpu":
            assert "# oracle: q7_gaps 0" in res.stderr
        merged = tmp_path / (tag + "." + which + ".merged")
        run([os.path.join(bins, "GapMerger"), "-scaffolds", merged, "-gaps", filled, "-contigs", contigs])
        return merged.read_text(), filled.read_text()

    multi = tmp_path / "multi.fa"
    multi.write_text(_simulated_scaffolds(genome, k, fuz, 7, 300, 9600))
    m_gpu, f_gpu = flow("gpu", multi, "multi")
    m_cpu, f_cpu = flow("cpu", multi, "multi")
    assert f_gpu == f_cpu and m_gpu == m_cpu
    assert m_gpu.count(">") == 300 and f_gpu.count(">") >= 600
    # (b) one gap per scaffold, true gap lengths 1-1500 with an estimate error
    rng = cases.SplitMix(99)
    recs = []
    for r in range(200):
        lo = r * 14000
        s = genome[lo:lo + 6000]
        pos = rng.randint(200, 3000)
        true_len = rng.choice([rng.randint(1, 60), rng.randint(100, 700), rng.randint(700, 1500)])
        est = max(1, true_len + rng.choice([0, 0, 0, -3, 7, -40, 60]))
        recs.append(">one%d sim\n%s%s%s\n" % (r, s[:pos], "N" * est, s[pos + true_len:]))
    single = tmp_path / "single.fa"
    single.write_text("".join(recs))
    m_gpu, _ = flow("gpu", single, "single")
    direct = tmp_path / "direct.fa"
    run([core["gpu"], "-k", k, "-fuz", fuz, "-solid", 1, "-nb-cores", 1, "-dist-error", 500, "-max-mem", 20, "-randseed", 1,
         "-reads", reads, "-filled", direct, "-scaffolds", single])
    want = direct.read_text()
    assert m_gpu.replace(">one", ">ONE").count(">ONE") == 200
    # the merged records carry the comment up to the first marker, the direct run the whole comment: same here
    assert m_gpu == want
    assert sum(1 for ln in want.splitlines() if not ln.startswith(">") and "N" not in ln) >= 150
