"""tools/gpu_check.py — quick GPU-vs-oracle comparison used while developing
(the real parity tests are tests/test_gpu_parity.py).  Usage: python tools/gpu_check.py [quick|c2]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
import oracle_lib as O  # noqa: E402
from gap2seq_amd import lib as P  # noqa: E402


def compare_batch(tag, seqs, k, gaps, e, skip=False, allp=True, verbose=False):
    og = O.OracleGraph(seqs, k, 1)
    pg = P.Graph.from_seqs(seqs, k, 1)
    sess = P.Session(pg, 0, d_err=e, skip_confident=skip, all_paths=allp, randseed=5)
    res, tm = sess.fill_batch([P.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gaps], True)
    rng = O.OracleRng(5)
    bad = 0
    nq7 = nfill = 0
    xb = sb = 0
    for i, (g, r) in enumerate(zip(gaps, res)):
        o = O.fill_gap(og, rng, g["left"], g["right"], g["gap_len"], e, g["lmf"], g["rmf"], skip, allp)
        if o.info.q7 or (r.flags & P.G2S_GAP_Q7):
            nq7 += 1
            if o.info.q7 and not (r.flags & P.G2S_GAP_Q7):
                print(tag, i, "oracle q7 but GPU not flagged")
                bad += 1
            # resync the oracle stream is impossible when draws differ; stop comparing after divergence
            if o.info.draws != r.draws:
                print(tag, "q7 gap with different draw count; stopping batch compare at", i)
                break
            continue
        xb += o.info.ctr[2]
        sb += o.info.ctr[3]
        ok = (r.count == o.count and r.draws == o.info.draws and r.phaseC_count == o.info.phaseC_count
              and r.lengths == o.lengths)
        if o.phase_d:
            ok = ok and (r.left_fuz, r.right_fuz) == (o.left_fuz, o.right_fuz) and r.fill == o.fill
            if not skip:
                ok = ok and r.substats == o.substats
        if not ok:
            bad += 1
            if bad < 6 or verbose:
                print(tag, "MISMATCH gap", i, g["gap_len"], g["lmf"], g["rmf"], "gpu", r.count, r.phaseC_count,
                      r.lengths, r.draws, r.left_fuz, r.right_fuz, "flags", hex(r.flags), "oracle", o.count,
                      o.info.phaseC_count, o.lengths, o.info.draws, o.left_fuz, o.right_fuz)
                if o.phase_d and r.fill != o.fill:
                    print("   fill gpu", r.fill[:80], "\n   fill ora", o.fill[:80])
        if o.count > 0:
            nfill += 1
    devx, devs = tm.xB, tm.sB
    print("%s: gaps %d filled %d q7 %d mismatches %d | kernels A %.3f ms B %.3f ms d2h %.3f ms post %.3f ms total %.3f ms"
          " | xB dev %d oracle(non-q7) %d sB dev %d oracle %d retried %d" %
          (tag, len(gaps), nfill, nq7, bad, tm.ms_right_bfs, tm.ms_left_dp, tm.ms_d2h, tm.ms_host_post, tm.ms_total,
           devx, xb, devs, sb, tm.retried_gaps))
    sess.destroy()
    pg.free()
    og.free()
    return bad


def quick():
    bad = 0
    for seed in range(12):
        k = [5, 7, 9, 11, 13, 15][seed % 6]
        seqs = cases.toy_genome(seed, 900, k, repeats=seed % 4, tandem=seed % 3, inverted=(1 if seed % 5 == 0 else 0),
                                snp_every=(0 if seed % 2 else 83))
        e = [0, 4, 9, 20, 31][seed % 5] + k
        gaps = cases.cut_gaps(seed, seqs[0], k, fuz=seed % 5 + 1, ngaps=40, min_len=1, max_len=60, d_err=e)
        for (skip, allp) in [(False, True), (False, False), (True, True)]:
            bad += compare_batch("toy s%d k%d skip%d all%d" % (seed, k, skip, allp), seqs, k, gaps, e, skip, allp)
    # k=31 mid-size with repeats and bubbles, default parameters
    for variant in (0, 1, 2, 3):
        reads = P.G2S.synth_genome(200000, variant, 20240101)
        seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
        scaf = P.G2S.synth_gaps(reads, 31, 10, 60, 50, 400, 20240103)
        gaps = []
        lines = scaf.splitlines()
        for j in range(0, len(lines), 2):
            s = lines[j + 1]
            a = s.index("N")
            b = len(s) - s[::-1].index("N")
            gaps.append(dict(left=s[:a], right=s[b:], gap_len=b - a, lmf=10, rmf=10))
        bad += compare_batch("k31 V%d" % variant, seqs, 31, gaps, 500)
    print("TOTAL MISMATCHES", bad)
    return bad


def c2(ngaps=500, variant=3, length=3000000):
    t0 = time.time()
    reads = P.G2S.synth_genome(length, variant, 20240101)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    scaf = P.G2S.synth_gaps(reads, 31, 10, ngaps, 200, 1000, 20240103)
    gaps = []
    lines = scaf.splitlines()
    for j in range(0, len(lines), 2):
        s = lines[j + 1]
        a = s.index("N")
        b = len(s) - s[::-1].index("N")
        gaps.append(dict(left=s[:a], right=s[b:], gap_len=b - a, lmf=10, rmf=10))
    print("synth %.1fs" % (time.time() - t0))
    return compare_batch("C2 V%d n%d" % (variant, ngaps), seqs, 31, gaps, 500)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
    print("devices", P.G2S.device_count())
    rc = quick() if mode == "quick" else c2()
    sys.exit(1 if rc else 0)
