#!/usr/bin/env python3
"""tools/oracle_units.py — the unit counts SURVEY.md §8(d) prices a launch with, counted by the CPU ORACLE.

    python tools/oracle_units.py C2 C3 C3/8 C2:V0 C2:V1 C2:V2 C5 [C4]     (no GPU needed)

§8(d): algorithmic bytes = 24 X + 8 S + per-gap I/O, where X = expansions and S = newly set states of the
REFERENCE algorithm's phases A, B and D1 (Gap2Seq.cpp:912,930,1031,1049-1065,1266-1301) "counted by the CPU
oracle".  The product's kernels search over unitig segments and do not perform those expansions one by one: their
own counters are estimates (the phase A one is an upper bound) and must not price a roofline.  This script runs the
oracle (oracle/g2s_oracle.cpp, all host cores) over a whole bench workload and records the six counts in
profiles/oracle_units.json under the key bench.py builds from the workload's parameters (bench.units_key), so that
`bench.py --config …` can quote a fraction for lists the oracle would take too long to count inside a bench run.
The workload generator is the library's own (g2s_synth_genome / g2s_synth_gaps: host code, seeded).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402  (units_key, the table's path and the workload definitions live there)


def count_units(seqs, k, gaps, d_err, nthreads=0):
    """The oracle over the whole list: dict(xA, sA, xB, sB, xD, sD, filled, seconds, threads)."""
    import oracle_lib as O
    og = O.OracleGraph(seqs, k, 1)
    nthreads = nthreads or (os.cpu_count() or 1)
    secs, filled, c = O.time_fill_batch(og, gaps, d_err, nthreads)
    og.free()
    return dict(xA=c[0], sA=c[1], xB=c[2], sB=c[3], xD=c[4], sD=c[5], filled=filled, seconds=round(secs, 3),
                threads=nthreads)


def main(argv):
    from gap2seq_amd import lib as P
    units = bench.load_oracle_units()
    for spec in argv:
        name, _, var = spec.partition(":")
        variant = int(var[1:]) if var else 3
        name, _, div = name.partition("/")
        genome_bp, k, ngaps, min_len, max_len, d_err, _ = bench.CONFIGS[name]
        if div:
            ngaps //= int(div)
        fuz = 10
        key = bench.units_key(genome_bp, variant, k, ngaps, min_len, max_len, fuz, d_err)
        t0 = time.time()
        reads = P.G2S.synth_genome(genome_bp, variant, bench.GENOME_SEED)
        seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
        gaps = bench.parse_gaps(P.G2S.synth_gaps(reads, k, fuz, ngaps, min_len, max_len, bench.GAP_SEED), fuz)
        u = count_units(seqs, k, gaps, d_err)
        u["gaps"] = len(gaps)
        u["spec"] = spec
        units[key] = u
        print("%-8s %s  X=%d S=%d filled %d/%d  (oracle %.1f s on %d threads, %.1f s in all)" % (
            spec, key, u["xA"] + u["xB"] + u["xD"], u["sA"] + u["sB"] + u["sD"], u["filled"], len(gaps), u["seconds"],
            u["threads"], time.time() - t0), flush=True)
        with open(bench.UNITS_FILE, "w") as f:
            json.dump(units, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main(sys.argv[1:] or ["C2", "C3", "C3/8", "C2:V0", "C2:V1", "C2:V2", "C5"])
