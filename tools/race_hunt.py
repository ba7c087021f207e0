#!/usr/bin/env python3
"""GPU box: one list N times through one session (srand(1) before every call) on one kernel path; prints a digest of
the first call's results (every field and the fill text of every gap) and how many later calls differ from it.
With G2S_LIBRARY pointing at a race-hunting build (gap2seq_amd/_jit: -DG2S_JITTER, gap2seq_amd/_par:
-DG2S_PARANOID_SYNC — csrc/sync_debug.h) the digest has to be the normal build's: tests/test_gpu_resident.py asks.

  python tools/race_hunt.py PATH [N]
     PATH: seg2   300 gaps of config 2's kind: two waves per gap, four-wave trace kernel, closures handed to the host early
           seg    3 500 gaps: one wave per gap, g2s_d2_small behind it, one-wave trace kernel, long-list phase D3
           segw   48 gaps of config 5's kind (-dist-error 2000): the eight-wave kernel, closures on the host's threads
           d2     the same list with G2S_DEVICE_D2=1: g2s_d2_small + g2s_d2_big, closures walked in device memory
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PATHS = {  # genome bp, k, gaps, min len, max len, -dist-error, environment
    "seg2": (3000000, 31, 300, 200, 1000, 500, {}),
    "seg": (3000000, 31, 3500, 200, 1000, 500, {}),
    "segw": (3000000, 31, 48, 2000, 5000, 2000, {"G2S_RESIDENT": "1", "G2S_DEVICE_D2": "0"}),
    "d2": (3000000, 31, 48, 2000, 5000, 2000, {"G2S_RESIDENT": "1", "G2S_DEVICE_D2": "1"}),
}


def main():
    path = sys.argv[1]
    n_runs = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    genome_bp, k, ngaps, min_len, max_len, d_err, env = PATHS[path]
    os.environ.update(env)
    import bench
    from gap2seq_amd import lib as P
    reads = P.G2S.synth_genome(genome_bp, 3, bench.GENOME_SEED)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = bench.parse_gaps(P.G2S.synth_gaps(reads, k, 10, ngaps, min_len, max_len, bench.GAP_SEED), 10)
    graph = P.Graph.from_seqs(seqs, k, 1)
    sess = P.Session(graph, 0, d_err=d_err, randseed=1)
    run = bench.Runner(P, [sess], gaps, 0)

    def digest():
        return hashlib.sha256(repr([bench.result_key(r) for r in run.results()]).encode()).hexdigest()[:20]

    run.step()
    first = digest()
    bad = 0
    for it in range(1, n_runs):
        run.step()
        if digest() != first:
            bad += 1
    tm = run.timing()
    print(json.dumps({"path": path, "library": os.path.basename(os.path.dirname(P.library_path())), "gaps": len(gaps), "runs": n_runs,
                      "differ": bad, "digest": first, "filled": sum(1 for r in run.results() if r.count > 0),
                      "resident_launches": tm.resident_launches, "fallbacks": tm.resident_fallbacks,
                      "host_finished": tm.host_finished_gaps, "traced_in_fill": tm.traced_in_fill_gaps,
                      "segx_tier_gaps": tm.segx_tier_gaps}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
