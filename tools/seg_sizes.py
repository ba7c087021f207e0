"""How large do the segment tier's tables have to be?  Runs the executable model (tests/seg_model.py) on a
sample of a bench configuration's gaps, on the CPU, and prints per gap: right-set entries, peak pending
events, segments, rounds.  usage: python tools/seg_sizes.py C5 [first_gap] [count]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
import seg_model as M  # noqa: E402
from gap2seq_amd import lib as P  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 20
genome_bp, k, ngaps, min_len, max_len, d_err, _ = bench.CONFIGS[cfg]
reads = P.G2S.synth_genome(genome_bp, 3, 20240101)
seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
gaps = bench.parse_gaps(P.G2S.synth_gaps(reads, k, 10, ngaps, min_len, max_len, 20240103), 10)
pg = P.Graph.from_seqs(seqs, k, 1)
tb = M.Tables(P, pg)
print("# gap g entries peak_pending peak_batch segments rounds seconds")
for gi in range(first, min(len(gaps), first + count)):
    g = gaps[gi]
    lmf, rmf, left, right = g["lmf"], g["rmf"], g["left"], g["right"]
    lseeds = [pg.node(left[d:d + k]) for d in range(lmf + 1)]
    rseeds = [pg.node(right[len(right) - k - d:len(right) - d]) for d in range(rmf + 1)]
    targets = [pg.node(right[d:d + k]) for d in range(rmf + 1)]
    mg = M.Gap(g["gap_len"], d_err, lmf, rmf, lseeds, rseeds, targets)
    t0 = time.time()
    lab = M.right_entries(tb, mg)
    rs = M.entries_to_set(tb, mg, lab)
    m = M.fill_model(tb, mg, rs=rs)
    print(gi, g["gap_len"], len(lab), m.max_pending, m.max_batch, m.n_seg, m.rounds, "%.1f" % (time.time() - t0), flush=True)
