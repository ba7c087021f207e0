"""Summarise a G2S_DUMP_STATS file (one line per gap of an LDS-tier launch): where the wave-cycles
of a launch go, by phase, and how the per-gap totals are distributed."""
import sys

rows = []
seg = False
for ln in open(sys.argv[1]):
    if ln.startswith("#"):
        rows = []  # keep the last block only
        seg = "segment tier" in ln
        continue
    p = ln.split()
    rows.append([int(p[0]), int(p[1]), int(p[2], 16)] + [int(x) for x in p[3:]])
n = len(rows)
cyc = sorted(r[7] + r[8] + r[11] for r in rows)
tot = sum(cyc)
A = sum(r[7] for r in rows); B = sum(r[8] for r in rows); D = sum(r[11] for r in rows)
print("gaps %d | wave-cycles total %.1f M: A %.1f%% B %.1f%% D1 %.1f%%" % (n, tot / 1e6, 100 * A / tot, 100 * B / tot, 100 * D / tot))
print("per gap: mean %.0f k, median %.0f k, p90 %.0f k, p99 %.0f k, max %.0f k cycles" % (
    tot / n / 1e3, cyc[n // 2] / 1e3, cyc[int(n * 0.9)] / 1e3, cyc[int(n * 0.99)] / 1e3, cyc[-1] / 1e3))
sb = sum(r[5] for r in rows); bb = sum(r[6] for r in rows); sd = sum(r[9] for r in rows); bd = sum(r[10] for r in rows)
sa = sum(r[3] for r in rows)
print("steps: A groups/rounds %d (%.0f cycles each) | B per-level/rounds %d bulk/segments %d (%.0f cycles per step or round) | D1 per-level %d bulk %d (%.0f cycles per step; segment tier: per gap %.0f)" % (
    sa, A / max(1, sa), sb, bb, B / max(1, sb + (0 if seg else bb)), sd, bd, D / max(1, sd + bd), D / max(1, n)))
print("states B %d, closure %d, right set %d" % (sum(r[14] for r in rows), sum(r[16] for r in rows), sum(r[12] for r in rows)))
pool = sum(1 for r in rows if r[2] & 0x2000); lp = sum(1 for r in rows if r[2] & 0x1000)
print("right set in the spill pool: %d gaps, log pool: %d gaps" % (pool, lp))
