#!/bin/bash
# GPU box: per-gap cycle counts of config 2's launch (host path, so that the records come back): the gaps with the
# longest phase A, the longest phase B + tail, and how the two relate
rm -f /tmp/stats.txt
G2S_RESIDENT=0 G2S_DUMP_STATS=/tmp/stats.txt timeout 200 python bench.py --no-cpu-baseline --no-c3-beside --steps 2 --warmup 1 --prime-seconds 0 > /dev/null 2>&1
python3 - <<'PY'
rows = []
for ln in open("/tmp/stats.txt"):
    if ln.startswith("#"):
        rows = []
        continue
    p = ln.split()
    rows.append(dict(gap=int(p[0]), g=int(p[1]), ar=int(p[3]), ae=int(p[4]), br=int(p[5]), seg=int(p[6]), ca=int(p[7]), cb=int(p[8]), cd=int(p[11])))
print("gaps", len(rows))
print("by phase A cycles:")
for r in sorted(rows, key=lambda r: -r["ca"])[:12]:
    print("  gap %4d g %4d | A rounds %3d entries %3d cycles %7d (%5d per round) | B rounds %3d segments %3d cycles %7d | tail %7d | B+tail %7d" % (
        r["gap"], r["g"], r["ar"], r["ae"], r["ca"], r["ca"] // max(1, r["ar"]), r["br"], r["seg"], r["cb"], r["cd"], r["cb"] + r["cd"]))
print("by B + tail cycles:")
for r in sorted(rows, key=lambda r: -(r["cb"] + r["cd"]))[:12]:
    print("  gap %4d g %4d | A rounds %3d entries %3d cycles %7d | B rounds %3d segments %3d cycles %7d | tail %7d | B+tail %7d" % (
        r["gap"], r["g"], r["ar"], r["ae"], r["ca"], r["br"], r["seg"], r["cb"], r["cd"], r["cb"] + r["cd"]))
tot_a = sum(r["ca"] for r in rows); tot_r = sum(r["ar"] for r in rows); tot_e = sum(r["ae"] for r in rows)
print("phase A: %d rounds, %d entries, %.0f cycles per round" % (tot_r, tot_e, tot_a / max(1, tot_r)))
PY
