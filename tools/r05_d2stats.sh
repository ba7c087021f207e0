#!/bin/bash
# GPU box: the shape of the graphs phase D2 works on (post.cpp, G2S_D2_STATS), one line per host-analysed closure
O=gpurun_out/${1:-r05d2stats}; rm -rf $O; mkdir -p $O
for C in C5 C3; do
  rm -f /tmp/d2stats.txt
  G2S_D2_STATS=/tmp/d2stats.txt timeout 600 python bench.py --config $C --no-cpu-baseline --steps 1 --warmup 0 --prime-seconds 0 > $O/$C.json 2> $O/$C.err
  sort /tmp/d2stats.txt | uniq > $O/d2stats_$C.txt; wc -l $O/d2stats_$C.txt
done
python - <<'PY'
import sys,glob
for f in sorted(glob.glob(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/*/d2stats_*.txt')):
    rows=[dict(zip(l.split()[::2], map(int, l.split()[1::2]))) for l in open(f)]
    if not rows: continue
    print(f, len(rows))
    for key in ('segs','runs','edges','comps','nontrivial','size_nontrivial','levels','branch_comps','branch_levels'):
        v=sorted(r[key] for r in rows); print('  %-16s min %d median %d p90 %d max %d' % (key, v[0], v[len(v)//2], v[len(v)*9//10], v[-1]))
PY
