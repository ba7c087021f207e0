#!/bin/bash
# GPU box: rocprofv3 kernel statistics of the named configs (default C2 C3 C5), a summary per config
O=gpurun_out/${1:-r05ks}; shift; CFGS=${@:-C2 C3 C5}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for C in $CFGS; do
  ST=10; [ $C = C5 ] && ST=4
  B="python3 bench.py --config $C --steps $ST --warmup 2 --no-cpu-baseline --no-c3-beside"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$C -- $B > $O/bench_${C}_under_rocprof.json 2> $O/rp_$C.err
  cp $O/stats_$C/*/*_kernel_stats.csv $O/${C}_kernel_stats.csv; rm -rf $O/stats_$C
  echo "== $C"; python3 tools/kstats.py $O/${C}_kernel_stats.csv | head -16
done
