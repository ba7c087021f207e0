#!/bin/bash
# GPU box: the default bench line and config 2's rocprofv3 kernel statistics alone (the first two steps of tools/measure_r04.sh)
O=gpurun_out/${1:-r04c2}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
python tools/bsum.py C2-full < $O/bench.json
B="python3 bench.py --config C2 --steps 10 --warmup 2 --no-cpu-baseline --no-c3-beside"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- $B > $O/bench_c2_under_rocprof.json 2> $O/rp_c2.err
cp $O/stats_c2/*/*_kernel_stats.csv $O/c2_kernel_stats.csv; rm -rf $O/stats_c2
python3 tools/kstats.py $O/c2_kernel_stats.csv | head -9
