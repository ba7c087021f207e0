#!/bin/bash
# GPU box: per-kernel times of config 3's list (rocprofv3 --kernel-trace --stats; program directly after --)
O=gpurun_out/${1:-r03f}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CFG=${2:-C3}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --config $CFG --steps 10 --warmup 2 --no-cpu-baseline --no-c3-beside > $O/bench.json 2> $O/rp.err
cat $O/stats/*/*_kernel_stats.csv | sed 's/"\([a-zA-Z0-9_]*\)([^"]*"/\1/; s/"_ZN[0-9a-zA-Z_]*"/&/' | cut -c1-240 | head -30
cp $O/stats/*/*_kernel_stats.csv $O/kernel_stats.csv
