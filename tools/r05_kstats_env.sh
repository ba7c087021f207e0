#!/bin/bash
# GPU box: kernel statistics of a config with environment settings, one line per kernel of interest
# usage: r05_kstats_env.sh OUT CONFIG PATTERN "ENV1" "ENV2" ...
O=gpurun_out/$1; C=$2; PAT=$3; shift 3; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for E in "$@"; do
  ( export $E; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python3 bench.py --config $C --steps 10 --warmup 2 --no-cpu-baseline --no-c3-beside > /dev/null 2> $O/err.txt )
  echo "== $E"; python3 tools/kstats.py $(find $O/s -name "*kernel_stats.csv" | head -1) | grep "$PAT"; rm -rf $O/s
done
