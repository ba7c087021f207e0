#!/bin/bash
# GPU box: kernel statistics of a config on round 4's build (_r04/) and on this tree
O=gpurun_out/${1:-r05ksv}; C=${2:-C4}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
R=$PWD
for T in _r04 .; do
  cd $R/$T
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/s -- python3 bench.py --config $C --steps 10 --warmup 2 --no-cpu-baseline --no-c3-beside > /dev/null 2> $R/$O/err.txt
  cd $R
  echo "== $T $C"; python3 tools/kstats.py $(find $O/s -name "*kernel_stats.csv" | head -1) | head -10; rm -rf $O/s
done
