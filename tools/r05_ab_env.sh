#!/bin/bash
# GPU box: a config with and without an environment switch, alternating (A B A B ...): gaps/s and ms per step
# usage: r05_ab_env.sh OUT CONFIG VAR=VALUE [rounds]
O=gpurun_out/${1:-r05ab}; C=${2:-C3}; KV=$3; R=${4:-4}; rm -rf $O; mkdir -p $O
for i in $(seq 1 $R); do
  for V in base switch; do
    if [ $V = switch ]; then export "$KV"; else unset ${KV%%=*}; fi
    timeout 600 python bench.py --config $C --no-cpu-baseline --no-c3-beside > $O/${V}_$i.json 2> $O/err.txt
    echo -n "$V $KV: "; python tools/bsum.py $C < $O/${V}_$i.json | cut -c1-110
  done
done
