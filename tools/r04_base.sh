#!/bin/bash
# GPU box: where round 3's build stands on config 5 (debug lines of the slow gaps, bench line), and C2/C3 lines
O=gpurun_out/${1:-r04base}; rm -rf $O; mkdir -p $O
G2S_DEBUG=1 timeout 600 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $O/c5.json 2> $O/c5_debug.txt; python tools/bsum.py C5 < $O/c5.json
grep -E "slow gap|analysis|lap|run_tier" $O/c5_debug.txt | tail -40
timeout 300 python bench.py --config C3 --no-cpu-baseline | tee $O/c3.json | python tools/bsum.py C3
timeout 300 python bench.py --no-cpu-baseline --no-c3-beside | tee $O/c2.json | python tools/bsum.py C2
