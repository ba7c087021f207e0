#!/bin/bash
# GPU box: lap timers (G2S_DEBUG) of config 2 in resident mode and of config 5
O=gpurun_out/${1:-r03d}; rm -rf $O; mkdir -p $O
G2S_DEBUG=1 timeout 300 python bench.py --no-cpu-baseline --no-c3-beside --steps 6 --warmup 3 > $O/c2.json 2> $O/c2.err; tail -40 $O/c2.err
G2S_DEBUG=1 timeout 400 python bench.py --no-cpu-baseline --config C5 --steps 4 --warmup 2 > $O/c5.json 2> $O/c5.err; tail -60 $O/c5.err
