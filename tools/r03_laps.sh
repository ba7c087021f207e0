#!/bin/bash
# GPU box: phase D3's lap stamps on config 2 and config 3
O=gpurun_out/${1:-r03l}; rm -rf $O; mkdir -p $O
G2S_DEBUG=1 timeout 300 python bench.py --no-cpu-baseline --no-c3-beside --steps 6 --warmup 3 > $O/c2.json 2> $O/c2.err; grep "laps" $O/c2.err | tail -4; grep "resident mode" $O/c2.err | tail -2
G2S_DEBUG=1 timeout 300 python bench.py --no-cpu-baseline --config C3 --steps 4 --warmup 2 > $O/c3.json 2> $O/c3.err; grep "laps" $O/c3.err | tail -2; grep "resident mode" $O/c3.err | tail -2
