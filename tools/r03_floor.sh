#!/bin/bash
# GPU box: the latency floor (tools/lat_floor.hip) and per-gap statistics of configs 2, 3/8 and 3 on the host path
O=gpurun_out/${1:-r03floor}; rm -rf $O; mkdir -p $O
timeout 120 tools/lat_floor.bin > $O/lat_floor.txt 2>&1; cat $O/lat_floor.txt
export G2S_RESIDENT=0
G2S_DUMP_STATS=$O/st_c2.txt python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-c3-beside --prime-seconds 0 > /dev/null 2>&1
G2S_DUMP_STATS=$O/st_c3_8.txt python bench.py --config C3 --gaps 1250 --steps 1 --warmup 0 --no-cpu-baseline --prime-seconds 0 > /dev/null 2>&1
G2S_DUMP_STATS=$O/st_c3.txt python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline --prime-seconds 0 > /dev/null 2>&1
# (the dump holds one block per step of the run: first, second, ... call; latency_floor.py keeps the last)
python tools/latency_floor.py $O/lat_floor.txt "config 2 (500 gaps)":$O/st_c2.txt:42934763:2 "config 3 share of one GPU of 8 (1250 gaps)":$O/st_c3_8.txt:104600000:2 "config 3 on one GPU (10 000 gaps)":$O/st_c3.txt:848500000:1 | tee $O/latency_floor.txt
python tools/gapstats.py $O/st_c3.txt | tee $O/gapstats_c3.txt
