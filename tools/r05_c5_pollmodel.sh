#!/bin/bash
# GPU box: config 5, phase D2 on the device: when the fill kernels list the closures, and what taking them as they are
# listed would leave behind the fill kernels (tools/d2_log.py --poll-model)
O=gpurun_out/${1:-r05c5pm}; rm -rf $O; mkdir -p $O
export G2S_DEVICE_D2=1
G2S_D2_LOG=$O/d2log.txt G2S_D2_PROF=1 timeout 600 python bench.py --config C5 --no-cpu-baseline --no-c3-beside --steps 2 --warmup 1 --prime-seconds 0 > $O/prof.json 2> $O/prof.err
python tools/d2_log.py $O/d2log.txt --top 0 --poll-model 32 16 | grep -A8 "the last list"
python tools/d2_log.py $O/d2log.txt --top 0 --poll-model 64 32 | grep -A3 "the last list" | tail -2
