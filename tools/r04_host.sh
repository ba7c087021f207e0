#!/bin/bash
# GPU box: config 5 in resident mode with G2S_DEBUG: what the host's share behind the hand-over is made of;
# the handed-over closures into gpurun_out (tools/host_items_replay.py replays them without a GPU)
O=gpurun_out/${1:-r04host}; rm -rf $O; mkdir -p $O
G2S_HOST_ITEMS_DUMP=$O/items_c5.bin G2S_DEBUG=1 timeout 600 python bench.py --config C5 --steps 3 --warmup 1 --prime-seconds 1 --no-cpu-baseline > $O/c5.json 2> $O/c5_debug.txt; python tools/bsum.py C5 < $O/c5.json
grep -E "host-finished|run analysis|resident mode, phase D3" $O/c5_debug.txt | tail -12 | cut -c1-700
for r in 1 2; do timeout 600 python bench.py --config C5 --steps 5 --warmup 1 --no-cpu-baseline | tee -a $O/c5_runs.json | python tools/bsum.py C5; done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "c5" > $O/pytest_c5.txt 2>&1; tail -3 $O/pytest_c5.txt
timeout 1200 python -m pytest tests/test_gpu_resident.py -x -q > $O/pytest_res.txt 2>&1; tail -3 $O/pytest_res.txt
