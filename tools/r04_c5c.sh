#!/bin/bash
# GPU box: config 5 with the early launch of the large variant: the full-size test repeated, bench lines
O=gpurun_out/${1:-r04c5c}; rm -rf $O; mkdir -p $O
for r in 1 2 3 4 5; do timeout 400 python -m pytest tests/test_gpu_fullsize.py -x -q -k "c5_full_size" > $O/t$r.txt 2>&1; grep -E "passed|failed" $O/t$r.txt | tail -1; grep -E "^E " $O/t$r.txt | head -5; done
for r in 1 2 3 4 5 6; do timeout 600 python bench.py --config C5 --steps 8 --warmup 1 --no-cpu-baseline | tee -a $O/c5_runs.json | python tools/bsum.py C5 | cut -c1-110; done
G2S_NO_EARLY_SEGW=1 timeout 600 python bench.py --config C5 --steps 8 --warmup 1 --no-cpu-baseline | tee -a $O/c5_noearly.json | python tools/bsum.py C5-no-early-launch | cut -c1-110
for r in 1 2; do timeout 400 python bench.py --config C5 --no-cpu-baseline --stream-lists 6 --steps 3 | tee -a $O/c5_stream.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 stream', d['stream_lists']['value'], d['stream_lists']['ms_per_list'])"; done
