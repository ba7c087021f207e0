#!/bin/bash
# GPU box: config 5: the tests that run deep gaps in resident mode, bench lines with and without the early hand-over
O=gpurun_out/${1:-r04c5}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "c5" > $O/pytest_c5.txt 2>&1; tail -3 $O/pytest_c5.txt
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q > $O/pytest_res.txt 2>&1; tail -3 $O/pytest_res.txt
for r in 1 2; do timeout 600 python bench.py --config C5 --steps 5 --warmup 1 --no-cpu-baseline | tee -a $O/c5_runs.json | python tools/bsum.py C5; done
G2S_NO_EARLY_HANDOVER=1 timeout 600 python bench.py --config C5 --steps 5 --warmup 1 --no-cpu-baseline | tee -a $O/c5_noearly.json | python tools/bsum.py C5
timeout 400 python bench.py --config C5 --no-cpu-baseline --stream-lists 6 --steps 3 | tee -a $O/c5_stream.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 stream', d['stream_lists'])" | cut -c1-300
G2S_DEBUG=1 timeout 600 python bench.py --config C5 --steps 3 --warmup 1 --prime-seconds 1 --no-cpu-baseline 2> $O/c5_debug.txt > /dev/null; grep -E "resident mode, phase D3" $O/c5_debug.txt | tail -2 | cut -c1-400
