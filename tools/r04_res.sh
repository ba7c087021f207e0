#!/bin/bash
# GPU box: resident mode per gap: the new tests, the resident suite, config 5 in resident mode, C2/C3 lines
O=gpurun_out/${1:-r04res}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q -k "deep_gap or long_short_long" > $O/pytest_new.txt 2>&1; tail -15 $O/pytest_new.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "c5" > $O/pytest_c5.txt 2>&1; tail -15 $O/pytest_c5.txt
timeout 1200 python -m pytest tests/test_gpu_resident.py -x -q > $O/pytest_res.txt 2>&1; tail -3 $O/pytest_res.txt
G2S_DEBUG=1 timeout 600 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $O/c5.json 2> $O/c5_debug.txt; python tools/bsum.py C5 < $O/c5.json
grep -E "resident mode|slow gap" $O/c5_debug.txt | tail -6 | cut -c1-600
for r in 1 2; do timeout 300 python bench.py --config C3 --no-cpu-baseline | tee -a $O/c3.json | python tools/bsum.py C3; done
for r in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-c3-beside | tee -a $O/c2.json | python tools/bsum.py C2; done
