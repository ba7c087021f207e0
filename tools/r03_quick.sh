#!/bin/bash
# GPU box: resident tests, then the bench lines of configs 3 and 2 (three runs each)
O=gpurun_out/${1:-r03q}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for r in 1 2 3; do timeout 300 python bench.py --config C3 --no-cpu-baseline | tee -a $O/c3.json | python tools/bsum.py C3; done
for r in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-c3-beside | tee -a $O/c2.json | python tools/bsum.py C2; done
