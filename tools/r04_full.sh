#!/bin/bash
# GPU box: the whole GPU suite, then the bench lines
O=gpurun_out/${1:-r04full}; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 600 python bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline | tee $O/c5.json | python tools/bsum.py C5
for r in 1 2; do timeout 300 python bench.py --config C3 --no-cpu-baseline | tee -a $O/c3.json | python tools/bsum.py C3; done
for r in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-c3-beside | tee -a $O/c2.json | python tools/bsum.py C2; done
timeout 300 python bench.py --config C4 --no-cpu-baseline | tee $O/c4.json | python tools/bsum.py C4
