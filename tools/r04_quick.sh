#!/bin/bash
# GPU box: a quick look at a build: the parity tests of the kernels, then bench lines C2 (x3), C3, C4, C5
O=gpurun_out/${1:-r04quick}; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_resident.py -x -q > $O/pytest_par.txt 2>&1; tail -3 $O/pytest_par.txt
for r in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline | tee -a $O/c2.json | python tools/bsum.py C2; done
timeout 300 python bench.py --config C3 --no-cpu-baseline | tee -a $O/c3.json | python tools/bsum.py C3
timeout 300 python bench.py --config C4 --no-cpu-baseline | tee -a $O/c4.json | python tools/bsum.py C4
timeout 300 python bench.py --config C5 --no-cpu-baseline --steps 5 | tee -a $O/c5.json | python tools/bsum.py C5
