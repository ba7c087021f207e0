#!/bin/bash
# GPU box: resident tests, then config 2 / config 3 / C3 share / config 4 bench lines and config 2's kernel timeline
O=gpurun_out/${1:-r03e}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for r in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-c3-beside | tee -a $O/c2.json | python tools/bsum.py C2; done
timeout 300 python bench.py --config C3 --no-cpu-baseline | tee -a $O/c3.json | python tools/bsum.py C3
timeout 100 python bench.py --no-cpu-baseline --config C3 --gaps 1250 --steps 100 | tee -a $O/c3s.json | python tools/bsum.py C3-1250
timeout 400 python bench.py --no-cpu-baseline --config C4 | tee -a $O/c4.json | python tools/bsum.py C4
bash tools/r03_c2stats.sh ${1:-r03e}k | tail -16
