#!/bin/bash
# GPU box: phase D1's sweep by chunks in the regular tier: bench lines, the whole GPU suite, a fuzz leg
O=gpurun_out/${1:-r04d1b}; rm -rf $O; mkdir -p $O
for r in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline | tee -a $O/c2.json | python tools/bsum.py C2; done
timeout 300 python bench.py --config C3 --no-cpu-baseline | tee -a $O/c3.json | python tools/bsum.py C3
timeout 300 python bench.py --config C4 --no-cpu-baseline | tee -a $O/c4.json | python tools/bsum.py C4
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
echo "## default" | tee -a $O/fuzz.txt
timeout 400 python tools/fuzz_parity.py --seconds 150 --seed ${2:-950} --big 0.5 --scaffold 0.2 2>&1 | tail -2 | tee -a $O/fuzz.txt
echo "## resident forced" | tee -a $O/fuzz.txt
G2S_RESIDENT=1 timeout 400 python tools/fuzz_parity.py --seconds 150 --seed ${2:-951} --big 0.4 --scaffold 0.3 2>&1 | tail -2 | tee -a $O/fuzz.txt
