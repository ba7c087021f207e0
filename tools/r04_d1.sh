#!/bin/bash
# GPU box: phase D1's sweep by chunks: config 5 lines, parity of the large variant (full size, toy graphs, a fuzz leg)
O=gpurun_out/${1:-r04d1}; rm -rf $O; mkdir -p $O
for r in 1 2; do timeout 600 python bench.py --config C5 --steps 5 --warmup 1 --no-cpu-baseline | tee -a $O/c5_runs.json | python tools/bsum.py C5; done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "c5" > $O/pytest_c5.txt 2>&1; tail -3 $O/pytest_c5.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_resident.py -x -q > $O/pytest_par.txt 2>&1; tail -3 $O/pytest_par.txt
echo "## every gap through g2s_fill_segw (G2S_FORCE_SEGX=1)" | tee -a $O/fuzz.txt
G2S_FORCE_SEGX=1 timeout 400 python tools/fuzz_parity.py --seconds 150 --seed ${2:-900} --big 0.3 --scaffold 0.2 2>&1 | tail -3 | tee -a $O/fuzz.txt
for r in 1 2; do timeout 300 python bench.py --no-cpu-baseline | tee -a $O/c2.json | python tools/bsum.py C2; done
