#!/bin/bash
# GPU box: the whole GPU suite, then the default bench line
O=gpurun_out/${1:-r04all}; rm -rf $O; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench_err.txt; python tools/bsum.py C2 < $O/bench.json
