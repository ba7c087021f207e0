#!/bin/bash
# GPU box: the whole GPU suite (every failure listed), then the default bench line
O=gpurun_out/${1:-r05all}; rm -rf $O; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench_err.txt; python tools/bsum.py C2 < $O/bench.json
