#!/bin/bash
# GPU box: config 5, six bench runs (spread of the step with the early hand-over)
O=gpurun_out/${1:-r04c5b}; rm -rf $O; mkdir -p $O
for r in 1 2 3 4 5 6; do timeout 600 python bench.py --config C5 --steps 8 --warmup 1 --no-cpu-baseline | tee -a $O/c5_runs.json | python tools/bsum.py C5 | cut -c1-120; done
for r in 1 2; do G2S_NO_EARLY_HANDOVER=1 timeout 600 python bench.py --config C5 --steps 8 --warmup 1 --no-cpu-baseline | tee -a $O/c5_noearly.json | python tools/bsum.py C5-noearly | cut -c1-120; done
