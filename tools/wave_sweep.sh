#!/bin/bash
# GPU box: kernel time of the segment tier with one and with two waves per gap over list lengths (config 3's graph);
# where the two-wave kernel stops paying is the threshold run_tier uses.
for rep in 1 2; do
  for n in 1250 2500 3500 5000 7000 10000; do
    for w in 1 2; do
      G2S_SEG_WAVES=$w timeout 100 python bench.py --config C3 --gaps $n --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null < /dev/null |
        W=$w python -c 'import sys, json, os; d = json.loads(sys.stdin.readline()); print("gaps", d["config"]["gaps"], "waves", os.environ["W"], "kernel ms", d["roofline"]["kernel_ms_per_launch"], "step ms", d["ms_per_step"])'
    done
  done
done
