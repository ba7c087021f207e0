#!/bin/bash
# GPU box: round 4's build (a worktree of its last commit at _r04/, built there) against this tree, alternating, one box
for R in 1 2; do for W in "--no-c3-beside" "--config C3" "--config C4" "--config C5 --steps 5" "--no-c3-beside --variant 0" "--no-c3-beside --variant 1" "--config C3 --gaps 1250 --steps 100"; do
  for T in _r04 .; do
    echo -n "$T [$W]: "; (cd $T && timeout 400 python bench.py --no-cpu-baseline $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  done
done; done
