#!/bin/bash
# GPU box: one build, a setting on and off, interleaved on one box (the pool's boxes differ by 5-10 %).
# Usage: tools/ab_env.sh REPS "ENV=1 ENV2=x" -- bench.py arguments      (first leg: with the settings; second: without)
R=$1; E=$2; shift 2; [ "$1" = "--" ] && shift
for rep in $(seq 1 $R); do
  env $E timeout 200 python bench.py --no-cpu-baseline "$@" < /dev/null | python tools/bsum.py "with[$E]" | cut -c1-210
  timeout 200 python bench.py --no-cpu-baseline "$@" < /dev/null | python tools/bsum.py "default" | cut -c1-210
done
