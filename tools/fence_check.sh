#!/bin/bash
# GPU box: the bench line (C2, with C3 on one GPU beside it) a few times, then C3 alone — used to compare builds.
OUT=${1:-gpurun_out/fence}
mkdir -p $OUT
for rep in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline < /dev/null > $OUT/c2_$rep.json 2> $OUT/c2_$rep.err
  python tools/bsum.py $OUT/c2_$rep.json < /dev/null | cut -c1-330
done
for rep in 1 2 3; do
  timeout 300 python bench.py --config C3 --steps 40 --warmup 5 --no-cpu-baseline --prime-seconds 0.5 < /dev/null > $OUT/c3_$rep.json 2> $OUT/c3_$rep.err
  python tools/bsum.py $OUT/c3_$rep.json < /dev/null | cut -c1-260
done
