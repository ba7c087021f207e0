#!/bin/bash
# GPU box: step and kernel time against the batch size of the announcements (G2S_PUBLISH_BATCH; 1 = every gap by itself).
for rep in 1 2; do
  for b in 1 2 4 8 16; do
    G2S_PUBLISH_BATCH=$b timeout 100 python bench.py --no-cpu-baseline --no-c3-beside < /dev/null | python tools/bsum.py C2-b$b | cut -c1-150
  done
  for b in 1 4 16; do
    G2S_PUBLISH_BATCH=$b timeout 100 python bench.py --no-cpu-baseline --no-c3-beside --variant 0 < /dev/null | python tools/bsum.py V0-b$b | cut -c1-150
    G2S_PUBLISH_BATCH=$b timeout 100 python bench.py --config C4 --no-cpu-baseline < /dev/null | python tools/bsum.py C4-b$b | cut -c1-150
  done
  for b in 1 4 8 16 32 64; do
    G2S_PUBLISH_BATCH=$b timeout 100 python bench.py --config C3 --steps 30 --warmup 5 --no-cpu-baseline < /dev/null | python tools/bsum.py C3-b$b | cut -c1-150
  done
done
