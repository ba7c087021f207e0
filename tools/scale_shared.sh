#!/bin/bash
# One MI355X standing in for N: `bench.py --gpus N --share-device` puts the dispatcher's N sessions
# on device 0 (what a one-GPU box can show of the N>1 path: the group queue, the in-order stage 2,
# the equality with the one-session result — not a scaling curve).
for N in 1 2 4 8; do
  timeout 300 python bench.py --gpus $N --config C3 --share-device --no-cpu-baseline --steps 10 --warmup 2 | python tools/bsum.py N$N
done
