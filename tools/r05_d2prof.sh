#!/bin/bash
# GPU box: where g2s_d2_* spends its time (G2S_D2_PROF), per config
O=gpurun_out/${1:-r05d2prof}; rm -rf $O; mkdir -p $O
for C in C2 C3 C5; do
  G2S_DEVICE_D2=1 G2S_D2_PROF=1 timeout 600 python bench.py --config $C --no-cpu-baseline --no-c3-beside --steps 4 --warmup 1 --prime-seconds 0 > $O/$C.json 2> $O/$C.err
  echo "== $C"; grep "g2s_d2" $O/$C.err | tail -3
done
