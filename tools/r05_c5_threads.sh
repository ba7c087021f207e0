#!/bin/bash
# GPU box: config 5 by the number of host threads of the session, closures on the host's threads / on the device
O=gpurun_out/${1:-r05c5t}; rm -rf $O; mkdir -p $O
for T in ${2:-2 4 8 16 32}; do for V in 0 1; do
  G2S_DEVICE_D2=$V timeout 600 python bench.py --config C5 --no-cpu-baseline --no-c3-beside --host-threads $T --steps 5 > $O/c5_${T}_$V.json 2> $O/err.txt
  echo -n "host threads $T, phase D2 on the $([ $V = 1 ] && echo device || echo host): "; python tools/bsum.py C5 < $O/c5_${T}_$V.json | cut -c1-48
done; done
