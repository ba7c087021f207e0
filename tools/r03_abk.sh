#!/bin/bash
# GPU box: the fill kernel's duration on config 3 and the step on configs 2 and 3 with two builds of the library
# (gap2seq_amd/_ab/old.so, new.so), interleaved
one() { G2S_KERNEL_TIMING=all G2S_DEBUG=1 timeout 200 python bench.py --no-cpu-baseline --config C3 --steps 8 --warmup 3 2>&1 > /dev/null | grep "resident mode: 10000" | tail -5 | sed 's/.*fill kernel \([0-9.]*\) ms.*/\1/' | tr '\n' ' '; echo " $1"; }
for rep in 1 2 3; do
  for v in old new; do
    cp gap2seq_amd/_ab/$v.so gap2seq_amd/libg2s_hip.so; one $v
    timeout 200 python bench.py --no-cpu-baseline --config C3 < /dev/null | python tools/bsum.py $v-C3 | cut -c1-70
    timeout 200 python bench.py --no-cpu-baseline --no-c3-beside < /dev/null | python tools/bsum.py $v-C2 | cut -c1-70
  done
done
cp gap2seq_amd/_ab/new.so gap2seq_amd/libg2s_hip.so
