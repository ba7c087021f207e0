#!/bin/bash
# GPU box: per-step preparation times of config 3 (G2S_DEBUG lap lines), two builds interleaved
one() { G2S_DEBUG=1 timeout 200 python bench.py --no-cpu-baseline --config C3 --steps 40 --warmup 3 2>&1 > /dev/null | grep "fill_batch: prepare" | tail -40 | sed 's/.*prepare \([0-9.]*\) ms.*/\1/' | tr '\n' ' '; echo " $1"; }
for rep in 1 2 3; do
  for v in old new; do cp gap2seq_amd/_ab/$v.so gap2seq_amd/libg2s_hip.so; one $v; done
done
cp gap2seq_amd/_ab/new.so gap2seq_amd/libg2s_hip.so
