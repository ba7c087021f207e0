#!/bin/bash
# GPU box: two builds of the library against each other, interleaved, on one box.  Before the call: build each and copy
# its gap2seq_amd/libg2s_hip.so to gap2seq_amd/_ab/old.so / new.so (git-ignored; they travel with the snapshot).
for rep in 1 2 3 4 5 6; do
  for v in old new; do
    cp gap2seq_amd/_ab/$v.so gap2seq_amd/libg2s_hip.so
    timeout 200 python bench.py --no-cpu-baseline "$@" < /dev/null | python tools/bsum.py $v | cut -c1-330
  done
done
cp gap2seq_amd/_ab/new.so gap2seq_amd/libg2s_hip.so
