#!/bin/bash
# GPU box: two builds of the library against each other, interleaved, on one box.  Before the call: build each and copy
# its gap2seq_amd/libg2s_hip.so to gap2seq_amd/_ab/old.so / new.so (git-ignored; they travel with the snapshot).
# Usage: tools/ab.sh [repetitions] -- bench.py arguments
R=4; if [ "$1" != "--" ] && [ -n "$1" ]; then R=$1; shift; fi; [ "$1" = "--" ] && shift
for rep in $(seq 1 $R); do
  for v in old new; do
    cp gap2seq_amd/_ab/$v.so gap2seq_amd/libg2s_hip.so
    timeout 200 python bench.py --no-cpu-baseline "$@" < /dev/null | python tools/bsum.py $v | cut -c1-150
  done
done
cp gap2seq_amd/_ab/new.so gap2seq_amd/libg2s_hip.so
