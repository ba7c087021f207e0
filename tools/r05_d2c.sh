#!/bin/bash
O=gpurun_out/${1:-r05d2c}; rm -rf $O; mkdir -p $O
for V in "A=1" "G2S_D2_NO_CHAINS=1"; do
  echo "== $V"
  env $V G2S_D2_PROF=1 timeout 900 python -m pytest tests/test_gpu_resident.py -q -m gpu -x -s -k "equals_the_host_path and default and False" > $O/pytest1.txt 2>&1; grep "g2s_d2" $O/pytest1.txt | tail -3; tail -3 $O/pytest1.txt
done
