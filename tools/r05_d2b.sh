#!/bin/bash
# GPU box: phase D2 on the device — parity first (resident tests, every closure through either instantiation), then
# where the kernel spends its time and what the kernels of a step take
O=gpurun_out/${1:-r05d2b}; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_resident.py -q -m gpu -x > $O/pytest1.txt 2>&1; tail -6 $O/pytest1.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "res or small_k or golden" > $O/pytest2.txt 2>&1; tail -6 $O/pytest2.txt
bash tools/r05_d2prof.sh $(basename $O)_prof
bash tools/r05_kstats.sh $(basename $O)_ks | grep -v "k_\|rocclr"
