#!/bin/bash
# GPU box: cycles per section of the regular tier's kernels (instrumented build gap2seq_amd/_prof, -DG2S_SEG_PROFILE)
O=gpurun_out/${1:-r04segprof}; rm -rf $O; mkdir -p $O
export G2S_LIBRARY=$PWD/gap2seq_amd/_prof/libg2s_hip.so
timeout 600 python tools/seg_profile.py C2 2>&1 | tee $O/segprof_c2.txt | tail -24 | cut -c1-250
G2S_SEG_WAVES=1 timeout 600 python tools/seg_profile.py C3 2>&1 | tee $O/segprof_c3.txt | tail -24 | cut -c1-250
