#!/bin/bash
# GPU box: cycles per section of a phase B round and of the tail (instrumented build gap2seq_amd/_ab/prof, made with
# -DG2S_SEG_PROFILE), configs 2 and 3; the product library is put back afterwards.
O=gpurun_out/${1:-r03prof}; rm -rf $O; mkdir -p $O
cp gap2seq_amd/libg2s_hip.so /tmp/product.so
cp gap2seq_amd/_ab/prof/libg2s_hip.so gap2seq_amd/libg2s_hip.so
G2S_RESIDENT=0 python tools/seg_profile.py C2 | tee $O/segprof_c2.txt
G2S_RESIDENT=0 python tools/seg_profile.py C3 | tee $O/segprof_c3.txt
cp /tmp/product.so gap2seq_amd/libg2s_hip.so
