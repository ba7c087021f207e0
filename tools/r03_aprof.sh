#!/bin/bash
# GPU box: cycles per section of a phase A round (instrumented build gap2seq_amd/_ab/prof, one wave per gap)
cp gap2seq_amd/libg2s_hip.so /tmp/product.so
cp gap2seq_amd/_ab/prof/libg2s_hip.so gap2seq_amd/libg2s_hip.so
G2S_RESIDENT=0 G2S_SEG_WAVES=1 python tools/seg_profile.py C2 | tail -8
G2S_RESIDENT=0 G2S_SEG_WAVES=1 python tools/seg_profile.py C3 | tail -8
cp /tmp/product.so gap2seq_amd/libg2s_hip.so
