#!/bin/bash
# GPU box: config 5, phase D2 on the device, by the number of threads of a g2s_d2_big workgroup (rebuilds d2_device.o)
O=gpurun_out/${1:-r05c5nt}; rm -rf $O; mkdir -p $O
export G2S_DEVICE_D2=1
for NT in ${2:-256 512 1024}; do
  touch gap2seq_amd/csrc/d2_device.hip; make -C gap2seq_amd/csrc EXTRA=-DG2S_D2_BIG_NT=${NT}u > $O/make_$NT.txt 2>&1 || { tail -5 $O/make_$NT.txt; continue; }
  timeout 600 python bench.py --config C5 --no-cpu-baseline --no-c3-beside --prime-seconds 0 > $O/c5_$NT.json 2> $O/err.txt
  echo "== big threads $NT"; python tools/bsum.py C5 < $O/c5_$NT.json | cut -c1-60
  G2S_D2_LOG=$O/d2log_$NT.txt G2S_D2_PROF=1 timeout 600 python bench.py --config C5 --no-cpu-baseline --no-c3-beside --steps 2 --warmup 1 --prime-seconds 0 > $O/prof.json 2> $O/prof.err
  python tools/d2_log.py $O/d2log_$NT.txt --top 3 | grep -A12 "g2s_d2_big"
done
