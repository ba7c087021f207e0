#!/bin/bash
# GPU box: config 5, phase D2 on the device, by the number of g2s_d2_small workgroups: step, ticks per closure
O=gpurun_out/${1:-r05c5wgs}; rm -rf $O; mkdir -p $O
export G2S_DEVICE_D2=1
for W in ${2:-64 128 256 512 1024}; do
  export G2S_D2_SMALL_WGS=$W
  timeout 600 python bench.py --config C5 --no-cpu-baseline --no-c3-beside --prime-seconds 0 > $O/c5_$W.json 2> $O/err.txt
  echo "== small wgs $W"; python tools/bsum.py C5 < $O/c5_$W.json | cut -c1-60
  G2S_D2_LOG=$O/d2log_$W.txt G2S_D2_PROF=1 timeout 600 python bench.py --config C5 --no-cpu-baseline --no-c3-beside --steps 2 --warmup 1 --prime-seconds 0 > $O/prof.json 2> $O/prof.err
  python tools/d2_log.py $O/d2log_$W.txt --top 0 | grep -v "sections\|slowest"
done
