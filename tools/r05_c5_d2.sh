#!/bin/bash
# GPU box: config 5 with phase D2 on the device — g2s_d2_* sections (G2S_D2_PROF), kernel statistics and the kernels
# of the last lists in time order (rocprofv3 --kernel-trace)
O=gpurun_out/${1:-r05c5d2}; rm -rf $O; mkdir -p $O
export G2S_DEVICE_D2=1
G2S_D2_LOG=$O/d2log.txt G2S_D2_PROF=1 timeout 600 python bench.py --config C5 --no-cpu-baseline --no-c3-beside --steps 2 --warmup 1 --prime-seconds 0 > $O/prof.json 2> $O/prof.err
python tools/d2_log.py $O/d2log.txt
grep "g2s_d2" $O/prof.err | tail -3
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o c5 -- python3 bench.py --config C5 --no-cpu-baseline --no-c3-beside --steps 6 --warmup 2 --prime-seconds 0 > $O/c5.json 2> $O/err.txt
python tools/bsum.py C5 < $O/c5.json
python3 tools/kstats.py $(find $O/trace -name "*kernel_stats.csv" | head -1) | head -12
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/kernel_timeline.py $F --from-end-ms ${2:-25} --window-ms ${3:-24} > $O/timeline.txt; tail -90 $O/timeline.txt
gzip -c $F > $O/c5_kernel_trace.csv.gz; rm -rf $O/trace
