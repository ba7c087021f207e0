#!/bin/bash
# GPU box: the kernels of a config's last steps in time order (rocprofv3 --kernel-trace, tools/kernel_timeline.py)
# usage: r05_timeline.sh OUT CONFIG FROM_END_MS WINDOW_MS [bench flags]
O=gpurun_out/${1:-r05tl}; C=${2:-C2}; FE=${3:-1.0}; WIN=${4:-0.6}; shift 4
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --config $C --no-cpu-baseline --no-c3-beside --steps 20 --warmup 5 --prime-seconds 0.3 "$@" > $O/bench.json 2> $O/err.txt
python tools/bsum.py $C < $O/bench.json | cut -c1-200
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/kernel_timeline.py $F --from-end-ms $FE --window-ms $WIN > $O/timeline.txt; cat $O/timeline.txt | head -70
gzip -c $F > $O/kernel_trace.csv.gz; rm -rf $O/trace
