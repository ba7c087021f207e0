#!/bin/bash
# GPU box: kernel timeline of config 3 lists with two in flight (rocprofv3 --kernel-trace, timestamps)
O=gpurun_out/${1:-r04tl}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o c3 -- python3 bench.py --config C3 --no-cpu-baseline --stream-lists 12 --steps 3 --warmup 1 --prime-seconds 0.5 > $O/c3.json 2> $O/err.txt
python -c "import sys,json; d=json.loads(open('$O/c3.json').read()); print('C3', d['value'], d['stream_lists'])"
F=$(find $O/trace -name "*kernel_trace.csv" | head -1); ls -la $F
python tools/kernel_timeline.py $F --from-end-ms 4 --window-ms 2.2 > $O/timeline.txt; tail -80 $O/timeline.txt
gzip -c $F > $O/c3_kernel_trace.csv.gz; rm -rf $O/trace
