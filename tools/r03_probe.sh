#!/bin/bash
# GPU box: where the host's time goes on config 3's list (G2S_DEBUG breakdown of a few steps), CPU facts of the box.
O=gpurun_out/r03a; rm -rf $O; mkdir -p $O
nproc > $O/cpu.txt; cat /sys/fs/cgroup/cpu.max >> $O/cpu.txt 2>/dev/null; lscpu | head -20 >> $O/cpu.txt
G2S_DEBUG=1 timeout 300 python bench.py --config C3 --steps 6 --warmup 2 --no-cpu-baseline --prime-seconds 0.3 > $O/c3.json 2> $O/c3_debug.txt
tail -60 $O/c3_debug.txt
python tools/bsum.py C3 < $O/c3.json
timeout 300 python bench.py --no-cpu-baseline > $O/c2.json 2> $O/c2.err
python tools/bsum.py C2 < $O/c2.json
