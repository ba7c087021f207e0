#!/bin/bash
# GPU box: the whole GPU suite, then the per-kernel profile and bench lines of configs 3 and 2
O=gpurun_out/${1:-r03k}; rm -rf $O; mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline --no-c3-beside > $O/bench.json 2> $O/rp.err
cp $O/stats/*/*_kernel_stats.csv $O/kernel_stats.csv; rm -rf $O/stats
python3 tools/kstats.py $O/kernel_stats.csv | head -10
G2S_DEBUG=1 timeout 300 python bench.py --config C3 --steps 6 --warmup 2 --no-cpu-baseline > $O/c3.json 2> $O/c3_debug.txt
grep "resident mode\|fill_batch:" $O/c3_debug.txt | tail -2
python tools/bsum.py C3 < $O/c3.json
timeout 300 python bench.py --no-cpu-baseline > $O/c2.json 2> $O/c2.err
python tools/bsum.py C2 < $O/c2.json
G2S_RESIDENT=0 timeout 300 python bench.py --no-cpu-baseline --config C3 > $O/c3host.json 2> $O/c3host.err
python tools/bsum.py C3host < $O/c3host.json
