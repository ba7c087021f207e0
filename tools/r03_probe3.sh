#!/bin/bash
# GPU box: resident-mode tests, then per-kernel times and the debug breakdown of config 3's list
O=gpurun_out/${1:-r03g}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline --no-c3-beside > $O/bench.json 2> $O/rp.err
cp $O/stats/*/*_kernel_stats.csv $O/kernel_stats.csv; rm -rf $O/stats
python3 tools/kstats.py $O/kernel_stats.csv | head -14
G2S_DEBUG=1 timeout 300 python bench.py --config C3 --steps 6 --warmup 2 --no-cpu-baseline > $O/c3.json 2> $O/c3_debug.txt
grep "resident mode\|fill_batch:" $O/c3_debug.txt | tail -4
python tools/bsum.py C3 < $O/c3.json
timeout 300 python bench.py --no-cpu-baseline > $O/c2.json 2> $O/c2.err
python tools/bsum.py C2 < $O/c2.json
G2S_D3_STAGE=device timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -- python3 bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline --no-c3-beside > $O/bench_stage.json 2> $O/rp2.err
cp $O/stats2/*/*_kernel_stats.csv $O/kernel_stats_stage.csv; rm -rf $O/stats2
python3 tools/kstats.py $O/kernel_stats_stage.csv | head -8
python tools/bsum.py C3stage < $O/bench_stage.json
