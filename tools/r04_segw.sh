#!/bin/bash
# GPU box: the large variant on eight waves (fill_segw.hip): parity suite through it, config 5 at full size, bench lines
O=gpurun_out/${1:-r04segw}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "segx" > $O/pytest_segx.txt 2>&1; tail -5 $O/pytest_segx.txt

timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "c5 or deep" > $O/pytest_c5.txt 2>&1; tail -5 $O/pytest_c5.txt
G2S_DEBUG=1 timeout 600 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $O/c5.json 2> $O/c5_debug.txt; python tools/bsum.py C5 < $O/c5.json
grep -E "slow gap|analysis" $O/c5_debug.txt | tail -12
timeout 600 python tools/segw_profile.py C5 2>&1 | tail -12 | tee $O/segw_profile_c5.txt
