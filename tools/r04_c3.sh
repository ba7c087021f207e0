#!/bin/bash
# GPU box: config 3 lines (and the link variant of the look-ups), lists in flight, config 4
O=gpurun_out/${1:-r04c3}; rm -rf $O; mkdir -p $O
for r in 1 2 3; do timeout 300 python bench.py --config C3 --no-cpu-baseline | tee -a $O/c3.json | python tools/bsum.py C3 | cut -c1-200; done
for r in 1 2; do G2S_FLANKS_OVER_THE_LINK=1 timeout 300 python bench.py --config C3 --no-cpu-baseline | tee -a $O/c3_link.json | python tools/bsum.py C3-link | cut -c1-200; done
for r in 1 2; do timeout 400 python bench.py --config C3 --no-cpu-baseline --stream-lists 12 --steps 10 | tee -a $O/c3s.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 stream', d['stream_lists']['value'], d['stream_lists']['ms_per_list'])"; done
G2S_FLANKS_OVER_THE_LINK=1 timeout 400 python bench.py --config C3 --no-cpu-baseline --stream-lists 12 --steps 10 | tee -a $O/c3s_link.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 stream link', d['stream_lists']['value'], d['stream_lists']['ms_per_list'])"
timeout 300 python bench.py --config C4 --no-cpu-baseline | tee -a $O/c4.json | python tools/bsum.py C4 | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "c3 or c4" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
