#!/bin/bash
# GPU box: two lists in flight: the test, then config 3 and config 2 lists streamed
O=gpurun_out/${1:-r04stream}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q -k "in_flight" > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
for r in 1 2; do timeout 400 python bench.py --config C3 --no-cpu-baseline --stream-lists 10 --steps 10 | tee -a $O/c3.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3', d['value'], d['stream_lists'])"; done
timeout 400 python bench.py --no-cpu-baseline --no-c3-beside --stream-lists 10 --steps 50 | tee -a $O/c2.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2', d['value'], d['stream_lists'])"
