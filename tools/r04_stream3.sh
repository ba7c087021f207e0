#!/bin/bash
# GPU box: lists in flight with the rand() stream chained on the device: tests, then config 3 / config 2 lists streamed
O=gpurun_out/${1:-r04stream3}; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_resident.py -x -q -k "in_flight" > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "lists_in_flight" > $O/pytest2.txt 2>&1; tail -5 $O/pytest2.txt
for r in 1 2; do timeout 400 python bench.py --config C3 --no-cpu-baseline --stream-lists 12 --steps 10 | tee -a $O/c3.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3', d['value'], d['stream_lists'])"; done
G2S_NO_DEVICE_CHAIN=1 timeout 400 python bench.py --config C3 --no-cpu-baseline --stream-lists 10 --steps 10 | tee -a $O/c3_nochain.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 no chain', d['value'], d['stream_lists'])"
timeout 400 python bench.py --no-cpu-baseline --no-c3-beside --stream-lists 10 --steps 50 | tee -a $O/c2.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2', d['value'], d['stream_lists'])"
timeout 400 python bench.py --config C5 --no-cpu-baseline --stream-lists 4 --steps 3 | tee -a $O/c5.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5', d['value'], d['stream_lists'])"
timeout 400 python bench.py --config C3 --no-cpu-baseline --stream-lists 12 --in-flight 2 --steps 10 | tee -a $O/c3_depth2.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 depth 2', d['value'], d['stream_lists'])"
