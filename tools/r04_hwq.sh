#!/bin/bash
# GPU box: HIP's hardware queues per process (GPU_MAX_HW_QUEUES, default 4): lists in flight use six streams, N sessions
# on one device 2 N — streams that share a hardware queue run in order
O=gpurun_out/${1:-r04hwq}; rm -rf $O; mkdir -p $O
for Q in 4 8 16; do
  export GPU_MAX_HW_QUEUES=$Q
  for r in 1 2; do timeout 400 python bench.py --config C3 --no-cpu-baseline --stream-lists 12 --steps 5 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues $Q: C3 stream', d['stream_lists']['value'], d['stream_lists']['ms_per_list'], 'single', d['value'])"; done
  timeout 400 python bench.py --no-cpu-baseline --no-c3-beside --stream-lists 10 --steps 50 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues $Q: C2 stream', d['stream_lists']['value'], d['stream_lists']['ms_per_list'], 'single', d['value'])"
  timeout 400 python bench.py --config C3 --no-cpu-baseline --gpus 8 --share-device --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues $Q: C3 8 sessions on one device', d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/hwq.txt
