#!/bin/bash
# GPU box: lists in flight with fewer gaps of the fill kernel resident per compute unit (room for the trace kernel's waves)
for pad in 0 2000 5500 9000; do
  echo "pad $pad"
  G2S_SEG_LDS_PAD=$pad timeout 400 python bench.py --config C3 --no-cpu-baseline --stream-lists 10 --steps 10 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stream_lists']; print('C3', d['value'], d['roofline']['kernel_ms_per_launch'], s['value'], s['ms_per_list'], s['one_list_at_a_time'])"
done
G2S_DEBUG=1 timeout 400 python bench.py --config C3 --no-cpu-baseline --stream-lists 4 --steps 3 2>&1 | grep -E "resident mode" | tail -12 | cut -c1-420
