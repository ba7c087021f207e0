#!/bin/bash
# GPU box: resident tests; config 2 on the device against the host path (three runs each, interleaved); config 3
O=gpurun_out/${1:-r03t}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for r in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-c3-beside | tee -a $O/c2.json | python tools/bsum.py C2-device
  G2S_RESIDENT=0 timeout 300 python bench.py --no-cpu-baseline --no-c3-beside | tee -a $O/c2host.json | python tools/bsum.py C2-host
done
G2S_DEBUG=1 timeout 300 python bench.py --no-cpu-baseline --no-c3-beside --steps 5 --warmup 2 2>&1 | grep "resident mode" | tail -3
timeout 300 python bench.py --config C3 --no-cpu-baseline | tee -a $O/c3.json | python tools/bsum.py C3
