#!/bin/bash
# GPU box: the full GPU suite, then the bench lines with the resident mode's debug breakdown.
O=gpurun_out/r03e; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
G2S_DEBUG=1 timeout 300 python bench.py --config C3 --steps 6 --warmup 2 --no-cpu-baseline > $O/c3.json 2> $O/c3_debug.txt
grep "resident mode" $O/c3_debug.txt | tail -4
python tools/bsum.py C3 < $O/c3.json
timeout 300 python bench.py --no-cpu-baseline > $O/c2.json 2> $O/c2.err
python tools/bsum.py C2 < $O/c2.json
G2S_RESIDENT=1 timeout 300 python bench.py --no-cpu-baseline --no-c3-beside > $O/c2r.json 2> $O/c2r.err
python tools/bsum.py C2res < $O/c2r.json
