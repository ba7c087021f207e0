#!/bin/bash
O=gpurun_out/${1:-r04team}; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_resident.py -x -q -k "team or slice" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
