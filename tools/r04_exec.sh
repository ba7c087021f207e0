#!/bin/bash
O=gpurun_out/${1:-r04exec}; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "execute or cli or team or c1" > $O/pytest_exec.txt 2>&1; tail -8 $O/pytest_exec.txt
timeout 1200 python -m pytest tests/test_gpu_resident.py tests/test_gpu_fullsize.py -x -q > $O/pytest_res.txt 2>&1; tail -5 $O/pytest_res.txt
