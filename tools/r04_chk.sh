#!/bin/bash
O=gpurun_out/${1:-r04chk}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fuzz_regressions or segx or execute" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "c5 or c4" > $O/pytest2.txt 2>&1; tail -4 $O/pytest2.txt
bash tools/r04_toy.sh
echo "## every gap through g2s_fill_segw (G2S_FORCE_SEGX=1): --seconds 150 --seed 400" | tee -a $O/fuzz.txt
G2S_FORCE_SEGX=1 timeout 400 python tools/fuzz_parity.py --seconds 150 --seed 400 --big 0.3 --scaffold 0.2 2>&1 | grep -E "FAIL|Error|done" | cut -c1-300 | tee -a $O/fuzz.txt
