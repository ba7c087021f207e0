#!/bin/bash
# GPU box: differential campaign (tools/fuzz_parity.py: random graphs, lists and parameters, product against the
# oracle gap by gap) with resident mode forced on every list, with the default choice, and on the host path
O=gpurun_out/${1:-r03fuzz}; rm -rf $O; mkdir -p $O
echo "## resident mode forced on every list (G2S_RESIDENT=1): --seconds 240 --seed 301 --big 0.3 --scaffold 0.3" | tee -a $O/fuzz.txt
G2S_RESIDENT=1 timeout 400 python tools/fuzz_parity.py --seconds 240 --seed 301 --big 0.3 --scaffold 0.3 2>&1 | tail -3 | tee -a $O/fuzz.txt
echo "## default (lists of 256 gaps and more on the device): --seconds 200 --seed 302 --big 0.5 --scaffold 0.2" | tee -a $O/fuzz.txt
timeout 400 python tools/fuzz_parity.py --seconds 200 --seed 302 --big 0.5 --scaffold 0.2 2>&1 | tail -3 | tee -a $O/fuzz.txt
echo "## host path only (G2S_RESIDENT=0): --seconds 120 --seed 303 --big 0.3 --scaffold 0.2" | tee -a $O/fuzz.txt
G2S_RESIDENT=0 timeout 300 python tools/fuzz_parity.py --seconds 120 --seed 303 --big 0.3 --scaffold 0.2 2>&1 | tail -3 | tee -a $O/fuzz.txt
