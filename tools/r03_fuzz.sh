#!/bin/bash
# GPU box: differential campaign (tools/fuzz_parity.py: random graphs, lists and parameters, product against the
# oracle gap by gap) with resident mode forced on every list, with the default choice, and on the host path.
# usage: tools/r03_fuzz.sh [name] [seed base, default 300] [seconds scale, default 1]
O=gpurun_out/${1:-r03fuzz}; rm -rf $O; mkdir -p $O
B=${2:-300}; X=${3:-1}
echo "## resident mode forced on every list (G2S_RESIDENT=1): --seconds $((240 * X)) --seed $((B + 1)) --big 0.3 --scaffold 0.3" | tee -a $O/fuzz.txt
G2S_RESIDENT=1 timeout $((240 * X + 200)) python tools/fuzz_parity.py --seconds $((240 * X)) --seed $((B + 1)) --big 0.3 --scaffold 0.3 2>&1 | tail -3 | tee -a $O/fuzz.txt
echo "## default (lists of 256 gaps and more on the device): --seconds $((200 * X)) --seed $((B + 2)) --big 0.5 --scaffold 0.2" | tee -a $O/fuzz.txt
timeout $((200 * X + 200)) python tools/fuzz_parity.py --seconds $((200 * X)) --seed $((B + 2)) --big 0.5 --scaffold 0.2 2>&1 | tail -3 | tee -a $O/fuzz.txt
echo "## host path only (G2S_RESIDENT=0): --seconds $((120 * X)) --seed $((B + 3)) --big 0.3 --scaffold 0.2" | tee -a $O/fuzz.txt
G2S_RESIDENT=0 timeout $((120 * X + 200)) python tools/fuzz_parity.py --seconds $((120 * X)) --seed $((B + 3)) --big 0.3 --scaffold 0.2 2>&1 | tail -3 | tee -a $O/fuzz.txt
