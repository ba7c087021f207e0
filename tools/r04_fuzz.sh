#!/bin/bash
# GPU box: differential campaign (tools/fuzz_parity.py: random graphs, lists and parameters, product against the
# oracle gap by gap): every gap through the eight-wave large variant (G2S_FORCE_SEGX=1, host path), resident mode
# forced on every list (gaps that outgrow the regular tier rerun in the large variant on the stream), the default
# choice, the host path only; then the same gaps as several lists in flight (resident mode forced, and the default choice).
# usage: tools/r04_fuzz.sh [name] [seed base, default 400] [seconds scale, default 1]
O=gpurun_out/${1:-r04fuzz}; rm -rf $O; mkdir -p $O
B=${2:-400}; X=${3:-1}
echo "## every gap through g2s_fill_segw (G2S_FORCE_SEGX=1): --seconds $((200 * X)) --seed $((B + 0)) --big 0.3 --scaffold 0.2" | tee -a $O/fuzz.txt
G2S_FORCE_SEGX=1 timeout $((200 * X + 200)) python tools/fuzz_parity.py --seconds $((200 * X)) --seed $((B + 0)) --big 0.3 --scaffold 0.2 2>&1 | tail -3 | tee -a $O/fuzz.txt
echo "## resident mode forced on every list (G2S_RESIDENT=1): --seconds $((240 * X)) --seed $((B + 1)) --big 0.4 --scaffold 0.3" | tee -a $O/fuzz.txt
G2S_RESIDENT=1 timeout $((240 * X + 200)) python tools/fuzz_parity.py --seconds $((240 * X)) --seed $((B + 1)) --big 0.4 --scaffold 0.3 2>&1 | tail -3 | tee -a $O/fuzz.txt
echo "## default (lists of 256 gaps and more on the device): --seconds $((160 * X)) --seed $((B + 2)) --big 0.5 --scaffold 0.2" | tee -a $O/fuzz.txt
timeout $((160 * X + 200)) python tools/fuzz_parity.py --seconds $((160 * X)) --seed $((B + 2)) --big 0.5 --scaffold 0.2 2>&1 | tail -3 | tee -a $O/fuzz.txt
echo "## host path only (G2S_RESIDENT=0): --seconds $((100 * X)) --seed $((B + 3)) --big 0.3 --scaffold 0.2" | tee -a $O/fuzz.txt
G2S_RESIDENT=0 timeout $((100 * X + 200)) python tools/fuzz_parity.py --seconds $((100 * X)) --seed $((B + 3)) --big 0.3 --scaffold 0.2 2>&1 | tail -3 | tee -a $O/fuzz.txt
echo "## lists in flight, resident mode forced (G2S_RESIDENT=1 --in-flight: every configuration's gaps as 3-6 lists through g2s_fill_begin / g2s_fill_end): --seconds $((150 * X)) --seed $((B + 4)) --big 0.3 --scaffold 0" | tee -a $O/fuzz.txt
G2S_RESIDENT=1 timeout $((150 * X + 200)) python tools/fuzz_parity.py --in-flight --seconds $((150 * X)) --seed $((B + 4)) --big 0.3 --scaffold 0 2>&1 | tail -3 | tee -a $O/fuzz.txt
echo "## lists in flight, default (--in-flight): --seconds $((100 * X)) --seed $((B + 5)) --big 0.5 --scaffold 0" | tee -a $O/fuzz.txt
timeout $((100 * X + 200)) python tools/fuzz_parity.py --in-flight --seconds $((100 * X)) --seed $((B + 5)) --big 0.5 --scaffold 0 2>&1 | tail -3 | tee -a $O/fuzz.txt
