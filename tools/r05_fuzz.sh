#!/bin/bash
# GPU box: differential campaign of round 5 (tools/fuzz_parity.py: random graphs, lists and parameters, product against
# the oracle gap by gap) around phase D2 on the device (d2_device.hip), resident mode forced on every list:
#   every closure the fill kernels leave goes to g2s_d2_small / g2s_d2_big (G2S_DEVICE_D2=1);
#   ... all of them through the large instantiation (G2S_D2_BIG=2);
#   ... with no chains contracted (G2S_D2_NO_CHAINS=1: the whole graph of runs through the component search);
#   ... every gap through the large variant of the fill kernel first (G2S_FORCE_SEGX=1);
#   ... and as several lists in flight (--in-flight); then the default choice.
# usage: tools/r05_fuzz.sh [name] [seed base, default 500] [seconds scale, default 1]
O=gpurun_out/${1:-r05fuzz}; rm -rf $O; mkdir -p $O
B=${2:-500}; X=${3:-1}
leg() {  # title, seconds, seed, extra args, env...
  local title=$1 secs=$2 seed=$3 args=$4; shift 4
  echo "## $title: $* --seconds $secs --seed $seed $args" | tee -a $O/fuzz.txt
  # (a failing configuration's line — what tools/fuzz_parity.py --replay and tools/hammer_config.py take — is kept)
  env "$@" timeout $((secs + 200)) python tools/fuzz_parity.py --seconds $secs --seed $seed $args > $O/leg.txt 2>&1
  { grep -A1 "^FAIL" $O/leg.txt | cut -c1-900; tail -3 $O/leg.txt | grep "^#"; } | tee -a $O/fuzz.txt
}
leg "phase D2 on the device" $((200 * X)) $((B + 0)) "--big 0.4 --scaffold 0.2" G2S_RESIDENT=1 G2S_DEVICE_D2=1
leg "... every closure through the large instantiation" $((150 * X)) $((B + 1)) "--big 0.4 --scaffold 0.2" G2S_RESIDENT=1 G2S_DEVICE_D2=1 G2S_D2_BIG=2
leg "... no chains contracted" $((100 * X)) $((B + 2)) "--big 0.3 --scaffold 0.2" G2S_RESIDENT=1 G2S_DEVICE_D2=1 G2S_D2_NO_CHAINS=1
leg "... behind the large variant of the fill kernel" $((120 * X)) $((B + 3)) "--big 0.3 --scaffold 0.2" G2S_RESIDENT=1 G2S_DEVICE_D2=1 G2S_FORCE_SEGX=1
leg "... lists in flight" $((150 * X)) $((B + 4)) "--in-flight --big 0.3 --scaffold 0" G2S_RESIDENT=1 G2S_DEVICE_D2=1
leg "default choice" $((120 * X)) $((B + 5)) "--big 0.5 --scaffold 0.2" A=1
