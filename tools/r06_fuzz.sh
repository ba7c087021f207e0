#!/bin/bash
# GPU box: differential campaign of round 6's second session (tools/fuzz_parity.py: random graphs, lists and parameters,
# product against the oracle gap by gap), resident mode forced on every list, around what the session changed in the
# kernels: short lists' guesses (which gaps guess is the waves' timing — none, the default share, every gap, a third),
# phase B's rounds in two stretches (every short list), the trace kernel's walk and its last wave's copy of the summary
# (every list), one wave per gap in the trace kernel, lists in flight, the jitter build (sync_debug.h) on the default.
# usage: tools/r06_fuzz.sh [name] [seed base, default 700] [seconds scale, default 1]
O=gpurun_out/${1:-r06fuzz}; rm -rf $O; mkdir -p $O
B=${2:-700}; X=${3:-1}
leg() {  # title, seconds, seed, extra args, env...
  local title=$1 secs=$2 seed=$3 args=$4; shift 4
  echo "## $title: $* --seconds $secs --seed $seed $args" | tee -a $O/fuzz.txt
  # (a failing configuration's line — what tools/fuzz_parity.py --replay and tools/hammer_config.py take — is kept)
  env "$@" timeout $((secs + 200)) python tools/fuzz_parity.py --seconds $secs --seed $seed $args > $O/leg.txt 2>&1
  { grep -A1 "^FAIL" $O/leg.txt | cut -c1-900; tail -3 $O/leg.txt | grep "^#"; } | tee -a $O/fuzz.txt
}
leg "the default: the early gaps of a short list guess" $((240 * X)) $((B + 0)) "--big 0.3 --scaffold 0.2" G2S_RESIDENT=1
leg "... every gap guesses" $((150 * X)) $((B + 1)) "--big 0.2 --scaffold 0.2" G2S_RESIDENT=1 G2S_GUESS_PERCENT=100
leg "... a third of the gaps" $((120 * X)) $((B + 2)) "--big 0.2 --scaffold 0.2" G2S_RESIDENT=1 G2S_GUESS_PERCENT=33
leg "... no guesses" $((100 * X)) $((B + 3)) "--big 0.2 --scaffold 0.2" G2S_RESIDENT=1 G2S_TRACE_GUESS=0
leg "... the trace kernel on one wave a gap" $((100 * X)) $((B + 4)) "--big 0.2 --scaffold 0.1" G2S_RESIDENT=1 G2S_TRACE_WAVES=1
leg "... lists in flight" $((150 * X)) $((B + 5)) "--in-flight --big 0.3 --scaffold 0" G2S_RESIDENT=1
leg "... the jitter build" $((150 * X)) $((B + 6)) "--big 0.2 --scaffold 0.1" G2S_RESIDENT=1 G2S_LIBRARY=$PWD/gap2seq_amd/_jit/libg2s_hip.so
leg "... phase D2 on the device, behind the large variant of the fill kernel" $((100 * X)) $((B + 7)) "--big 0.3 --scaffold 0.2" G2S_RESIDENT=1 G2S_DEVICE_D2=1 G2S_FORCE_SEGX=1
leg "the library's own choice of path" $((120 * X)) $((B + 8)) "--big 0.5 --scaffold 0.2" A=1
