#!/bin/bash
# usage: run_guarded.sh SECONDS cmd...  Runs cmd in the background; when it is still running after SECONDS it is
# killed and the script returns 124 WITHOUT waiting for it (a process stuck in the GPU driver cannot be reaped,
# and waiting would spend the GPU box's whole time limit).
T=$1; shift
"$@" &
pid=$!
for ((i = 0; i < T; i++)); do
  if ! kill -0 $pid 2>/dev/null; then wait $pid; exit $?; fi
  sleep 1
done
kill -9 $pid 2>/dev/null
echo "guard: abandoned after ${T}s" >&2
exit 124
