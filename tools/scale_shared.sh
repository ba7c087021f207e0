#!/bin/bash
# One MI355X standing in for N (bench.py --gpus N --share-device: the team's N sessions on device 0, one group per
# session as N GPUs would get them): strong (config 3's one list), weak (10 000 gaps per session), and a STREAM of
# lists over the team (--stream-lists 6: every list cut into one share per session, the rand() stream carried from
# share to share and from list to list; against the same lists on one session).  What a one-GPU box can show of the
# N > 1 path: equality with the one-session results, the per-session times, the team path's overhead — not a curve.
for N in 1 2 4 8; do
  timeout 300 python bench.py --gpus $N --config C3 --share-device --no-cpu-baseline --steps 10 --warmup 2 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['resident']
print('strong N=$N', d['value'], 'gaps/s', d['ms_per_step'], 'ms/step', '|', r.get('team_phase_d3'), '| by session ms:', r.get('team_ms_by_session'))"
done
for N in 2 4; do
  timeout 400 python bench.py --gpus $N --config C3 --share-device --weak --no-cpu-baseline --steps 6 --warmup 2 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['resident']
print('weak N=$N', d['config']['gaps'], 'gaps', d['value'], 'gaps/s', d['ms_per_step'], 'ms/step', '|', r.get('team_phase_d3'), '| by session ms:', r.get('team_ms_by_session'))"
done
for N in 2 4; do
  timeout 600 python bench.py --gpus $N --config C3 --share-device --weak --stream-lists 6 --no-cpu-baseline --steps 4 --warmup 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('stream N=$N', d.get('stream_lists'))"
done
