#!/bin/bash
# One MI355X standing in for N: `bench.py --gpus N --share-device` puts the team's N sessions on device 0, ONE GROUP PER
# SESSION as N GPUs would get them.  What a one-GPU box can show of the N>1 path: phase D3 sharded over the sessions
# (each traces its own group and writes its own results), the equality with the one-session result, the team path's
# overhead — not a scaling curve.  Strong (one 10 000-gap list) and weak (10 000 gaps per session); and the gather
# form (G2S_TEAM_GATHER=1: the groups' records copied to the lead's device, phase D3 there) for comparison.
for N in 1 2 4 8; do
  timeout 300 python bench.py --gpus $N --config C3 --share-device --no-cpu-baseline --steps 10 --warmup 2 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['resident']
print('strong N=$N', d['value'], 'gaps/s', d['ms_per_step'], 'ms/step', '|', r.get('team_phase_d3'), '| by session ms:', r.get('team_ms_by_session'))"
done
for N in 2 4 8; do
  G2S_TEAM_GATHER=1 timeout 300 python bench.py --gpus $N --config C3 --share-device --no-cpu-baseline --steps 10 --warmup 2 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['resident']
print('strong N=$N (gather form)', d['value'], 'gaps/s', d['ms_per_step'], 'ms/step', '|', r.get('team_phase_d3'))"
done
for N in 2 4; do
  timeout 400 python bench.py --gpus $N --config C3 --share-device --weak --no-cpu-baseline --steps 6 --warmup 2 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['resident']
print('weak N=$N', d['config']['gaps'], 'gaps', d['value'], 'gaps/s', d['ms_per_step'], 'ms/step', '|', r.get('team_phase_d3'))"
done
