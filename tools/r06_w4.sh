for e in "X=1" "G2S_D2_SMALL_WAVES=1" "G2S_DEVICE_D2=0" "G2S_FLANK_KERNEL=1" "X=2"; do
env $e timeout 400 python bench.py --gpus 4 --config C3 --share-device --weak --no-cpu-baseline --steps 6 --warmup 2 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['resident']
print('weak N=4 [$e]', d['value'], 'gaps/s', d['ms_per_step'], 'ms/step', r.get('team_ms_by_session'))"
done
