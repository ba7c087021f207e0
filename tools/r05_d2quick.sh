#!/bin/bash
# GPU box: the phase D2 parity tests alone, config 3 and 5 with phase D2 on the host / on the device, g2s_d2_* sections
O=gpurun_out/${1:-r05d2quick}; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_resident.py -q -m gpu -k "phase_d2 or toy or long_short" > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
for V in 0 1; do for C in C3 C5; do
  G2S_DEVICE_D2=$V timeout 600 python bench.py --config $C --no-cpu-baseline --no-c3-beside > $O/${C}_$V.json 2> $O/err.txt
  python - $O/${C}_$V.json $V <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("resident", {})
print(d["config"]["config"], "device_d2=%s" % sys.argv[2], "gaps/s", d["value"], "ms/step", d["ms_per_step"], "| host-finished", r.get("gaps_finished_by_the_host"))
PY
done; done
bash tools/r05_d2prof.sh $(basename $O)_prof | grep -v "^\[gpurun"
