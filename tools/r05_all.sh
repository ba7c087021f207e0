#!/bin/bash
# GPU box: the whole GPU suite (every failure listed), the default bench line, config 3 and 5 lines, g2s_d2_* sections
O=gpurun_out/${1:-r05all}; rm -rf $O; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench_err.txt; python tools/bsum.py C2 < $O/bench.json
for V in 0 1; do for C in C3 C5; do
  G2S_DEVICE_D2=$V timeout 600 python bench.py --config $C --no-cpu-baseline > $O/${C}_$V.json 2> $O/err.txt
  python - $O/${C}_$V.json $V <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("resident", {})
print(d["config"]["config"], "device_d2=%s" % sys.argv[2], "gaps/s", d["value"], "ms/step", d["ms_per_step"], "| host-finished", r.get("gaps_finished_by_the_host"))
PY
done; done
bash tools/r05_d2prof.sh $(basename $O)_prof | grep -v "^\[gpurun"
