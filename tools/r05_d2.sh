#!/bin/bash
# GPU box: phase D2 on the device (d2_device.hip) — the tests that exercise it, then what the lists say
O=gpurun_out/${1:-r05d2}; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_resident.py tests/test_gpu_parity.py -q -m gpu -x -k "res or resident or small_k or in_flight or golden" > $O/pytest1.txt 2>&1; tail -15 $O/pytest1.txt
for C in C2 C3 C5; do
  timeout 600 python bench.py --config $C --no-cpu-baseline --steps 5 --warmup 2 --prime-seconds 0.2 > $O/$C.json 2> $O/$C.err
  python - $O/$C.json <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("resident", {})
    print(d["config"]["config"], "gaps/s", d["value"], "ms/step", d["ms_per_step"], "| finished on device", r.get("lists_finished_on_the_device"), "host-finished gaps", r.get("gaps_finished_by_the_host"), "fallbacks", r.get("lists_given_back_to_the_host_path"))
except Exception as e:
    print("no line:", e); print(open(sys.argv[1]).read()[-2000:])
PY
  tail -3 $O/$C.err
done
