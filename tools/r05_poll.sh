#!/bin/bash
# GPU box: where g2s_d2_small runs and with how many workgroups, on config 3's list
O=gpurun_out/${1:-r05poll}; rm -rf $O; mkdir -p $O
run() { local label=$1; shift
  env "$@" timeout 600 python bench.py --config C3 --no-cpu-baseline > $O/x.json 2> $O/err.txt
  python - $O/x.json "$label" <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("resident", {})
print(sys.argv[2].ljust(48), "gaps/s", d["value"], "ms/step", d["ms_per_step"], "| host-finished", r.get("gaps_finished_by_the_host"), "| fill kernel", d.get("roofline", {}).get("kernel_ms_per_launch"))
PY
}
for rep in 1 2; do
run "closures on the host" G2S_DEVICE_D2=0
run "behind, 512 wgs, large 8" G2S_D2_POLL=0
run "behind, 512 wgs, no large" G2S_D2_POLL=0 G2S_D2_BIG=0
run "behind, 1024 wgs, no large" G2S_D2_POLL=0 G2S_D2_BIG=0 G2S_D2_SMALL_WGS=1024
run "behind, 128 wgs, no large" G2S_D2_POLL=0 G2S_D2_BIG=0 G2S_D2_SMALL_WGS=128
run "beside (16 polling), 512 wgs, large 8" G2S_D2_POLL=1
run "beside (16 polling), 128 wgs, no large" G2S_D2_POLL=1 G2S_D2_BIG=0 G2S_D2_SMALL_WGS=128
run "beside (4 polling), 128 wgs, no large" G2S_D2_POLL=1 G2S_D2_BIG=0 G2S_D2_SMALL_WGS=128 G2S_D2_POLL_WGS=4
done
