#!/bin/bash
# GPU box: config 3's list with the repeated-k-mer closures on the host's threads (G2S_DEVICE_D2=0) and on the device, alternating
O=gpurun_out/${1:-r05abc3}; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do for V in 0 1; do
  G2S_DEVICE_D2=$V timeout 600 python bench.py --config C3 --no-cpu-baseline > $O/c3_$V_$rep.json 2> $O/err.txt
  python - $O/c3_$V_$rep.json $V <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("resident", {})
print("device_d2=%s" % sys.argv[2], "gaps/s", d["value"], "ms/step", d["ms_per_step"], "| host-finished", r.get("gaps_finished_by_the_host"), "| kernel", d.get("roofline", {}).get("kernel_ms_per_launch"))
PY
done; done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o c3 -- python3 bench.py --config C3 --no-cpu-baseline --steps 6 --warmup 2 --prime-seconds 0.3 > $O/c3_trace.json 2> $O/err2.txt
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/kernel_timeline.py $F --from-end-ms 2.2 --window-ms 1.2 > $O/timeline.txt; tail -40 $O/timeline.txt
