#!/bin/bash
# GPU box: rocprofv3 kernel statistics and the kernel timeline of one step (config $2, default C2)
O=gpurun_out/${1:-r03k}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --config ${2:-C2} --steps ${3:-200} --warmup 20 --no-cpu-baseline --no-c3-beside --prime-seconds 0.2 > $O/bench.json 2> $O/rp.err
cp $O/stats/*/*_kernel_stats.csv $O/c2_kernel_stats.csv
cp $O/stats/*/*_kernel_trace.csv $O/c2_kernel_trace.csv 2>/dev/null
python3 tools/kstats.py $O/c2_kernel_stats.csv | head -14
python3 - $O/c2_kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete step: from the last g2s_resolve_flanks on
idx = [i for i, r in enumerate(rows) if "resolve_flanks" in r["Kernel_Name"]]
i0 = idx[-2]; i1 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %8.1f us  %-40s grid %s wg %s" % (s / 1e3, e / 1e3, r["Kernel_Name"][:40], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
PY
rm -rf $O/stats; head -c 3000000 $O/c2_kernel_trace.csv > $O/trace_head.csv; rm -f $O/c2_kernel_trace.csv
