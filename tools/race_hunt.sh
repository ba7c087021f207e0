#!/bin/bash
# GPU box: every kernel path N times with the normal build and the two race-hunting builds (csrc/sync_debug.h); the
# digests of a path have to agree across the builds, and no call may differ from its run's first.  usage: race_hunt.sh OUT [N]
O=gpurun_out/${1:-r06race}; N=${2:-100}
mkdir -p $O; : > $O/race_hunt.jsonl
for p in seg2 seg segw d2; do
  for v in "" _jit _par; do
    L=gap2seq_amd/$v/libg2s_hip.so; [ -z "$v" ] && L=gap2seq_amd/libg2s_hip.so
    n=$N; [ "$v" = "_jit" ] && [ $p = seg ] && n=$((N / 2))
    G2S_LIBRARY=$PWD/$L timeout 900 python tools/race_hunt.py $p $n | tee -a $O/race_hunt.jsonl
  done
done
python - "$O/race_hunt.jsonl" <<'PY' | tee $O/race_hunt.txt
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
bad = 0
for p in ("seg2", "seg", "segw", "d2"):
    rs = [r for r in rows if r["path"] == p]
    dig = {r["digest"] for r in rs}
    calls = sum(r["runs"] for r in rs); diff = sum(r["differ"] for r in rs)
    ok = len(dig) == 1 and diff == 0 and len(rs) == 3
    bad += 0 if ok else 1
    print("%-5s %d gaps: %d calls over the builds %s, %d differ from their run's first; digests %s%s" % (
        p, rs[0]["gaps"] if rs else 0, calls, [r["library"] for r in rs], diff, sorted(dig), "" if ok else "   <-- DISAGREE"))
print("%d of 4 paths disagree" % bad)
PY
