"""Print the essentials of bench.py JSON lines (helper for GPU-box sweeps).
usage: bsum.py FILE...   or   ... | bsum.py [TAG]   (arguments that are files are read; otherwise stdin)"""
import itertools
import json
import os
import sys

files = [a for a in sys.argv[1:] if os.path.isfile(a)]
tag = "" if files or len(sys.argv) < 2 else sys.argv[1]
lines = itertools.chain.from_iterable(open(f) for f in files) if files else sys.stdin
for ln in lines:
    if not ln.startswith("{"):
        continue
    d = json.loads(ln)
    b = d["breakdown_ms_per_step"]
    r = d["roofline"]
    print(tag, d["value"], "gaps/s | ms/step", d["ms_per_step"], "prepare", b["prepare_flank_lookup_and_upload"], "kernel", r["kernel"], b.get("fill_seg_kernel", 0.0), "+segx", b.get("fill_segx_kernel", 0.0), "+lds",
          b["fill_lds_kernel"], "(per launch", r["kernel_ms_per_launch"], "x", r["launches_per_step"], ") hbm-tier",
          b["hbm_tier_kernels"], "host", b["host_phase_d"], "| frac", r["frac"], "| filled", d["filled"],
          "| c3:", (d.get("c3_on_one_gpu") or {}).get("value"), (d.get("c3_on_one_gpu") or {}).get("kernel_ms_per_launch"),
          (d.get("c3_on_one_gpu") or {}).get("roofline_frac"), "| host us", list((b.get("host_us_inside_the_call") or {}).values()))
