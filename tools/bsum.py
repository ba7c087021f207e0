"""Print the essentials of bench.py JSON lines read from stdin (helper for GPU-box sweeps)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
for ln in sys.stdin:
    if not ln.startswith("{"):
        continue
    d = json.loads(ln)
    b = d["breakdown_ms_per_step"]
    r = d["roofline"]
    print(tag, d["value"], "gaps/s | ms/step", d["ms_per_step"], "prepare", b["prepare_flank_lookup_and_upload"], "kernel", r["kernel"], b.get("fill_seg_kernel", 0.0), "+lds",
          b["fill_lds_kernel"], "(per launch", r["kernel_ms_per_launch"], "x", r["launches_per_step"], ") hbm-tier",
          b["hbm_tier_kernels"], "host", b["host_phase_d"], "| frac", r["frac"], "| filled", d["filled"],
          "| c3:", (d.get("c3_on_one_gpu") or {}).get("value"), (d.get("c3_on_one_gpu") or {}).get("kernel_ms_per_launch"),
          (d.get("c3_on_one_gpu") or {}).get("roofline_frac"))
