"""Print the essentials of bench.py JSON lines read from stdin (helper for GPU-box sweeps)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
for ln in sys.stdin:
    if not ln.startswith("{"):
        continue
    d = json.loads(ln)
    b = d["breakdown_ms_per_step"]
    print(tag, d["value"], "ms/step", d["ms_per_step"], "fill", b["fill_lds_kernel"], "extract", b["extract_lds_kernel"],
          "hbm", b["hbm_tier_kernels"], "d2h", b["d2h_closures"], "host", b["host_phase_d"], "frac", d["roofline"]["frac"])
