"""Summarise rocprofv3 --pmc passes (any counters) into one JSON: mean per launch of every
counter, per kernel.

usage: pmc_sq_summary.py <out.json> <kernel substring> <dir of pass 1> [<dir of pass 2> ...]
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles (4 shader cycles)
on gfx950 (MI355X_MICROARCH.md, cycle-constants table); they are reported as counted.
"""
import csv
import glob
import json
import os
import sys


def main():
    outp, kern = sys.argv[1], sys.argv[2]
    acc = {}
    for d in sys.argv[3:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row.get("Kernel_Name", "")
                if kern not in name:
                    continue
                c = row.get("Counter_Name")
                acc.setdefault(c, {}).setdefault(row.get("Dispatch_Id"), 0.0)
                acc[c][row.get("Dispatch_Id")] += float(row.get("Counter_Value", 0))
    res = {"kernel": kern, "how": "rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py ... ; mean over the "
                                  "launches of the kernel, summed over all waves/SEs of a launch",
           "counters": {c: {"launches": len(v), "mean_per_launch": sum(v.values()) / len(v)} for c, v in sorted(acc.items())}}
    json.dump(res, open(outp, "w"), indent=1)
    print(json.dumps({c: round(v["mean_per_launch"], 1) for c, v in res["counters"].items()}))


if __name__ == "__main__":
    main()
