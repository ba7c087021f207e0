"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/*.json.

usage: pmc_summary.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> <kernel substring> <out.json>
The counter CSVs are the *counter_collection.csv files rocprofv3 writes with --output-format csv.
gfx950 correction (guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE is reported in KB and
counts 128-byte requests of wide coalesced reads as 64 bytes, so the read side is doubled; the
uncorrected figure is kept next to it.
"""
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row.get("Kernel_Name", "")
            key = (name, row.get("Dispatch_Id"))
            out.setdefault(name, {}).setdefault(key, 0.0)
            out[name][key] += float(row.get("Counter_Value", 0))
    return {k: list(v.values()) for k, v in out.items()}


def main():
    fdir, wdir, kern, outp = sys.argv[1:5]
    fetch, write = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    res = {"how": "two separate passes: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --output-format csv -- python3 "
                  "bench.py --steps 10 --warmup 2 --no-cpu-baseline; mean over the launches of the kernel",
           "correction": "gfx950: FETCH_SIZE (KB) tallies 128 B requests of wide coalesced reads at 64 B -> read side "
                         "doubled as MI355X_MICROARCH.md prescribes (upper estimate for 16 B records at a 32 B stride)",
           "per_kernel": {}}
    for name in sorted(set(fetch) | set(write)):
        short = name.split("(")[0]
        fv, wv = fetch.get(name, []), write.get(name, [])
        res["per_kernel"][short] = {"launches": max(len(fv), len(wv)),
                                    "FETCH_SIZE_KB_mean": sum(fv) / len(fv) if fv else None,
                                    "WRITE_SIZE_KB_mean": sum(wv) / len(wv) if wv else None}
        if kern in name and fv and wv:
            fb, wb = sum(fv) / len(fv) * 1024, sum(wv) / len(wv) * 1024
            res.update(kernel=short, fetch_bytes_raw=fb, write_bytes=wb, hbm_bytes_per_launch=int(2 * fb + wb),
                       hbm_bytes_per_launch_uncorrected=int(fb + wb))
    json.dump(res, open(outp, "w"), indent=1)
    print(json.dumps({k: res.get(k) for k in ("kernel", "hbm_bytes_per_launch", "hbm_bytes_per_launch_uncorrected")}))


if __name__ == "__main__":
    main()
