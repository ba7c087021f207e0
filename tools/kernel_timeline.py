#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace csv: per kernel dispatch its start and duration relative to the first
dispatch of a window, with the queue it ran on — what overlaps what when two lists are in flight.

  python tools/kernel_timeline.py <..._kernel_trace.csv> [--skip-ms 200] [--window-ms 3]
"""
import argparse, csv, gzip, re, sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--from-end-ms", type=float, default=6.0, help="the window starts this long before the last dispatch ends")
    ap.add_argument("--window-ms", type=float, default=3.0)
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(gzip.open(a.csv, "rt") if a.csv.endswith(".gz") else open(a.csv)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.split(r"[(<]", r["Kernel_Name"].replace("void ", ""))[0][:28], r.get("Queue_Id", "?"),
                     int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
    rows.sort()
    t_end = max(r[1] for r in rows)
    t0 = t_end - int(a.from_end_ms * 1e6)
    win = [r for r in rows if r[0] >= t0 and r[0] < t0 + int(a.window_ms * 1e6)]
    if not win:
        sys.exit("no dispatch in the window")
    base = win[0][0]
    queues = sorted(set(r[3] for r in win))
    print("queues:", queues)
    busy = {}
    for s, e, name, q, grid in win:
        print("%9.1f us  +%7.1f us  q%-3s %-28s grid %d" % ((s - base) / 1e3, (e - s) / 1e3, queues.index(q), name, grid))
        busy[name] = busy.get(name, 0) + (e - s)
    span = (max(r[1] for r in win) - base) / 1e3
    # union of busy intervals
    iv = sorted((s, e) for s, e, *_ in win)
    u, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce: u += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    u += ce - cs
    print("window %.1f us, some kernel running %.1f us (%.0f %%), sum of kernel durations %.1f us" % (span, u / 1e3, 100.0 * u / 1e3 / span, sum(busy.values()) / 1e3))
    for k, v in sorted(busy.items(), key=lambda x: -x[1]):
        print("   %-28s %8.1f us" % (k, v / 1e3))


if __name__ == "__main__":
    main()
