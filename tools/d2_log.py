#!/usr/bin/env python3
"""Summary of a G2S_D2_LOG file (g2s_session_destroy writes it when G2S_D2_PROF and G2S_D2_LOG=path are set): one line
per closure g2s_d2_small / g2s_d2_big took — gap | records << 32, segments on sink paths | nodes << 32, edges | rc << 32,
ticks (s_memrealtime: 10 ns) of the 11 sections, ticks in all, kernel | workgroup << 1 | start tick << 16.

  python tools/d2_log.py <log> [--top 12]
"""
import argparse, statistics

NAMES = ["load+sort", "dag", "chains", "cuts", "runs", "edges", "csr", "comps", "stats", "order", "verdicts"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("log")
    ap.add_argument("--top", type=int, default=12)
    ap.add_argument("--poll-model", type=int, nargs=2, metavar=("SMALL", "BIG"), help="workgroups that take closures while the fill kernels run: when would the last one be through?")
    a = ap.parse_args()
    rows, listed = [], {}
    for line in open(a.log):
        if line.startswith("T "):
            _, g, t = line.split()
            listed[int(g)] = int(t)
            continue
        w = [int(x) for x in line.split()]
        if len(w) != 16:
            continue
        rows.append(dict(gap=w[0] & 0xFFFFFFFF, nrec=w[0] >> 32, ns=w[1] & 0xFFFFFFFF, nv=(w[1] >> 32) & 0xFFFF, peel=w[1] >> 48, ne=w[2] & 0xFFFFFFFF, rc=(w[2] >> 32) & 0xFFFF, kahn=w[2] >> 48,
                         sec=w[3:14], total=w[14], kernel=w[15] & 1, wg=(w[15] >> 1) & 0x7FFF, start=w[15] >> 16))
    for k in (0, 1):
        rs = [r for r in rows if r["kernel"] == k]
        if not rs:
            continue
        t = sorted(r["total"] for r in rs)
        print("== %s: %d closures; ticks per closure min %d median %d p90 %d p99 %d max %d" % (
            "g2s_d2_big" if k else "g2s_d2_small", len(rs), t[0], t[len(t) // 2], t[len(t) * 9 // 10], t[len(t) * 99 // 100], t[-1]))
        for mode, name in ((lambda r: r["rc"] == 0 and r["sec"][1] > 0, "as a DAG"), (lambda r: r["rc"] == 0 and r["sec"][1] == 0, "on runs"), (lambda r: r["rc"] != 0, "passed on / given up")):
            ms = [r for r in rs if mode(r)]
            if not ms:
                continue
            tt = sorted(r["total"] for r in ms)
            print("  %-22s %5d closures, records median %d max %d, ticks median %d p90 %d max %d" % (
                name, len(ms), statistics.median(r["nrec"] for r in ms), max(r["nrec"] for r in ms), tt[len(tt) // 2], tt[len(tt) * 9 // 10], tt[-1]))
            print("      sections (median): " + " ".join("%s %d" % (NAMES[i], statistics.median(r["sec"][i] for r in ms)) for i in range(11)))
        print("  the slowest:")
        for r in sorted(rs, key=lambda r: -r["total"])[:a.top]:
            print("    gap %5d rc %d records %5d segs %4d nodes %4d edges %5d rounds %d + %d ticks %8d: %s" % (
                r["gap"], r["rc"], r["nrec"], r["ns"], r["nv"], r["ne"], r["peel"], r["kahn"], r["total"],
                " ".join("%s %d" % (NAMES[i], r["sec"][i]) for i in range(11) if r["sec"][i])))
    if a.poll_model and listed:
        poll_model(rows, listed, *a.poll_model)


def poll_model(rows, listed, w_small, w_big):
    """The last list of the log: its closures by the time they were listed (fill kernels, ticks), worked through by
    w_small + w_big workgroups in that order as they come, each taking what its instantiation took in the log."""
    latest = {}
    for r in rows:  # (every list of the log has the same gaps: a gap's last entry per instantiation is the last list's)
        key = (r["gap"], r["kernel"])
        if key not in latest or r["start"] > latest[key]["start"]:
            latest[key] = r
    last = [r for r in latest.values() if r["gap"] in listed]
    if not last:
        return print("no listing times")
    fill_end = min(r["start"] & 0xFFFFFFFF for r in last)
    gaps = {}
    for r in last:
        g = gaps.setdefault(r["gap"], dict(t=min(listed[r["gap"]], fill_end), small=0, big=0, nrec=r["nrec"]))
        g["big" if r["kernel"] else "small"] += r["total"]
    ts = sorted(g["t"] for g in gaps.values())
    t0 = ts[0]
    print("== the last list: %d closures listed over %d ticks (first at 0, median %d, p90 %d, last %d; the launch behind the fill kernels began at %d)" % (
        len(ts), ts[-1] - t0, ts[len(ts) // 2] - t0, ts[len(ts) * 9 // 10] - t0, ts[-1] - t0, fill_end - t0))
    behind = max((r["start"] + r["total"]) & 0xFFFFFFFF for r in last) - fill_end
    import heapq
    free = {"small": [fill_end * 0] * w_small, "big": [0] * w_big}
    done = 0
    for g in sorted(gaps.values(), key=lambda g: g["t"]):
        for k in ("small", "big"):
            # (a closure of more than 512 records goes to the large instantiation at once — the fill kernel can tell)
            if g[k] == 0 or (k == "small" and g["big"] and g["nrec"] > 512):
                continue
            f = heapq.heappop(free[k])
            end = max(f, g["t"]) + g[k]
            heapq.heappush(free[k], end)
            done = max(done, end)
    print("   behind the fill kernels (as measured): the last closure through %d ticks after the fill kernels' end" % behind)
    print("   taken as listed by %d + %d workgroups: the last closure through %d ticks after the fill kernels' end" % (w_small, w_big, max(0, done - fill_end)))
    late = sorted(gaps.values(), key=lambda g: -g["t"])[:8]
    print("   the last listed: " + ", ".join("%d before the end: %d records, %d ticks" % (fill_end - g["t"], g["nrec"], g["small"] + g["big"]) for g in late))


if __name__ == "__main__":
    main()
