"""The latency floor of the segment search next to what the kernel achieves (profiles/r03_latency_floor.txt).

    python tools/latency_floor.py LAT_FLOOR_OUTPUT LABEL:DUMP:ALG_BYTES:WAVES_PER_GAP ...

LAT_FLOOR_OUTPUT: what tools/lat_floor.bin printed (cycles of one dependent 32-byte record load by the number of
waves on the chip).  DUMP: a G2S_DUMP_STATS file of one launch (per gap: rounds of phases A and B, shader cycles per
phase).  Every round of phase A and of phase B needs the records its addresses came out of the previous round's
records for: a gap cannot finish before rounds x (one dependent load); with two waves per gap phase A runs beside
phase B.  Phases D1/D2 and the emission work on LDS and are not in the floor (so it is a lower bound), and a launch
cannot end before its slowest gap does.  ALG_BYTES: SURVEY 8(d)'s algorithmic bytes of the launch (oracle counts),
which turns the floor into the largest fraction of the 8 TB/s roof this algorithm can show on that list."""
import sys

lat = {}
for ln in open(sys.argv[1]):
    p = ln.split()
    if len(p) == 5 and p[0].isdigit():
        lat[(int(p[0]), int(p[1]))] = (float(p[2]), float(p[3]), float(p[4]))
print("dependent record load (268 MB table), cycles / ns per step by waves on the chip (8 chains per wave):")
for (w, l), (c, ns, mhz) in sorted(lat.items()):
    if l == 8:
        print("  %6d waves: %7.0f cycles  %7.1f ns   (shader clock %.0f MHz)" % (w, c, ns, mhz))
for spec in sys.argv[2:]:
    label, path, alg, wpg = spec.split(":")
    alg, wpg = float(alg), int(wpg)
    rows = []
    for ln in open(path):
        if ln.startswith("#"):
            rows = []
            continue
        p = ln.split()
        rows.append([int(p[0]), int(p[1])] + [int(x, 16) if x.startswith("0x") else int(x) for x in p[2:]])
    n = len(rows)
    # columns: gap g flags A_rounds A_entries B_rounds segments cycA cycB 0 0 cycD ...
    tot = lambda r: r[7] + r[8] + r[11]
    rows.sort(key=tot, reverse=True)
    waves = min(lat, key=lambda k: (abs(k[0] - n * wpg), k[1] != 8))
    L, L_ns, mhz = lat[(waves[0], 8)]
    L1 = lat[(1, 8)][0]
    print("\n%s: %d gaps, %d wave(s) per gap; dependent load at this occupancy %.0f cycles (%.0f alone)" % (label, n, wpg, L, L1))
    for r in rows[:3]:
        ra, rb = r[3], r[5]
        rounds = max(ra, rb) if wpg == 2 else ra + rb
        # (two waves: the phase B wave's cycles include its wait for phase A; one wave: A, then B, then D)
        ach = (max(r[7], r[8]) if wpg == 2 else r[7] + r[8]) + r[11]
        print("  gap %5d (g %4d): rounds A %3d B %3d, %3d segments | achieved %7.0f k cycles = %6.1f us (A %d k, B %d k, D1+D2+emission %d k) | floor %3d rounds x %4.0f = %6.0f k cycles = %6.1f us (%.0f %% of achieved; %5.1f us with the unloaded latency)"
              % (r[0], r[1], ra, rb, r[6], ach / 1e3, ach / mhz, r[7] // 1000, r[8] // 1000, r[11] // 1000, rounds, L, rounds * L / 1e3,
                 rounds * L / mhz, 100.0 * rounds * L / ach, rounds * L1 / mhz))
    worst = rows[0]
    rounds = max(worst[3], worst[5]) if wpg == 2 else worst[3] + worst[5]
    floor_us = rounds * L / mhz
    print("  the launch cannot end before its slowest gap: >= %.1f us  ->  at most %.1f %% of the 8 TB/s roof on this list (%.1f MB algorithmic); the 40 %% target would need the launch to take %.1f us"
          % (floor_us, 100.0 * alg / (floor_us * 1e-6) / 8e12, alg / 1e6, alg / (0.4 * 8e12) * 1e6))
