"""Per-kernel summary of a rocprofv3 *_kernel_stats.csv (names shortened)."""
import csv
import re
import sys

for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(g2s_[a-z0-9_]+|k_[a-z_]+|rocclr_[A-Za-z]+)", r["Name"])
    print((m.group(1) if m else r["Name"][:40]).ljust(28), r["Calls"].rjust(6), ("%.1f us avg" % (float(r["AverageNs"]) / 1e3)).rjust(14),
          ("%.1f%%" % float(r["Percentage"])).rjust(7))
