#!/usr/bin/env python3
"""Print per-kernel average durations from a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, sys
d = sys.argv[1] if len(sys.argv) > 1 else "."
for f in sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True)):
    print("#", f)
    for r in csv.DictReader(open(f)):
        print("%-44s calls %4s  avg %10.1f us  total %6.2f%%" % (r["Name"].split("(")[0][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
