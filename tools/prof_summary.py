#!/usr/bin/env python
"""Summarise a rocprofv3 kernel_stats.csv: ms per step per kernel.  usage: prof_summary.py <dir> <steps_in_run> [rows]"""
import csv, glob, sys
f = sys.argv[1] if sys.argv[1].endswith(".csv") else (glob.glob(sys.argv[1] + "/*kernel_stats.csv") + glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[0]
n = int(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step:", round(tot / n / 1e6, 3))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print("%-62s calls=%5s ms/step=%8.3f avg_us=%9.1f" % (r["Name"][:62], r["Calls"], int(r["TotalDurationNs"]) / n / 1e6, float(r["AverageNs"]) / 1e3))
