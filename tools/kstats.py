#!/usr/bin/env python
"""Per-step view of a rocprofv3 kernel_stats.csv.   usage: kstats.py <kernel_stats.csv> <steps in the trace> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(int(r['TotalDurationNs']) for r in rows)
calls = sum(int(r['Calls']) for r in rows)
print(f"kernel time {tot / 1e6 / n:.3f} ms/step, {calls / n:.1f} launches/step")
for r in rows[:top]:
    print(f"{int(r['TotalDurationNs']) / 1e6 / n:7.3f} ms/step {int(r['Calls']) / n:6.1f} calls  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:100]}")
