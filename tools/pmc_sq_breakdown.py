#!/usr/bin/env python
"""Where the waves of a kernel spend their cycles, from one rocprofv3 PMC pass over SQ counters (quad-cycle units, disjoint
buckets per the MI355X guide: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~= WAVE_CYCLES).
usage: pmc_sq_breakdown.py <dir of the pass> [kernel-name filter]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
    if flt not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, v in agg.items():
    wc = v.get("SQ_WAVE_CYCLES", 1.0)
    print(k[:70], "launches", cnt[k])
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {x / max(cnt[k], 1):14.0f} per launch   {x / wc:6.3f} of WAVE_CYCLES")
