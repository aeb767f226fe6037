#!/usr/bin/env python
"""Per-kernel MFMA utilisation from one rocprofv3 PMC pass: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (both summed over the
shader engines by rocprofv3; the ratio is what the MI355X guide calls MFMA busy).
usage: pmc_mfma_busy.py <dir of the `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES` pass> <out.json>"""
import collections, csv, glob, json, sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CYCLES":
        cnt[k] += 1
out = {}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
    busy, mf = v.get("SQ_BUSY_CYCLES", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if busy > 0 and mf > 0:
        out[k] = {"launches_seen": cnt[k], "mfma_busy_cycles_per_launch": round(mf / max(cnt[k], 1)),
                  "sq_busy_cycles_per_launch": round(busy / max(cnt[k], 1)), "mfma_busy_over_sq_busy": round(mf / busy, 4)}
json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- python3 bench.py --steps 2 --warmup 1, MMRCA_CONCURRENT_ENCODERS=0",
           "kernels": out}, open(sys.argv[2], "w"), indent=1)
