#!/usr/bin/env python
"""Kernels of ONE steady-state train step from a rocprofv3 --kernel-trace CSV: everything between two consecutive optimizer
launches (sgd_k) late in the run, so that model construction, warm-up and the parity / cpu-baseline legs of bench.py do not
pollute the per-step counts (round 2's "96 copies per step" was calls / steps over the whole process).
usage: step_census.py <kernel_trace.csv> [out.json]"""
import collections, csv, json, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
opt = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(("sgd_k", "adamw_k"))]
assert len(opt) >= 3, "need at least three optimizer launches in the trace"
a, b = opt[-2], opt[-1]
cnt, dur = collections.Counter(), collections.Counter()
for r in rows[a + 1:b + 1]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()[:100]
    cnt[n] += 1
    dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
native = {n: (c, dur[n] / 1e3) for n, c in cnt.items() if n.startswith(("at::", "__amd_rocclr"))}
out = {"launches_in_step": sum(cnt.values()), "kernel_time_ms": round(sum(dur.values()) / 1e6, 3),
       "torch_native_launches": {n: {"calls": c, "us": round(u, 1)} for n, (c, u) in sorted(native.items(), key=lambda kv: -kv[1][1])},
       "torch_native_total": {"calls": sum(c for c, _ in native.values()), "us": round(sum(u for _, u in native.values()), 1)},
       "top": [{"kernel": n, "calls": cnt[n], "ms": round(dur[n] / 1e6, 3)} for n, _ in dur.most_common(24)]}
print(json.dumps({k: v for k, v in out.items() if k != "top"}, indent=1))
for t in out["top"]:
    print(f"{t['kernel'][:90]:90s} {t['calls']:5d} {t['ms']:9.3f} ms")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
