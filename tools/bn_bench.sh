#!/bin/bash
# usage (on the GPU box): tools/bn_bench.sh <tag>   -> gpurun_out/bn_bench_<tag>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/bn_bench_$1.txt; : > $out
for shape in "115200 1536" "28800 2304" "460800 384" "115200 768"; do
  set -- $shape
  rm -rf gpurun_out/bnb
  rocprofv3 --kernel-trace --stats -d gpurun_out/bnb -o b --output-format csv -- python3 tools/bn_bench.py $1 $2 >> $out 2>/dev/null
  python3 - "$1" "$2" >> $out <<'P'
import csv, sys
rows, C = int(sys.argv[1]), int(sys.argv[2]); unit = rows * C * 2
mult = {"col_moment2": 1, "bn_act_fwd": 2, "bn_act_bwd_reduce": 2, "bn_act_bwd_apply": 3, "bn_flat_reduce_k": 0}
for r in csv.DictReader(open("gpurun_out/bnb/b_kernel_stats.csv")):
    for k, m in mult.items():
        if k in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print(f"   {r['Name'][:44]:44s} {us:8.1f} us  {m * unit / us / 1e6:6.2f} TB/s")
P
done
rm -rf gpurun_out/bnb
cat $out
