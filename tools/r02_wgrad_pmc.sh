#!/bin/bash
# wgrad-only PMC check: FETCH_SIZE / WRITE_SIZE of the split-K weight-gradient kernel on the ViT shapes (tools/gemm_bench.py wgrad)
set -o pipefail
R=$PWD; O=$R/gpurun_out/wg; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/tools/gemm_bench.py wgrad > $O/f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/tools/gemm_bench.py wgrad > $O/w.log 2>&1 || exit 1
cd $R
F=$(dirname $(find $O/fetch -name "*counter_collection.csv" | head -1)); W=$(dirname $(find $O/write -name "*counter_collection.csv" | head -1))
python3 tools/pmc_traffic.py $F $W $O/wgrad_traffic.json
python3 - <<'PY'
import json
t = json.load(open("gpurun_out/wg/wgrad_traffic.json"))
for k, v in t["kernels"].items():
    if "gemm" in k or "splitk" in k: print(k[:50], v)
PY
find $O -name "*counter_collection.csv" -delete
