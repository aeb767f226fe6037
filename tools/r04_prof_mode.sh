#!/bin/bash
# rocprofv3 kernel stats of one bench.py mode on ONE stream (serialized).  usage: tools/r04_prof_mode.sh <dtype> <tag> [extra bench args]
set -o pipefail
DT=$1; TAG=$2; shift 2
R=$PWD; O=$R/gpurun_out/r04; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export MMRCA_CONCURRENT_ENCODERS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/$TAG -o $TAG -- python3 $R/bench.py --dtype $DT --steps 6 --warmup 2 --no_cpu_baseline "$@" > $O/$TAG.log 2>&1 || exit 1
cd $R
cp $(find $O/$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
find $O/$TAG -name "*kernel_trace.csv" -delete; find $O/$TAG -name "*.db" -delete
python3 tools/prof_summary.py $O/${TAG}_kernel_stats.csv 13 40
