#!/bin/bash
# Round-5 measurement artifacts (run on the GPU box from the repo root; ~15 min).  Order matters: the PMC passes come FIRST and their
# summaries are copied into profiles/ of the box's working copy, so that the default bench line that follows quotes
# `roofline.traffic` (headline) and `compliant.roofline.traffic` (bf16x3f) measured on THIS build's kernels.
# Summaries land in gpurun_out/r05/; copy them to profiles/ with the r05_ prefix.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export MMRCA_CONCURRENT_ENCODERS=0
B="--steps 2 --warmup 1 --no_cpu_baseline --no_compliant"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py $B > $O/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py $B > $O/write.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch3 -o f -- python3 $R/bench.py --dtype bf16x3f $B > $O/fetch3.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write3 -o w -- python3 $R/bench.py --dtype bf16x3f $B > $O/write3.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/mfma -o m -- python3 $R/bench.py $B > $O/mfma.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/mfma_x3f -o m -- python3 $R/bench.py --dtype bf16x3f $B > $O/mfma_x3f.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ser -o ser -- python3 $R/bench.py --steps 8 --warmup 3 --no_cpu_baseline --no_compliant > $O/ser.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/x3f -o x3f -- python3 $R/bench.py --dtype bf16x3f --steps 6 --warmup 2 --no_cpu_baseline > $O/x3f.log 2>&1 || exit 1
unset MMRCA_CONCURRENT_ENCODERS
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ov -o ov -- python3 $R/bench.py --steps 8 --warmup 3 --no_cpu_baseline --no_compliant > $O/ov.log 2>&1 || exit 1
cd $R
F=$(dirname $(find $O/fetch -name "*counter_collection.csv" | head -1)); W=$(dirname $(find $O/write -name "*counter_collection.csv" | head -1))
python3 tools/pmc_traffic.py $F $W $O/pmc_hbm_traffic.json
F=$(dirname $(find $O/fetch3 -name "*counter_collection.csv" | head -1)); W=$(dirname $(find $O/write3 -name "*counter_collection.csv" | head -1))
python3 tools/pmc_traffic.py $F $W $O/pmc_hbm_traffic_bf16x3f.json
python3 tools/pmc_mfma_busy.py $O/mfma $O/pmc_mfma_busy.json
python3 tools/pmc_mfma_busy.py $O/mfma_x3f $O/pmc_mfma_busy_bf16x3f.json
cp $O/pmc_hbm_traffic.json profiles/r05_pmc_hbm_traffic.json
cp $O/pmc_hbm_traffic_bf16x3f.json profiles/r05_pmc_hbm_traffic_bf16x3f.json
python3 tools/step_census.py $(find $O/ov -name "*kernel_trace.csv" | head -1) $O/step_census_overlapped.json > $O/step_census_overlapped.txt
cp $(find $O/ov -name "*kernel_stats.csv" | head -1) $O/kernel_stats_overlapped.csv
cp $(find $O/ser -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serialized.csv
cp $(find $O/x3f -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bf16x3f.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
# the driver's own command, now that profiles/ holds this build's PMC summaries
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
python3 bench.py --dtype bf16x3f --steps 24 --warmup 4 --no_cpu_baseline --parity > $O/bench_bf16x3f.json 2> $O/bench_bf16x3f.err || exit 1
ls -la $O; cat $O/bench_default.json
