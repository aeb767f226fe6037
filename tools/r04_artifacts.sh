#!/bin/bash
# Round-4 measurement artifacts (run on the GPU box from the repo root; ~15 min): the default bench line (bf16 headline + the
# `compliant` leg in bf16x3f), the bf16x3f / bf16x3 lines, rocprofv3 kernel stats of the shipped (overlapped) and the one-stream
# (serialized) configuration of the headline and of bf16x3f, and the PMC passes (separate runs, per the MI355X guide).
# Summaries land in gpurun_out/r04/; copy them to profiles/ with the r04_ prefix.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
python3 $R/bench.py --dtype bf16x3f --steps 24 --warmup 4 > $O/bench_bf16x3f.json 2> $O/bench_bf16x3f.err || exit 1
python3 $R/bench.py --dtype bf16x3 --steps 16 --warmup 4 --no_cpu_baseline > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ov -o ov -- python3 $R/bench.py --steps 8 --warmup 3 --no_cpu_baseline --no_compliant > $O/ov.log 2>&1 || exit 1
export MMRCA_CONCURRENT_ENCODERS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ser -o ser -- python3 $R/bench.py --steps 8 --warmup 3 --no_cpu_baseline --no_compliant > $O/ser.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/x3f -o x3f -- python3 $R/bench.py --dtype bf16x3f --steps 6 --warmup 2 --no_cpu_baseline > $O/x3f.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_compliant > $O/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_compliant > $O/write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/mfma -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_compliant > $O/mfma.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/mfma_x3f -o m -- python3 $R/bench.py --dtype bf16x3f --steps 2 --warmup 1 --no_cpu_baseline > $O/mfma_x3f.log 2>&1 || exit 1
unset MMRCA_CONCURRENT_ENCODERS
cd $R
F=$(dirname $(find $O/fetch -name "*counter_collection.csv" | head -1)); W=$(dirname $(find $O/write -name "*counter_collection.csv" | head -1))
python3 tools/pmc_traffic.py $F $W $O/pmc_hbm_traffic.json
python3 tools/pmc_mfma_busy.py $O/mfma $O/pmc_mfma_busy.json
python3 tools/pmc_mfma_busy.py $O/mfma_x3f $O/pmc_mfma_busy_bf16x3f.json
python3 tools/step_census.py $(find $O/ov -name "*kernel_trace.csv" | head -1) $O/step_census_overlapped.json > $O/step_census_overlapped.txt
cp $(find $O/ov -name "*kernel_stats.csv" | head -1) $O/kernel_stats_overlapped.csv
cp $(find $O/ser -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serialized.csv
cp $(find $O/x3f -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bf16x3f.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
ls -la $O; cat $O/bench_default.json
