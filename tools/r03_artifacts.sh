#!/bin/bash
# Round-3 measurement artifacts (run on the GPU box from the repo root; ~15 min): the default bench line (bf16, configs[1]), the
# bf16x3 line, rocprofv3 kernel stats of the shipped (overlapped) and the one-stream (serialized) configuration, one step's kernel
# census, and the PMC passes (separate runs, per the MI355X guide).  Summaries land in gpurun_out/r03/; copy them to profiles/.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
python3 $R/bench.py --dtype bf16x3 --steps 16 --warmup 4 > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ov -o ov -- python3 $R/bench.py --steps 8 --warmup 3 --no_cpu_baseline > $O/ov.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/x3 -o x3 -- python3 $R/bench.py --dtype bf16x3 --steps 4 --warmup 2 --no_cpu_baseline > $O/x3.log 2>&1 || exit 1
export MMRCA_CONCURRENT_ENCODERS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ser -o ser -- python3 $R/bench.py --steps 8 --warmup 3 --no_cpu_baseline > $O/ser.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline > $O/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline > $O/write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/mfma -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline > $O/mfma.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/mfma_x3 -o m -- python3 $R/bench.py --dtype bf16x3 --steps 2 --warmup 1 --no_cpu_baseline > $O/mfma_x3.log 2>&1 || exit 1
unset MMRCA_CONCURRENT_ENCODERS
cd $R
F=$(dirname $(find $O/fetch -name "*counter_collection.csv" | head -1)); W=$(dirname $(find $O/write -name "*counter_collection.csv" | head -1))
python3 tools/pmc_traffic.py $F $W $O/pmc_hbm_traffic.json
python3 tools/pmc_mfma_busy.py $O/mfma $O/pmc_mfma_busy.json
python3 tools/pmc_mfma_busy.py $O/mfma_x3 $O/pmc_mfma_busy_bf16x3.json
python3 tools/step_census.py $(find $O/ov -name "*kernel_trace.csv" | head -1) $O/step_census_overlapped.json > $O/step_census_overlapped.txt
python3 tools/step_census.py $(find $O/x3 -name "*kernel_trace.csv" | head -1) $O/step_census_bf16x3.json > $O/step_census_bf16x3.txt
cp $(find $O/ov -name "*kernel_stats.csv" | head -1) $O/kernel_stats_overlapped.csv
cp $(find $O/ser -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serialized.csv
cp $(find $O/x3 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bf16x3.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
ls -la $O; cat $O/bench_default.json; cat $O/bench_bf16x3.json
