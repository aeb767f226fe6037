#!/bin/bash
# Round-2 measurement artifacts (run on the GPU box from the repo root): default bench line, rocprofv3 kernel stats of the
# shipped (overlapped) and the one-stream (serialized) configuration, and the PMC passes (separate runs, per the guide).
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ov -o ov -- python3 $R/bench.py --steps 8 --warmup 3 > $O/ov.log 2>&1 || exit 1
export MMRCA_CONCURRENT_ENCODERS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ser -o ser -- python3 $R/bench.py --steps 8 --warmup 3 > $O/ser.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 > $O/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 2 --warmup 1 > $O/write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/mfma -o m -- python3 $R/bench.py --steps 2 --warmup 1 > $O/mfma.log 2>&1 || exit 1
cd $R
F=$(dirname $(find $O/fetch -name "*counter_collection.csv" | head -1)); W=$(dirname $(find $O/write -name "*counter_collection.csv" | head -1))
python3 tools/pmc_traffic.py $F $W $O/pmc_hbm_traffic.json
python3 tools/pmc_mfma_busy.py $O/mfma $O/pmc_mfma_busy.json
cp $(find $O/ov -name "*kernel_stats.csv" | head -1) $O/kernel_stats_overlapped.csv
cp $(find $O/ser -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serialized.csv
# keep the merged output small: drop the raw traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
ls -la $O; cat $O/bench_default.json
# secondary lines (not the headline): frozen phase, configs[3], configs[2], configs[0]-shaped run, fp32 mode
cd $R
python3 bench.py --frozen --no_cpu_baseline > $O/bench_frozen.json 2>/dev/null
python3 bench.py --text_model bert --image_model transformer_L16 --cross_attention_only --seq_len 128 --batch 128 --steps 16 --warmup 4 --no_cpu_baseline > $O/bench_cfg3.json 2>/dev/null
python3 bench.py --image_model eff_v2_large --image_size 480 --batch 128 --steps 3 --warmup 1 --no_cpu_baseline > $O/bench_cfg2.json 2>/dev/null
python3 bench.py --image_model shuffle_net --batch 4 --steps 16 --warmup 4 --no_cpu_baseline > $O/bench_cfg0.json 2>/dev/null
python3 bench.py --dtype fp32 --steps 6 --warmup 2 --no_cpu_baseline > $O/bench_fp32.json 2>/dev/null
for f in frozen cfg3 cfg2 cfg0 fp32; do python3 -c "import json,sys; d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['achieved'])"; done
