#!/bin/bash
# Round-6 measurement artifacts (run on the GPU box from the repo root; ~15 min).  Order matters: the PMC passes come FIRST and their
# summaries are copied into profiles/ of the box's working copy, so that the default bench line that follows quotes
# `roofline.traffic` (headline) and `compliant.roofline.traffic` (bf16x3f) measured on THIS build's kernels.
# Summaries land in gpurun_out/r06/; copy them to profiles/ with the r06_ prefix.
# Two parts, each inside one gpurun call (<= 20 min): `tools/r06_artifacts.sh A` = PMC passes, kernel traces, the driver's own command and
# the compliant leg; `tools/r06_artifacts.sh B` = the hipBLASLt yardstick and the secondary lines (all with `parity`).
set -o pipefail
PART=${1:-A}
R=$PWD; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
if [ "$PART" = "A" ]; then
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
cp $O/pmc_hbm_traffic.json profiles/r06_pmc_hbm_traffic.json
cp $O/pmc_hbm_traffic_bf16x3f.json profiles/r06_pmc_hbm_traffic_bf16x3f.json
python3 tools/step_census.py $(find $O/ov -name "*kernel_trace.csv" | head -1) $O/step_census_overlapped.json > $O/step_census_overlapped.txt
cp $(find $O/ov -name "*kernel_stats.csv" | head -1) $O/kernel_stats_overlapped.csv
cp $(find $O/ser -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serialized.csv
cp $(find $O/x3f -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bf16x3f.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
# the driver's own command, now that profiles/ holds this build's PMC summaries
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
python3 bench.py --dtype bf16x3f --steps 24 --warmup 4 --no_cpu_baseline --parity > $O/bench_bf16x3f.json 2> $O/bench_bf16x3f.err || exit 1
ls -la $O; cat $O/bench_default.json
fi
if [ "$PART" = "B" ]; then
cd $R
# the vendor library on the nine ViT shapes of this round's build (a same-round yardstick), and the stream-K A/B
MMRCA_YARDSTICK=1 python3 tools/gemm_bench.py "fwd " "dgrad" "wgrad" "epi ffn1" > $O/gemm_yardstick_hipblaslt.txt 2>&1
# secondary lines (not the headline): configs[0]-shaped, configs[2], configs[3], configs[4] in both modes, the reference's default model, its launch
# shape, the frozen phase; cfg3 / cfg4 carry `parity`
python3 bench.py --image_model shuffle_net --batch 4 --steps 16 --warmup 4 --no_cpu_baseline --parity > $O/bench_cfg0.json 2>/dev/null
python3 bench.py --image_model eff_v2_large --text_model roberta --image_size 480 --batch 128 --steps 3 --warmup 1 --no_cpu_baseline --parity > $O/bench_cfg2.json 2>/dev/null
python3 bench.py --text_model bert --image_model transformer_L16 --cross_attention_only --seq_len 128 --batch 128 --steps 16 --warmup 4 --no_cpu_baseline --parity > $O/bench_cfg3.json 2>/dev/null
python3 bench.py --workload qformer --batch 64 --steps 8 --warmup 2 --no_cpu_baseline --parity > $O/bench_cfg4_qformer.json 2>/dev/null
python3 bench.py --workload qformer --dtype bf16x3f --batch 64 --steps 6 --warmup 2 --no_cpu_baseline --parity > $O/bench_cfg4_qformer_bf16x3f.json 2>/dev/null
python3 bench.py --image_model eff_v2_medium --image_size 480 --batch 64 --steps 8 --warmup 3 --no_cpu_baseline --parity > $O/bench_effv2m.json 2>/dev/null
python3 bench.py --image_model eff_v2_medium --image_size 480 --batch 64 --dtype bf16x3f --steps 8 --warmup 3 --no_cpu_baseline --parity > $O/bench_effv2m_bf16x3f.json 2>/dev/null
python3 bench.py --image_model eff_v2_medium --image_size 480 --batch 16 --seq_len 16 --dtype bf16x3f --steps 12 --warmup 4 --no_cpu_baseline --parity > $O/bench_reference_launch_shape_b16.json 2>/dev/null
python3 bench.py --image_model eff_v2_medium --image_size 480 --batch 16 --seq_len 16 --dtype bf16x3f --frozen --steps 12 --warmup 4 --no_cpu_baseline --parity > $O/bench_reference_launch_shape_b16_frozen.json 2>/dev/null
python3 bench.py --frozen --no_cpu_baseline --parity > $O/bench_frozen.json 2>/dev/null
for f in cfg0 cfg2 cfg3 cfg4_qformer cfg4_qformer_bf16x3f effv2m effv2m_bf16x3f reference_launch_shape_b16 reference_launch_shape_b16_frozen frozen; do python3 -c "import json,sys; d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['achieved'], (d.get('parity') or {}))"; done
fi
