#!/bin/bash
# Final bench lines of round 3 (after tools/r03_artifacts.sh left profiles/r03_pmc_hbm_traffic.json for these GEMM sources)
R=$PWD; O=$R/gpurun_out/r03f; mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
python3 bench.py --dtype bf16x3 --steps 16 --warmup 4 > $O/bench_bf16x3.json 2>/dev/null
python3 bench.py --image_model eff_v2_large --image_size 480 --text_model roberta --batch 128 --steps 6 --warmup 2 > $O/bench_cfg2.json 2>/dev/null
python3 bench.py --frozen --no_cpu_baseline > $O/bench_frozen.json 2>/dev/null
python3 bench.py --text_model bert --image_model transformer_L16 --cross_attention_only --seq_len 128 --batch 128 --steps 16 --warmup 4 --no_cpu_baseline > $O/bench_cfg3.json 2>/dev/null
python3 bench.py --text_model bert --image_model transformer_L16 --cross_attention_only --seq_len 128 --batch 128 --dtype bf16x3 --steps 6 --warmup 2 --no_cpu_baseline > $O/bench_cfg3_bf16x3.json 2>/dev/null
python3 bench.py --image_model shuffle_net --batch 4 --steps 16 --warmup 4 --no_cpu_baseline > $O/bench_cfg0.json 2>/dev/null
python3 bench.py --image_model eff_v2_medium --image_size 480 --batch 64 --steps 6 --warmup 2 --no_cpu_baseline > $O/bench_effv2m.json 2>/dev/null
python3 bench.py --dtype fp32 --steps 6 --warmup 2 --no_cpu_baseline > $O/bench_fp32.json 2>/dev/null
python3 bench.py --workload qformer --batch 64 --steps 8 --warmup 2 --no_cpu_baseline > $O/bench_cfg4_qformer.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg2prof -o c -- python3 $R/bench.py --image_model eff_v2_large --image_size 480 --text_model roberta --batch 128 --steps 4 --warmup 2 --no_cpu_baseline > $O/cfg2prof.log 2>&1
cd $R
cp $(find $O/cfg2prof -name "*kernel_stats.csv" | head -1) $O/cfg2_kernel_stats.csv; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
for f in default bf16x3 cfg2 frozen cfg3 cfg3_bf16x3 cfg0 effv2m fp32 cfg4_qformer; do python3 -c "import json; d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline'].get('traffic'))"; done
