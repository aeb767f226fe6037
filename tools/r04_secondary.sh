#!/bin/bash
# Round-4 secondary bench lines (not the headline): frozen phase, configs[3] (bf16 / bf16x3f), configs[2] (with the byte roofline),
# the reference's default image model, a configs[0]-shaped run, fp32 mode, configs[4]
R=$PWD; O=$R/gpurun_out/r04; mkdir -p $O
python3 bench.py --frozen --no_cpu_baseline > $O/bench_frozen.json 2>/dev/null
python3 bench.py --text_model bert --image_model transformer_L16 --cross_attention_only --seq_len 128 --batch 128 --steps 16 --warmup 4 --no_cpu_baseline --no_compliant > $O/bench_cfg3.json 2>/dev/null
python3 bench.py --text_model bert --image_model transformer_L16 --cross_attention_only --seq_len 128 --batch 128 --dtype bf16x3f --steps 8 --warmup 2 --no_cpu_baseline > $O/bench_cfg3_bf16x3f.json 2>/dev/null
python3 bench.py --image_model eff_v2_large --image_size 480 --text_model roberta --batch 128 --steps 6 --warmup 2 > $O/bench_cfg2.json 2>/dev/null
python3 bench.py --image_model shuffle_net --batch 4 --steps 16 --warmup 4 --no_cpu_baseline > $O/bench_cfg0.json 2>/dev/null
python3 bench.py --image_model eff_v2_medium --image_size 480 --batch 64 --steps 6 --warmup 2 --no_cpu_baseline > $O/bench_effv2m.json 2>/dev/null
python3 bench.py --dtype fp32 --steps 6 --warmup 2 --no_cpu_baseline > $O/bench_fp32.json 2>/dev/null
python3 bench.py --workload qformer --batch 64 --steps 8 --warmup 2 --no_cpu_baseline > $O/bench_cfg4_qformer.json 2>/dev/null
# kernel traces of the two conv-backbone configs (eager launches: a graph replay shows up as one opaque launch per node anyway)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/p0 -o cfg0 -- python3 $R/bench.py --image_model shuffle_net --batch 4 --steps 20 --warmup 5 --graph off --no_cpu_baseline --no_compliant > $O/cfg0_prof.log 2>&1 )
cp $(find $O/p0 -name "*kernel_stats.csv" | head -1) $O/bench_cfg0_kernel_stats.csv
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2 -o cfg2 -- python3 $R/bench.py --image_model eff_v2_large --image_size 480 --text_model roberta --batch 128 --steps 6 --warmup 2 --no_cpu_baseline --no_compliant > $O/cfg2_prof.log 2>&1 )
cp $(find $O/p2 -name "*kernel_stats.csv" | head -1) $O/bench_cfg2_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete
for f in frozen cfg3 cfg3_bf16x3f cfg2 cfg0 effv2m fp32 cfg4_qformer; do python3 -c "import json,sys; d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['achieved'])"; done
