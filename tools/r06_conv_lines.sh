#!/bin/bash
# Round 6, second part: the conv-backbone lines after the side-stream weight gradients (run on the GPU box from the repo root).
set -o pipefail
R=$PWD; O=$R/gpurun_out/r06b; mkdir -p $O
python3 bench.py --image_model eff_v2_large --text_model roberta --image_size 480 --batch 128 --steps 3 --warmup 1 --no_cpu_baseline --parity > $O/bench_cfg2.json 2>/dev/null
python3 bench.py --image_model eff_v2_medium --image_size 480 --batch 64 --steps 8 --warmup 3 --no_cpu_baseline --parity > $O/bench_effv2m.json 2>/dev/null
python3 bench.py --image_model eff_v2_medium --image_size 480 --batch 64 --dtype bf16x3f --steps 8 --warmup 3 --no_cpu_baseline --parity > $O/bench_effv2m_bf16x3f.json 2>/dev/null
python3 bench.py --image_model eff_v2_medium --image_size 480 --batch 16 --seq_len 16 --dtype bf16x3f --steps 12 --warmup 4 --no_cpu_baseline --parity > $O/bench_reference_launch_shape_b16.json 2>/dev/null
python3 bench.py --image_model eff_v2_medium --image_size 480 --batch 16 --seq_len 16 --dtype bf16x3f --graph on --steps 12 --warmup 4 --no_cpu_baseline > $O/bench_reference_launch_shape_b16_graph_on.json 2>/dev/null
python3 bench.py --image_model eff_v2_medium --image_size 480 --batch 16 --seq_len 16 --dtype bf16x3f --frozen --steps 12 --warmup 4 --no_cpu_baseline --parity > $O/bench_reference_launch_shape_b16_frozen.json 2>/dev/null
python3 bench.py --image_model shuffle_net --batch 4 --steps 16 --warmup 4 --no_cpu_baseline --parity > $O/bench_cfg0.json 2>/dev/null
for f in cfg2 effv2m effv2m_bf16x3f reference_launch_shape_b16 reference_launch_shape_b16_graph_on reference_launch_shape_b16_frozen cfg0; do python3 -c "import json,sys; d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], d['ms_per_step'], (d['roofline'].get('hbm') or {}).get('frac'), (d.get('parity') or {}), (d['config'].get('hip_graph') is not None))"; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b16 -o b16 -- python3 $R/bench.py --image_model eff_v2_medium --image_size 480 --batch 16 --seq_len 16 --dtype bf16x3f --steps 24 --warmup 4 --no_cpu_baseline > $O/b16_trace.log 2>&1
cd $R
cp $(find $O/b16 -name "*kernel_stats.csv" | head -1) $O/effv2m_b16_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 -c "import json; d=json.load(open('$O/bench_default.json')); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['compliant']['value'], d['compliant'].get('logits_rel'))"
