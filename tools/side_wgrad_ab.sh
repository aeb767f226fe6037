#!/bin/bash
# A/B of MMRCA_CONV_SIDE_WGRAD (conv weight gradients on a side stream) on the EfficientNetV2-M steps; run on the GPU box from the repo root.
set -o pipefail
O=gpurun_out/side_ab; mkdir -p $O
run() { # name env... -- bench args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 bench.py "$@" --no_cpu_baseline > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }
  python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'])"
}
B16="--image_model eff_v2_medium --image_size 480 --batch 16 --seq_len 16 --dtype bf16x3f --steps 24 --warmup 6"
B64="--image_model eff_v2_medium --image_size 480 --batch 64 --dtype bf16x3f --steps 10 --warmup 3"
CFG2="--image_model eff_v2_large --text_model roberta --image_size 480 --batch 128 --steps 4 --warmup 2"
if [ "$1" = "first" ]; then
run b16_off_1 MMRCA_CONV_SIDE_WGRAD=0 -- $B16 &&
run b16_on_all_1 MMRCA_CONV_SIDE_WGRAD=1 -- $B16 &&
run b16_on_14400 MMRCA_CONV_SIDE_WGRAD=1 MMRCA_CONV_SIDE_MAXROWS=14400 -- $B16 &&
run b16_on_3600 MMRCA_CONV_SIDE_WGRAD=1 MMRCA_CONV_SIDE_MAXROWS=3600 -- $B16 &&
run b16_off_2 MMRCA_CONV_SIDE_WGRAD=0 -- $B16 &&
run b16_on_all_2 MMRCA_CONV_SIDE_WGRAD=1 -- $B16 &&
run b64_off MMRCA_CONV_SIDE_WGRAD=0 -- $B64 &&
run b64_on_all MMRCA_CONV_SIDE_WGRAD=1 -- $B64 &&
run b64_on_57600 MMRCA_CONV_SIDE_WGRAD=1 MMRCA_CONV_SIDE_MAXROWS=57600 -- $B64 &&
run b64_on_14400 MMRCA_CONV_SIDE_WGRAD=1 MMRCA_CONV_SIDE_MAXROWS=14400 -- $B64
else
run s_b64_off MMRCA_CONV_SIDE_WGRAD=0 -- $B64 &&
run s_b64_auto MMRCA_CONV_SIDE_WGRAD=auto -- $B64 &&
run s_b64_off2 MMRCA_CONV_SIDE_WGRAD=0 -- $B64 &&
run s_b64_auto2 MMRCA_CONV_SIDE_WGRAD=auto -- $B64 &&
run s_cfg2_off MMRCA_CONV_SIDE_WGRAD=0 -- $CFG2 &&
run s_cfg2_auto MMRCA_CONV_SIDE_WGRAD=auto -- $CFG2 &&
run s_cfg2_off2 MMRCA_CONV_SIDE_WGRAD=0 -- $CFG2 &&
run s_cfg2_auto2 MMRCA_CONV_SIDE_WGRAD=auto -- $CFG2 &&
run s_b16_graph_auto MMRCA_CONV_SIDE_WGRAD=auto -- $B16 &&
run s_b16_eager_off MMRCA_CONV_SIDE_WGRAD=0 -- $B16 --graph off &&
run s_b16_eager_on MMRCA_CONV_SIDE_WGRAD=1 -- $B16 --graph off &&
run s_b64_bf16_off MMRCA_CONV_SIDE_WGRAD=0 -- --image_model eff_v2_medium --image_size 480 --batch 64 --steps 10 --warmup 3 &&
run s_b64_bf16_auto MMRCA_CONV_SIDE_WGRAD=auto -- --image_model eff_v2_medium --image_size 480 --batch 64 --steps 10 --warmup 3
fi
