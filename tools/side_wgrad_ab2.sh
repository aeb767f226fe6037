#!/bin/bash
# A/B of MMRCA_CONV_SIDE_DW (depthwise weight gradients on the side stream too); GPU box, repo root.
set -o pipefail
O=gpurun_out/side_ab; mkdir -p $O
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 bench.py "$@" --no_cpu_baseline > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }
  python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'])"; }
B16="--image_model eff_v2_medium --image_size 480 --batch 16 --seq_len 16 --dtype bf16x3f --steps 24 --warmup 6"
B64="--image_model eff_v2_medium --image_size 480 --batch 64 --dtype bf16x3f --steps 10 --warmup 3"
run d_b16_dw0 MMRCA_CONV_SIDE_DW=0 -- $B16 &&
run d_b16_dw1 MMRCA_CONV_SIDE_DW=1 -- $B16 &&
run d_b16_dw0b MMRCA_CONV_SIDE_DW=0 -- $B16 &&
run d_b16_dw1b MMRCA_CONV_SIDE_DW=1 -- $B16 &&
run d_b64_dw0 MMRCA_CONV_SIDE_DW=0 -- $B64 &&
run d_b64_dw1 MMRCA_CONV_SIDE_DW=1 -- $B64 &&
run d_b64_dw0b MMRCA_CONV_SIDE_DW=0 -- $B64 &&
run d_b64_dw1b MMRCA_CONV_SIDE_DW=1 -- $B64
