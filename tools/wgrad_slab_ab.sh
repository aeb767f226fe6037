#!/bin/bash
# A/B of MMRCA_CONV_WGRAD_SLAB (conv 1x1 weight gradients on the ragged split-K slab kernel); GPU box, repo root.
set -o pipefail
O=gpurun_out/slab_ab; mkdir -p $O
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 bench.py "$@" --no_cpu_baseline > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }
  python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'])"; }
B16="--image_model eff_v2_medium --image_size 480 --batch 16 --seq_len 16 --dtype bf16x3f --steps 24 --warmup 6"
B64="--image_model eff_v2_medium --image_size 480 --batch 64 --dtype bf16x3f --steps 10 --warmup 3"
CFG2="--image_model eff_v2_large --text_model roberta --image_size 480 --batch 128 --steps 4 --warmup 2"
run b64_slab0 MMRCA_CONV_WGRAD_SLAB=0 -- $B64 &&
run b64_slab1 MMRCA_CONV_WGRAD_SLAB=1 -- $B64 &&
run b64_slab1_f30 MMRCA_CONV_WGRAD_SLAB=1 MMRCA_CONV_WGRAD_SLAB_MINFILL=0.3 -- $B64 &&
run b64_slab0b MMRCA_CONV_WGRAD_SLAB=0 -- $B64 &&
run b64_slab1b MMRCA_CONV_WGRAD_SLAB=1 -- $B64 &&
run b16_slab0 MMRCA_CONV_WGRAD_SLAB=0 -- $B16 &&
run b16_slab1 MMRCA_CONV_WGRAD_SLAB=1 -- $B16 &&
run b16_slab1_f30 MMRCA_CONV_WGRAD_SLAB=1 MMRCA_CONV_WGRAD_SLAB_MINFILL=0.3 -- $B16 &&
run b16_slab0b MMRCA_CONV_WGRAD_SLAB=0 -- $B16 &&
run b16_slab1b MMRCA_CONV_WGRAD_SLAB=1 -- $B16 &&
run cfg2_slab0 MMRCA_CONV_WGRAD_SLAB=0 -- $CFG2 &&
run cfg2_slab1 MMRCA_CONV_WGRAD_SLAB=1 -- $CFG2 &&
MMRCA_BENCH_SHAPES=1 python3 bench.py $B64 --no_cpu_baseline > $O/shapes_b64.json 2> $O/shapes_b64.err
