#!/bin/bash
# A/B of MMRCA_CONV_BN_FOLD (BatchNorm finish inside the apply pass); GPU box, repo root.
set -o pipefail
O=gpurun_out/fold_ab; mkdir -p $O
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 bench.py "$@" --no_cpu_baseline > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }
  python3 -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'])"; }
B16="--image_model eff_v2_medium --image_size 480 --batch 16 --seq_len 16 --dtype bf16x3f --steps 24 --warmup 6"
B64="--image_model eff_v2_medium --image_size 480 --batch 64 --dtype bf16x3f --steps 10 --warmup 3"
CFG2="--image_model eff_v2_large --text_model roberta --image_size 480 --batch 128 --steps 4 --warmup 2"
CFG0="--image_model shuffle_net --batch 4 --steps 32 --warmup 6"
run b16_fold0 MMRCA_CONV_BN_FOLD=0 -- $B16 &&
run b16_fold1 MMRCA_CONV_BN_FOLD=1 -- $B16 &&
run b16_fold0b MMRCA_CONV_BN_FOLD=0 -- $B16 &&
run b16_fold1b MMRCA_CONV_BN_FOLD=1 -- $B16 &&
run b64_fold0 MMRCA_CONV_BN_FOLD=0 -- $B64 &&
run b64_fold1 MMRCA_CONV_BN_FOLD=1 -- $B64 &&
run b64_fold0b MMRCA_CONV_BN_FOLD=0 -- $B64 &&
run b64_fold1b MMRCA_CONV_BN_FOLD=1 -- $B64 &&
run cfg0_fold0 MMRCA_CONV_BN_FOLD=0 -- $CFG0 &&
run cfg0_fold1 MMRCA_CONV_BN_FOLD=1 -- $CFG0 &&
run cfg0_fold0b MMRCA_CONV_BN_FOLD=0 -- $CFG0 &&
run cfg0_fold1b MMRCA_CONV_BN_FOLD=1 -- $CFG0 &&
run cfg2_fold0 MMRCA_CONV_BN_FOLD=0 -- $CFG2 &&
run cfg2_fold1 MMRCA_CONV_BN_FOLD=1 -- $CFG2
