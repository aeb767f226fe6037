#!/bin/bash
# bf16x3 mode: gradient accuracy (float64 oracle) and speed for every backward pass set
for cfg in "3 3" "3 2" "2 3" "2 2" "1 2" "2 1" "1 1"; do
  set -- $cfg
  echo "=== MMRCA_X3_WGRAD_PASSES=$1 MMRCA_X3_DGRAD_PASSES=$2"
  MMRCA_X3_WGRAD_PASSES=$1 MMRCA_X3_DGRAD_PASSES=$2 timeout -k 10 200 python tools/x3_grad_error.py 2>&1 | grep -E "logits|Error" | cut -c1-220
  MMRCA_X3_WGRAD_PASSES=$1 MMRCA_X3_DGRAD_PASSES=$2 timeout -k 10 200 python3 bench.py --dtype bf16x3 --steps 8 --warmup 3 --no_cpu_baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
done
