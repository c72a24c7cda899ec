#!/bin/bash
# round 4, visit e: gemm_v7 inside the B = 32 step, per shape family (TA355_V7_MASK), against the ping-pong kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_e
for i in 1 2; do
  for m in 0 1 2 4 8 16 3; do
    TA355_V7_MASK=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TA355_V7_MASK=$m', d['ms_per_step'], d['value'])"
  done
done 2>&1 | tee gpurun_out/r04_e/ab_v7_mask.txt
