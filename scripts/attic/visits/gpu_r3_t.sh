#!/bin/bash
# round 3, visit t: what bounds the LoRA TN (adapter gradient) launches -- rows per workgroup sweep (atomics vs parallelism)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
for i in 1 2; do
  for v in 0 96 192 576 1152 3008; do
    TA355_LORA_TN_ROWS=$v python bench.py --lora --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TA355_LORA_TN_ROWS=$v', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3t_ab_lora_tn_rows.txt
