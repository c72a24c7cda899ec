#!/bin/bash
# round 3, visit y: LoRA rank-space projection with 16 rows per workgroup (376 workgroups instead of 188)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q -x -k "lora or generate or decode" 2>&1 | tail -3 | tee $OUT/r3y_pytest.log
for i in 1 2 3; do
  for v in 32 16; do
    TA355_LORA_NT_ROWS=$v python bench.py --lora --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TA355_LORA_NT_ROWS=$v', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3y_ab_lora_nt_rows.txt
