#!/bin/bash
# round 3, visit u: LoRA TN pair -- rows per workgroup chosen for a target workgroup count over both problems
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q -x -k "lora" 2>&1 | tail -3 | tee $OUT/r3u_pytest.log
for i in 1 2; do
  for setting in "TA355_LORA_TN_ROWS=576" "TA355_LORA_TN_WGS=192" "TA355_LORA_TN_WGS=256" "TA355_LORA_TN_WGS=320" "TA355_LORA_TN_WGS=384" "TA355_LORA_TN_WGS=512"; do
    env $setting python bench.py --lora --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$setting', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3u_ab_lora_tn_wgs.txt
