#!/bin/bash
# round 3, visit aa: row-merged stores on the 256-column ping-pong tiles
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
python scripts/store_merge_probe.py 2>&1 | tail -8 | tee $OUT/r3aa_store_merge_probe.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "gemm or b32 or smoke" 2>&1 | tail -4 | tee $OUT/r3aa_pytest.log
for i in 1 2 3; do
  for v in 2048 0; do
    TA355_GEMM_DEBUG=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TA355_GEMM_DEBUG=$v', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3aa_ab_store_merge.txt
