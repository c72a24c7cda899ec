#!/bin/bash
# round 3, visit w: which half of the three-interval DMA schedule costs time -- TA355_GEMM_DEBUG 0 | 256 (leading group waits late) | 512 (lagging group requests early) | 768
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
for i in 1 2; do
  for v in 0 256 512 768; do
    TA355_GEMM_DEBUG=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TA355_GEMM_DEBUG=$v', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3w_ab_dma_schedule.txt
