#!/bin/bash
# round 3, visit ag: row-merged stores for FULL tiles only (lean form) -- vs the previous library and vs its own base path
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 300 python scripts/gemm_repeat_check.py 2>&1 | tail -3 | tee $OUT/r3ag_repeat.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "gemm or b32 or smoke or full_depth" 2>&1 | tail -3 | tee $OUT/r3ag_pytest.log
for i in 1 2 3; do
  TA355_LIB=$REPO/tiny_audio_amd/libta355_prev.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('previous library', d['ms_per_step'], d['value'])"
  for v in 2048 0; do
    TA355_GEMM_DEBUG=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new library, TA355_GEMM_DEBUG=$v', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3ag_ab_store_merge_full.txt
