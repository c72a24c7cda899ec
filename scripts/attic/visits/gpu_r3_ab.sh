#!/bin/bash
# round 3, visit ab: 256x320 tile with the 4 + 1 column map and row-merged stores
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
python scripts/store_merge_probe.py 2>&1 | tail -14 | tee $OUT/r3ab_store_merge_probe.txt
timeout 300 python scripts/gemm_repeat_check.py 2>&1 | tail -3 | tee $OUT/r3ab_repeat.txt
timeout 1200 python -m pytest tests -m gpu -q -x -k "gemm or b32 or smoke or full_depth or encoder or asr_model or moe or lora" 2>&1 | tail -4 | tee $OUT/r3ab_pytest.log
for i in 1 2 3; do
  for v in 2048 0; do
    TA355_GEMM_DEBUG=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TA355_GEMM_DEBUG=$v', d['ms_per_step'], d['value'])"
  done
  TA355_LIB=$REPO/tiny_audio_amd/libta355_prev.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('library of visit aa (pairs everywhere except 256-wide)', d['ms_per_step'], d['value'])"
done | tee $OUT/r3ab_ab_store_merge_320.txt
