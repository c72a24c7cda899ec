#!/bin/bash
# round 3, visit am: row-merged full-tile stores also in the one-tile-per-CU kernel (LM N = 1024 products) and the non-persistent ping-pong kernel (LoRA K extension)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 300 python scripts/gemm_repeat_check.py 2>&1 | tail -2 | tee $OUT/r3am_repeat.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "gemm or b32 or smoke or lora" 2>&1 | tail -3 | tee $OUT/r3am_pytest.log
for i in 1 2 3; do
  for lib in libta355_prev.so libta355.so; do
    TA355_LIB=$REPO/tiny_audio_amd/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mlp $lib', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3am_ab_store_merge_v5_v2.txt
for i in 1 2; do
  for lib in libta355_prev.so libta355.so; do
    TA355_LIB=$REPO/tiny_audio_amd/$lib python bench.py --lora --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lora $lib', d['ms_per_step'], d['value'])"
  done
done | tee -a $OUT/r3am_ab_store_merge_v5_v2.txt
