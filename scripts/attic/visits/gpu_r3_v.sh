#!/bin/bash
# round 3, visit v: three barrier intervals per DMA in the ping-pong kernel -- race hunt, GEMM + model tests, A/B vs the previous library
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 300 python scripts/gemm_repeat_check.py 2>&1 | tail -5 | tee $OUT/r3v_repeat.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "gemm or b32 or smoke or full_depth or encoder or asr_model" 2>&1 | tail -4 | tee $OUT/r3v_pytest.log
for i in 1 2 3; do
  for lib in libta355_prev.so libta355.so; do
    TA355_LIB=$REPO/tiny_audio_amd/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3v_ab_dma_intervals.txt
