#!/bin/bash
# round 3, visit r: DMA offsets from a fresh thread index -- default bench A/B against the library of visit q; LoRA with the K extension on the persistent kernel
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -x -k "gemm or lora or b32 or smoke" 2>&1 | tail -4 | tee $OUT/r3r_pytest.log
echo "== default bench: library of visit q | new"
for i in 1 2 3; do
  for lib in libta355_prev.so libta355.so; do
    TA355_LIB=$REPO/tiny_audio_amd/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3r_ab_fresh_tid.txt
echo "== bench.py --lora: K extension on v2 | on the persistent kernel"
for i in 1 2 3; do
  for v in 0 1; do
    TA355_GEMM_PERSIST_KEXT=$v python bench.py --lora --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TA355_GEMM_PERSIST_KEXT=$v', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3r_ab_lora_persist_kext.txt
