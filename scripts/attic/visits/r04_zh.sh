#!/bin/bash
# round 4, visit zh: LoRA adapter gradients through per-chunk partial sums + one reduce launch (no float atomics): parity, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_zh
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round2.py -x -q -k "lora" 2>&1 | tail -4
for i in 1 2 3; do
  for v in 1 0; do
    TA355_LORA_TN_PARTS=$v python bench.py --lora --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TA355_LORA_TN_PARTS=$v', d['ms_per_step'], d['value'])"
  done
done | tee gpurun_out/r04_zh/ab_lora_tn_parts.txt
for w in 256 512 768; do
  TA355_LORA_TN_WGS=$w python bench.py --lora --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('parts, TA355_LORA_TN_WGS=$w', d['ms_per_step'], d['value'])"
done | tee -a gpurun_out/r04_zh/ab_lora_tn_parts.txt
