#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "grouped or moe" 2>&1 | tail -30
for g in 1 0 1 0; do
  TA355_MOE_GROUPED=$g timeout 300 python bench.py --projector moe --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('moe grouped=$g', d['ms_per_step'], d['value'], d['final_loss'])"
done | tee gpurun_out/r2d_moe_ab.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mlp', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r2d_moe_ab.txt
