#!/bin/bash
# round 4, visit n: LoRA adapter-gradient products on a side stream -- parity tests + in-step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_n
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -k "lora" 2>&1 | tail -5 > gpurun_out/r04_n/pytest.log
tail -3 gpurun_out/r04_n/pytest.log
for i in 1 2 3; do
  for v in 1 0; do
    TA355_LORA_SIDE_STREAM=$v python bench.py --lora --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TA355_LORA_SIDE_STREAM=$v', d['ms_per_step'], d['value'])"
  done
done 2>&1 | tee gpurun_out/r04_n/ab_lora_side_stream.txt
