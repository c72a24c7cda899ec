#!/bin/bash
# round 4, visit i: bf16 d(x) stream in the LM backward -- parity at full depth (MLP, LoRA), model-level tests, in-step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_i
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q -k "full_depth or b32_step or lm_ or asr_model or train" 2>&1 | tail -8 > gpurun_out/r04_i/pytest.log
tail -5 gpurun_out/r04_i/pytest.log
cp gpurun_out/r03_full_depth_drift.json gpurun_out/r04_i/full_depth_drift_dx_bf16.json 2>/dev/null
for i in 1 2 3; do
  for v in 0 1; do
    TA355_LM_DX_F32=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TA355_LM_DX_F32=$v', d['ms_per_step'], d['value'], d.get('parity'))"
  done
done 2>&1 | tee gpurun_out/r04_i/ab_dx_bf16.txt
