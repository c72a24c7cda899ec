#!/bin/bash
# round 4, visit g: LM attention backward as one workgroup per (clip, kv head) -- kernel test, model-level tests, in-step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_g
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "one_workgroup_per_kv_head or bwd_with_fused" 2>&1 | tail -12 > gpurun_out/r04_g/pytest_attn.log
tail -6 gpurun_out/r04_g/pytest_attn.log
for i in 1 2 3; do
  for v in 1 0; do
    TA355_ATTN_BWD_GQA=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TA355_ATTN_BWD_GQA=$v', d['ms_per_step'], d['value'], d.get('parity'))"
  done
done 2>&1 | tee gpurun_out/r04_g/ab_attn_bwd_gqa.txt
