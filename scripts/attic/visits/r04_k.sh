#!/bin/bash
# round 4, visit k: Delta folded into the tiled attention backward (no ta_attn_bwd_prep) -- kernel test + in-step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_k
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "bwd_with_fused or one_workgroup" 2>&1 | tail -5 > gpurun_out/r04_k/pytest.log
tail -3 gpurun_out/r04_k/pytest.log
for i in 1 2 3; do
  for v in 1 0; do
    TA355_ATTN_DELTA_FUSED=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TA355_ATTN_DELTA_FUSED=$v', d['ms_per_step'], d['value'])"
  done
done 2>&1 | tee gpurun_out/r04_k/ab_delta_fused.txt
