#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python scripts/gemm_desync.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c_desync.txt
for i in 1 2; do
  for r in 0 1; do
    TA355_GEMM_RING=$r timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ring=$r', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'])"
  done
done | tee gpurun_out/r2c_ring_ab.txt
timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -q --tb=short -p no:cacheprovider -k "opcheck or hf_trainer or moe" 2>&1 | tail -40
