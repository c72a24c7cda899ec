#!/bin/bash
# Fused encoder q|k|v path: tests, then an A/B of the step time in one visit (same box).
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "rope_epilogue or strided or attention_fwd or gemm" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "encoder or golden or full_model" 2>&1 | tail -5
for i in 1 2; do
  for f in 1 0; do
    TA355_ENC_QKV_FUSED=$f timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused=$f', d['ms_per_step'], d['value'], d['final_loss'])"
  done
done
