#!/bin/bash
# round-2 visit B: GEMM variant A/B (ring kernel v3 vs ping-pong v2, epilogue / main-loop split) + the round-2 tests
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python scripts/gemm_ab.py --variants 4,7 > gpurun_out/r2b_gemm_ab_47.txt 2>&1; echo "ab47 rc=$?"
cat gpurun_out/r2b_gemm_ab_47.txt
timeout 600 python scripts/gemm_ab.py --variants 3,6 --match lm_ > gpurun_out/r2b_gemm_ab_36.txt 2>&1; echo "ab36 rc=$?"
cat gpurun_out/r2b_gemm_ab_36.txt
timeout 600 python scripts/gemm_ab.py --variants 4,7 --cold --modes 0 > gpurun_out/r2b_gemm_ab_47_cold.txt 2>&1; echo "ab47cold rc=$?"
cat gpurun_out/r2b_gemm_ab_47_cold.txt
timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30
