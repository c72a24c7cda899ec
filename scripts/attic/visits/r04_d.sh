#!/bin/bash
# round 4, visit d: gemm_v7 with the bias row through wave-private LDS and LUT-only GELU -- tests + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_d
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -8 > gpurun_out/r04_d/pytest_round4.log
tail -4 gpurun_out/r04_d/pytest_round4.log
timeout 600 python scripts/gemm_v7_ab.py --reps 8 --variants ,14,15 > gpurun_out/r04_d/gemm_v7_ab.txt 2>&1
tail -17 gpurun_out/r04_d/gemm_v7_ab.txt | head -16
