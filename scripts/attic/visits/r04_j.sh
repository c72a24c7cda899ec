#!/bin/bash
# round 4, visit j: gemm_v8 (32x32x16 MFMAs, LDS-staged epilogue) -- tests, A/B against the ping-pong tiles and gemm_v7
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_j
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -k "v7 and (16 or 17 or 18)" 2>&1 | tail -12 > gpurun_out/r04_j/pytest_v8.log
tail -8 gpurun_out/r04_j/pytest_v8.log
timeout 600 python scripts/gemm_v7_ab.py --reps 8 --variants ,14,17,15,18,16 > gpurun_out/r04_j/gemm_v8_ab.txt 2>&1
tail -17 gpurun_out/r04_j/gemm_v8_ab.txt | head -16
