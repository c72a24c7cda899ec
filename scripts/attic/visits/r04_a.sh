#!/bin/bash
# round 4, visit a: the AGPR one-wave-per-SIMD GEMM (gemm_v7.hip) -- parity, bit-identity, race hunt, A/B against today's tiles
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -25 > gpurun_out/r04_a/pytest_round4.log
cat gpurun_out/r04_a/pytest_round4.log | tail -8
timeout 600 python scripts/gemm_v7_ab.py --reps 8 > gpurun_out/r04_a/gemm_v7_ab.txt 2>&1
tail -22 gpurun_out/r04_a/gemm_v7_ab.txt
