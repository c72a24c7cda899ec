#!/bin/bash
# round 5 visit a: the numerics contract -- both stream modes against the reference-recipe fixture, boundary tests, bench line with streams_other
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r05_a_pytest_round5.log
cat gpurun_out/r05_a_pytest_round5.log | tail -15
timeout 600 python bench.py > gpurun_out/r05_a_bench_mlp.json 2> gpurun_out/r05_a_bench_mlp.err
tail -c 3000 gpurun_out/r05_a_bench_mlp.json
