#!/bin/bash
# round-2 visit A: every -m gpu test, smoke, the default bench line (with cpu_baseline)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r2a_tests.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/r2a_tests.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/r2a_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r2a_smoke.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err; echo "bench rc=$?"
tail -1 gpurun_out/r2a_bench.json | cut -c1-1500
tail -3 gpurun_out/r2a_bench.err
