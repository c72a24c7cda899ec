#!/bin/bash
# round 4, visit p: fused decode step -- fused-vs-unfused test, generate parity tests, gen_bench A/B, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_p
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -k "decode_step_fused" 2>&1 | tail -15 > gpurun_out/r04_p/pytest_fused.log
tail -5 gpurun_out/r04_p/pytest_fused.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -k "generate or greedy or decode or stream" 2>&1 | tail -8 > gpurun_out/r04_p/pytest_generate.log
tail -4 gpurun_out/r04_p/pytest_generate.log
for v in 1 0 1 0; do
  TA355_DECODE_FUSED=$v python scripts/gen_bench.py 32 64 2>/dev/null | tail -1
done | tee gpurun_out/r04_p/gen_bench_ab.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o b -- python $REPO/scripts/gen_bench.py 32 33 > /tmp/prof_dec.log 2>&1)
S=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && python scripts/summarize_rocprof.py "$S" gpurun_out/r04_p/decode_kernel_stats_fused.md --steps 1 --note "scripts/gen_bench.py 32 33 (4 x generate(1 token) + 4 x generate(33 tokens), B = 32), fused decode step, under rocprofv3 --kernel-trace --stats" | sed -n 10,24p
