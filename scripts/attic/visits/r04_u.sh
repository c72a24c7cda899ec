#!/bin/bash
# round 4, visit u: fused decode (final form): tests, gen_bench, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -k "decode_step_fused" 2>&1 | tail -5 > gpurun_out/r04_u/pytest_fused.log
tail -2 gpurun_out/r04_u/pytest_fused.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -k "generate or greedy or decode or stream" 2>&1 | tail -30 > gpurun_out/r04_u/pytest_generate.log
tail -3 gpurun_out/r04_u/pytest_generate.log
for cfg in "1 1" "1 0" "0 0"; do
  set -- $cfg
  TA355_DECODE_FUSED=$1 TA355_DECODE_PREFETCH=$2 python scripts/gen_bench.py 32 64 2>/dev/null | tail -1
done | tee gpurun_out/r04_u/gen_bench_ab.txt | cut -c1-110
python scripts/gen_bench.py 1 64 2>/dev/null | tail -1 | tee gpurun_out/r04_u/gen_bench_b1.json | cut -c1-110
python scripts/gen_bench.py 8 64 2>/dev/null | tail -1 | tee gpurun_out/r04_u/gen_bench_b8.json | cut -c1-110
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o b -- python $REPO/scripts/gen_bench.py 32 33 > /tmp/prof_dec.log 2>&1)
S=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && python scripts/summarize_rocprof.py "$S" gpurun_out/r04_u/decode_kernel_stats.md --steps 1 --note "scripts/gen_bench.py 32 33 (4 x generate(1 token) + 4 x generate(33 tokens), B = 32: 128 decode steps of 28 layers = 3584 launches of each layer kernel), fused decode step with next-kernel prefetch, under rocprofv3 --kernel-trace --stats" | sed -n 10,20p
