#!/bin/bash
# round 4, visit x: decode prefetch workgroups automatic vs fixed; tests; kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_x
export TMPDIR=/tmp
for rep in 1 2; do
for w in 0 128 96; do
  TA355_DECODE_PF_WGS=$w python scripts/gen_bench.py 32 64 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pf_wgs=$w', d['per_token_ms'], d['roofline']['frac'])"
done
done | tee gpurun_out/r04_x/gen_bench_pf_auto.txt
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -k "decode_step_fused" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -k "generate or greedy or decode or stream" 2>&1 | tail -2
python scripts/gen_bench.py 32 64 2>/dev/null | tail -1 > gpurun_out/r04_x/gen_bench_b32.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o b -- python $REPO/scripts/gen_bench.py 32 33 > /tmp/prof_dec.log 2>&1)
S=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && python scripts/summarize_rocprof.py "$S" gpurun_out/r04_x/decode_kernel_stats.md --steps 1 --note "scripts/gen_bench.py 32 33 (4 x generate(1 token) + 4 x generate(33 tokens), B = 32: 128 decode steps of 28 layers = 3584 launches of each layer kernel), fused decode step with next-kernel prefetch, under rocprofv3 --kernel-trace --stats" | sed -n 10,20p
