#!/bin/bash
# round 4, visit s: fused decode, blocked activations + next-kernel prefetch: probe, tests, gen_bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_s
timeout 200 ./scripts/probe/dec_probe | tee gpurun_out/r04_s/dec_probe.txt | tail -6
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -k "decode_step_fused" 2>&1 | tail -15 > gpurun_out/r04_s/pytest_fused.log
tail -5 gpurun_out/r04_s/pytest_fused.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -k "generate or greedy or decode or stream" 2>&1 | tail -8 > gpurun_out/r04_s/pytest_generate.log
tail -4 gpurun_out/r04_s/pytest_generate.log
for cfg in "1 1" "1 0" "0 0" "1 1" "1 0"; do
  set -- $cfg
  TA355_DECODE_FUSED=$1 TA355_DECODE_PREFETCH=$2 python scripts/gen_bench.py 32 64 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused=$1 prefetch=$2', d['per_token_ms'], d['roofline']['frac'])"
done | tee gpurun_out/r04_s/gen_bench_ab.txt
