#!/bin/bash
# Round-end visit: full GPU regression, the Qwen3-1.7B-width bench line, then the evidence round (profile + bench lines).
mkdir -p gpurun_out
STAGES="kernels parity smoke" bash scripts/gpu_check.sh
timeout 600 python bench.py --lm 1.7b --no-cpu-baseline > gpurun_out/bench_${1}_lm17.json 2> gpurun_out/bench_${1}_lm17.err; echo "bench 1.7b rc=$?"
tail -1 gpurun_out/bench_${1}_lm17.json | cut -c1-300
bash scripts/gpu_round.sh $1
