#!/bin/bash
# Evidence visit: rocprof of the default bench + full bench lines (MLP with cpu_baseline, MoE, LoRA).
TAG=${1:-r01_c}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
bash scripts/gpu_profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1; echo "profile rc=$?"
timeout 900 python bench.py > gpurun_out/bench_${TAG}_mlp.json 2> gpurun_out/bench_${TAG}_mlp.err; echo "bench mlp rc=$?"
timeout 600 python bench.py --projector moe --no-cpu-baseline > gpurun_out/bench_${TAG}_moe.json 2>/dev/null; echo "bench moe rc=$?"
timeout 600 python bench.py --lora --no-cpu-baseline > gpurun_out/bench_${TAG}_lora.json 2>/dev/null; echo "bench lora rc=$?"
for f in mlp moe lora; do tail -1 gpurun_out/bench_${TAG}_$f.json | cut -c1-260; done
