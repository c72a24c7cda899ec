#!/bin/bash
# LoRA bring-up visit: K-extension GEMM tests, LoRA parity tests, then the stage-2 bench line.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "k_extension" > gpurun_out/lora_kernels.log 2>&1; echo "kext rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "lora or text_only" > gpurun_out/lora_parity.log 2>&1; echo "lora parity rc=$?"
timeout 600 python bench.py --steps 6 --warmup 2 --lora > gpurun_out/bench_lora.log 2>&1; echo "bench lora rc=$?"
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_mlp.log 2>&1; echo "bench mlp rc=$?"
tail -30 gpurun_out/lora_kernels.log; tail -40 gpurun_out/lora_parity.log; tail -3 gpurun_out/bench_lora.log; tail -2 gpurun_out/bench_mlp.log
