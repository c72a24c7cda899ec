#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gelu or layernorm_res or attn_small" > gpurun_out/qf_kernels.log 2>&1; echo "prims rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "qformer" > gpurun_out/qf_parity.log 2>&1; echo "qformer parity rc=$?"
tail -30 gpurun_out/qf_kernels.log; tail -40 gpurun_out/qf_parity.log
