#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "relu_and_mix" > gpurun_out/mosa_kernels.log 2>&1; echo "prims rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "mosa" > gpurun_out/mosa_parity.log 2>&1; echo "mosa parity rc=$?"
tail -15 gpurun_out/mosa_kernels.log; tail -40 gpurun_out/mosa_parity.log
