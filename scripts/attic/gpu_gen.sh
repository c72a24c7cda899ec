#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "generate" > gpurun_out/gen_parity.log 2>&1; echo "generate parity rc=$?"
tail -40 gpurun_out/gen_parity.log
