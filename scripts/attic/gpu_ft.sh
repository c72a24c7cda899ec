#!/bin/bash
# Full decoder fine-tuning: kernel tests, parity tests, then a bench line of the embedded recipe.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "rmsnorm_dw or embed_grad or lm_qkv_post" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "full_finetune" 2>&1 | tail -25
