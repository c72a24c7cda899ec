#!/bin/bash
# round 4, visit zp: the round-3 decode sequence (TA355_DECODE_FUSED=0) and the fp32 dx / fused-Delta switches still pass their tests
cd "$GRAFT_REPO_ROOT" || exit 1
TA355_DECODE_FUSED=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -q -k "generate or greedy or stream" 2>&1 | tail -2
TA355_ATTN_DELTA_FUSED=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "qwen3 or lm_ or asr_model" 2>&1 | tail -2
TA355_LM_DX_F32=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "qwen3 or lm_ or asr_model" 2>&1 | tail -2
TA355_LORA_TN_PARTS=0 timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "lora" 2>&1 | tail -2
