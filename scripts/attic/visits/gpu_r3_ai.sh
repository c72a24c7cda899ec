#!/bin/bash
# round 3, visit ai: the documented experiment knobs still give correct results after this round's refactors (subset of the GPU suite per knob)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
for setting in "TA355_GEMM_RES_INIT=0" "TA355_GEMM_DEBUG=2048" "TA355_GELU_LUT=0" "TA355_ENC_ATTN_V2=0" "TA355_GEMM_PERSIST_KEXT=1" "TA355_GEMM_PERSIST=0" "TA355_LORA_NT_FULL=1 TA355_LORA_TN_DUAL=0" "TA355_LN_ROWS=1 TA355_LN_WIDE=0"; do
  echo "== $setting"
  env $setting timeout 600 python -m pytest tests -m gpu -q -k "gemm_step_shapes or gemm_w_blocked or gemm_rope or test_gemm_variants or encoder or lora or full_depth or layernorm or smoke" 2>&1 | tail -1
done | tee $OUT/r3ai_knobs.txt
