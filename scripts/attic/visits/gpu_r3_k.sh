#!/bin/bash
# round 3, visit k: LayerNorm with gamma / beta resident over 4 rows per half wave -- kernel tests, A/B in the step
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -k "layernorm or encoder or asr_model or b32 or full_depth" 2>&1 | tail -6 | tee $OUT/r3k_pytest.log
TA355_LN_ROWS=1 timeout 300 python -m pytest tests -m gpu -q -k "layernorm" 2>&1 | tail -2
echo "== A/B LayerNorm rows per half wave"; bash scripts/gpu_ab_env.sh TA355_LN_ROWS "1 2 4" 2>&1 | tee $OUT/r3k_ab_ln_rows.txt
