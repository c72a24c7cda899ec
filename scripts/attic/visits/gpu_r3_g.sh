#!/bin/bash
# round 3, visit g: full suite with the wide bf16-residual epilogue; its A/B; LayerNorm-fold re-measured on the round-2 encoder path
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $OUT/r3g_pytest.log
echo "pytest rc=${PIPESTATUS[0]}"; tail -6 $OUT/r3g_pytest.log
echo "== A/B wide bf16 residual (1 = store layout, 16 B per lane)"; bash scripts/gpu_ab_env.sh TA355_RES_WIDE "0 1" 2>&1 | tee $OUT/r3g_ab_res_wide.txt
echo "== LayerNorm fold on the round-2 encoder path (ATTN_V2=0): 0 / 1"; TA355_ENC_ATTN_V2=0 bash scripts/gpu_ab_env.sh TA355_ENC_LN_FOLD "0 1" 2>&1 | tee $OUT/r3g_ab_lnfold.txt
