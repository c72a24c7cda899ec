#!/bin/bash
# round 3, visit c: full -m gpu suite, 192x256 rate A/B, tile-order (GROUP_M) sweep on cold encoder shapes, encoder-attention PMC
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $OUT/r3c_pytest.log
echo "pytest rc=${PIPESTATUS[0]}"; tail -12 $OUT/r3c_pytest.log
python scripts/attn_enc_bench.py | tee $OUT/r3c_attn_enc_bench.txt
echo "== A/B 192x256 tile rate (0 = never)"; bash scripts/gpu_ab_env.sh TA355_RATE_192x256 "0 1.15" 2>&1 | tee $OUT/r3c_ab_gemm192.txt
timeout 300 python scripts/gemm_bench.py --cold --only gemm --match enc_ --variants 4g1,4g2,4g4,4g8,4g16 > $OUT/r3c_gemm_enc_groupm_cold.txt 2>&1; grep -v "^{" $OUT/r3c_gemm_enc_groupm_cold.txt | tail -24
PMC_CMD="python $REPO/scripts/attn_enc_bench.py --reps 3" bash scripts/gpu_pmc.sh r3c_attn 2>&1 | tail -8
python scripts/summarize_pmc.py $OUT/pmc_r3c_attn $OUT/r3c_attn_pmc_summary.md --note "scripts/attn_enc_bench.py, B = 32, 20 heads, S = 500"; head -30 $OUT/r3c_attn_pmc_summary.md
