#!/bin/bash
# round 3, visit b: full -m gpu suite (new encoder attention path, 192-row ping-pong GEMM tile), cold GEMM A/B of the LM shapes,
# same-box A/B of the two changes in the step, per-step kernel table
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/r3b_pytest.log
echo "pytest rc=${PIPESTATUS[0]}"; tail -12 $OUT/r3b_pytest.log
timeout 300 python scripts/gemm_bench.py --cold --only gemm --match lm --variants ,3,4,10,12 > $OUT/r3b_gemm_lm_cold.txt 2>&1; cat $OUT/r3b_gemm_lm_cold.txt | tail -45
echo "== A/B encoder attention v2 (1 = new)"; bash scripts/gpu_ab_env.sh TA355_ENC_ATTN_V2 "0 1" 2>&1 | tee $OUT/r3b_ab_enc_attn.txt
echo "== A/B 192x256 tile (rate 0 = never chosen)"; bash scripts/gpu_ab_env.sh TA355_RATE_192x256 "0 1.30" 2>&1 | tee $OUT/r3b_ab_gemm192.txt
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof_r3b; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-logits-full > $P/run.log 2>&1; echo "rocprof rc=$?"
TR=$(find $P -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/summarize_trace_steps.py $TR $OUT/r3b_kernel_steps.md --skip 1 --note "bench.py --steps 4 --warmup 1 (configs[1], B = 32), rocprofv3 --kernel-trace --stats; round 3 visit b" | head -40
find $P -name "*kernel_trace.csv" -delete
