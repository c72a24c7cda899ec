#!/bin/bash
# round 3, visit h: full suite; LoRA dual-TN A/B and step table
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $OUT/r3h_pytest.log
echo "pytest rc=${PIPESTATUS[0]}"; tail -6 $OUT/r3h_pytest.log
echo "== A/B LoRA adapter gradients in one launch (1 = dual)"; bash scripts/gpu_ab_env.sh TA355_LORA_TN_DUAL "0 1" --lora 2>&1 | tee $OUT/r3h_ab_lora_dual.txt
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof_r3h; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python $REPO/bench.py --lora --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-logits-full > $P/run.log 2>&1; echo "rocprof rc=$?"
TR=$(find $P -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/summarize_trace_steps.py $TR $OUT/r3h_lora_kernel_steps.md --skip 1 --note "bench.py --lora --steps 4 --warmup 1 (configs[4], B = 32), rocprofv3 --kernel-trace --stats; round 3 visit h" | head -30
find $P -name "*kernel_trace.csv" -delete
