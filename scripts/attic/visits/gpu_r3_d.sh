#!/bin/bash
# round 3, visit d: full -m gpu suite; A/B of the GELU chord table and of the q|k|v tile order in the step; attention microbench; step table
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/r3d_pytest.log
echo "pytest rc=${PIPESTATUS[0]}"; tail -8 $OUT/r3d_pytest.log
python scripts/attn_enc_bench.py --only new | tee $OUT/r3d_attn_enc_bench.txt
echo "== A/B GELU chord table (1 = table)"; bash scripts/gpu_ab_env.sh TA355_GELU_LUT "0 1" 2>&1 | tee $OUT/r3d_ab_gelu_lut.txt
echo "== A/B q|k|v tile-order group (1 = 16 row tiles)"; bash scripts/gpu_ab_env.sh TA355_GROUP_M_AUTO "0 1" 2>&1 | tee $OUT/r3d_ab_groupm.txt
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof_r3d; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-logits-full > $P/run.log 2>&1; echo "rocprof rc=$?"
TR=$(find $P -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/summarize_trace_steps.py $TR $OUT/r3d_kernel_steps.md --skip 1 --note "bench.py --steps 4 --warmup 1 (configs[1], B = 32), rocprofv3 --kernel-trace --stats; round 3 visit d" | head -32
find $P -name "*kernel_trace.csv" -delete
