#!/bin/bash
# per-step kernel tables (rocprofv3 --kernel-trace) of the MLP and MoE steps
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in mlp moe; do
  OUT=$REPO/gpurun_out/prof_r2e_$cfg
  rm -rf $OUT; mkdir -p $OUT
  extra=""; [ $cfg = moe ] && extra="--projector moe"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-logits-full $extra > $OUT/run.log 2>&1
  echo "rocprof $cfg rc=$?"
  TR=$(find $OUT -name "*kernel_trace.csv" | head -1)
  python $REPO/scripts/summarize_trace_steps.py $TR $REPO/gpurun_out/r2e_steps_$cfg.md --skip 1 --note "bench.py --steps 4 --warmup 1 ($cfg), rocprofv3 --kernel-trace" | head -70
  ST=$(find $OUT -name "*kernel_stats.csv" | head -1); cp $ST $REPO/gpurun_out/r2e_kernel_stats_$cfg.csv
  find $OUT -name "*kernel_trace.csv" -delete
done
