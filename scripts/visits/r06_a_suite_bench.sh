#!/bin/bash
# round 6, visit a: GPU suite (with skip reasons) on the ABI-4 library, bench line in the recipe's mode (f32 streams, default) with the
# bf16 mode beside it, and per-step kernel tables of BOTH modes (where the fp32 streams' +7 % goes)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_a
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -q -m gpu -rs -x > gpurun_out/r06_a/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06_a/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r06_a/bench_mlp.json 2> gpurun_out/r06_a/bench_mlp.err; echo "bench rc=$?"; tail -1 gpurun_out/r06_a/bench_mlp.json | cut -c1-400
for mode in f32 bf16; do
  OUT=/tmp/prof_$mode
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $REPO/bench.py --streams $mode --steps 4 --warmup 2 --no-cpu-baseline --no-logits-full --no-roofline > $OUT.log 2>&1 < /dev/null)
  T=$(find $OUT -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python scripts/summarize_trace_steps.py "$T" gpurun_out/r06_a/kernel_steps_$mode.md --skip 2 --note "bench.py --streams $mode under rocprofv3 --kernel-trace" | tail -2
  S=$(find $OUT -name "*kernel_stats.csv" | head -1)
  [ -n "$S" ] && head -40 "$S" > gpurun_out/r06_a/kernel_stats_$mode.csv
  tail -1 $OUT.log | cut -c1-200
done
ls -la gpurun_out/r06_a
