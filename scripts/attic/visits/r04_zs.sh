#!/bin/bash
# round 4, visit zs (as visit m, on the final code): per-step kernel tables of the MLP / LoRA / MoE steps (where LoRA's +5 ms and MoE's +2 ms go)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_zs
export TMPDIR=/tmp
for cfg in "--lora" "--projector moe" ""; do
  tag=$(echo "mlp$cfg" | tr -d ' -')
  OUT=/tmp/prof_$tag
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $REPO/bench.py $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-logits-full --no-roofline > $OUT.log 2>&1)
  T=$(find $OUT -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python scripts/summarize_trace_steps.py "$T" gpurun_out/r04_zs/kernel_steps_$tag.md --skip 2 --note "bench.py $cfg under rocprofv3 --kernel-trace" | tail -2
  tail -1 $OUT.log | cut -c1-200
done
ls -la gpurun_out/r04_zs
