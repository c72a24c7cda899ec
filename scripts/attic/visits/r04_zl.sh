#!/bin/bash
# round 4, visit zl: the MoE step's sporadic slow step: device allocations inside the run? Python GC?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_zl
for gcoff in 0 0 0 1 1 1; do
  TA355_BENCH_NO_GC=$gcoff python bench.py --projector moe --steps 12 --warmup 2 --no-cpu-baseline --no-logits-full --no-roofline --step-times 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('moe no_gc=$gcoff', d['ms_per_step'], d['step_ms'])"
done | tee gpurun_out/r04_zl/moe_step_times_gc.txt
