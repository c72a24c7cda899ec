#!/bin/bash
# round 4, visit zm: bench.py with the collector frozen after warm-up: MoE at the default flags, three runs; MLP beside it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_zm
for i in 1 2; do
  python bench.py --projector moe --no-cpu-baseline --no-logits-full --no-roofline --step-times 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('moe default flags', d['ms_per_step'], d['step_ms'])"
done | tee gpurun_out/r04_zm/moe_default_flags.txt
python bench.py --no-cpu-baseline --no-logits-full --no-roofline --step-times 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mlp default flags', d['ms_per_step'], d['step_ms'])" | tee -a gpurun_out/r04_zm/moe_default_flags.txt
python bench.py --projector moe --no-cpu-baseline --no-logits-full 2>/dev/null | tail -1 > gpurun_out/r04_zm/bench_moe.json
