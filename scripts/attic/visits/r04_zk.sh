#!/bin/bash
# round 4, visit zk: why the MoE step is 42 ms on some boxes / runs and 45-49 on others: per-step GPU times, 4 runs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_zk
for i in 1 2 3 4; do
  python bench.py --projector moe --steps 12 --warmup 2 --no-cpu-baseline --no-logits-full --no-roofline --step-times 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('moe', d['ms_per_step'], d['step_ms'])"
done | tee gpurun_out/r04_zk/moe_step_times.txt
python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-logits-full --no-roofline --step-times 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mlp', d['ms_per_step'], d['step_ms'])" | tee -a gpurun_out/r04_zk/moe_step_times.txt
