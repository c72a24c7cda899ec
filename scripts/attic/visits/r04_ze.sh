#!/bin/bash
# round 4, visit ze: MoE step time re-check (the evidence visit's 48.9 ms against 42.2 earlier in the round): three runs + MLP beside it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_ze
for i in 1 2 3; do
  for cfg in "--projector moe" ""; do
    python bench.py $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg=[$cfg]', d['ms_per_step'], d['value'])"
  done
done | tee gpurun_out/r04_ze/moe_recheck.txt
python bench.py --projector moe --no-cpu-baseline --no-logits-full 2>/dev/null | tail -1 > gpurun_out/r04_ze/bench_moe_default_flags.json
python -c "import json; d=json.load(open('gpurun_out/r04_ze/bench_moe_default_flags.json')); print('default flags', d['ms_per_step'], d['steps'], d['warmup'])"
