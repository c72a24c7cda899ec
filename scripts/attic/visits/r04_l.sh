#!/bin/bash
# round 4, visit l: LoRA rank / target-subset goldens on the GPU + kernel traces of the LoRA and MoE steps (where the +5 / +2.4 ms go)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_l
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "lora" 2>&1 | tail -8 > gpurun_out/r04_l/pytest.log
tail -4 gpurun_out/r04_l/pytest.log
export TMPDIR=/tmp
for cfg in "--lora" "--projector moe" ""; do
  tag=$(echo "mlp$cfg" | tr -d ' -')
  python bench.py $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | tail -1 > gpurun_out/r04_l/bench_$tag.json
  python -c "import json; d=json.load(open('gpurun_out/r04_l/bench_$tag.json')); print('$tag', d['ms_per_step'])"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o t -- python "$GRAFT_REPO_ROOT/bench.py" $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-logits-full --no-roofline > /dev/null 2>&1)
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -60 "$f" > gpurun_out/r04_l/kernel_stats_$tag.csv
done
