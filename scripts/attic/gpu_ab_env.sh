#!/bin/bash
# same-box A/B of one environment knob over the default bench: gpu_ab_env.sh VAR "v1 v2" [bench args]
VAR=$1; VALS=$2; shift 2
for i in 1 2 3; do
  for v in $VALS; do
    env $VAR=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['ms_per_step'], d['value'])"
  done
done
