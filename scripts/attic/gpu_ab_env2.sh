#!/bin/bash
# same-box comparison of several environment settings over the default bench: gpu_ab_env2.sh "A=1 B=2" "A=3" ... (each argument one setting)
for i in 1 2; do
  for setting in "$@"; do
    env $setting python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$setting', d['ms_per_step'], d['value'])"
  done
done
