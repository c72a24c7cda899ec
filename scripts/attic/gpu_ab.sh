#!/bin/bash
# Step-time A/B of one environment toggle in a single visit: VAR=NAME VALS="a b" [ARGS="--lora"]
VAR=${VAR:?}; VALS=${VALS:-"0 1"}
for i in 1 2; do
  for f in $VALS; do
    env $VAR=$f timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$f', d['ms_per_step'], d['value'], d['final_loss'])"
  done
done
