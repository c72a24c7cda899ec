#!/bin/bash
# what the chip reports while the step runs: power, clocks, caps (evidence for the sustained-clock discussion in DESIGN section 8)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_p_power.txt
: > $out
timeout 20 rocm-smi --showmaxpower --showpower --showclocks --showperflevel < /dev/null >> $out 2>&1
(timeout 250 python bench.py --steps 2000 --warmup 3 --no-cpu-baseline --no-logits-full < /dev/null > gpurun_out/r05_p_bench.json 2>/dev/null) &
BP=$!
sleep 55
for i in 1 2 3 4 5 6; do
  echo "--- sample $i (bench running)" >> $out
  timeout 10 rocm-smi --showpower --showclocks --showtemp < /dev/null 2>&1 | grep -v "^$\|=====\|WARNING" >> $out
  sleep 2
done
wait $BP
tail -c 400 gpurun_out/r05_p_bench.json >> $out
grep -i "power\|sclk\|mclk\|level\|temp" $out | head -60
