#!/bin/bash
# round 3, visit n: next-tile look-up ahead of the main loop (A/B vs the library of visit m); fresh MoE / LoRA kernel tables
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -x -k "gemm or b32 or smoke or moe" 2>&1 | tail -5 | tee $OUT/r3n_pytest.log
echo "== A/B: library of visit m | next-tile look-up before the main loop"
for i in 1 2 3; do
  for lib in libta355_prev.so libta355.so; do
    TA355_LIB=$REPO/tiny_audio_amd/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3n_ab_next_tile.txt
cd /tmp && export TMPDIR=/tmp
for f in moe lora; do
  flag="--projector moe"; [ $f = lora ] && flag="--lora"
  P=$OUT/prof_r3n_$f; rm -rf $P; mkdir -p $P
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python $REPO/bench.py $flag --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-logits-full > $P/run.log 2>&1; echo "rocprof $f rc=$?"
  TR=$(find $P -name "*kernel_trace.csv" | head -1)
  python $REPO/scripts/summarize_trace_steps.py $TR $OUT/r3n_${f}_kernel_steps.md --skip 1 --note "bench.py $flag --steps 4 --warmup 1 (B = 32), rocprofv3 --kernel-trace --stats; round 3 visit n" | head -12
  find $P -name "*kernel_trace.csv" -delete
done
