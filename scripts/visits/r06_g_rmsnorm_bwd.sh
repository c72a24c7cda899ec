#!/bin/bash
# round 6, visit g: half-wave RMSNorm backward (both stream modes): tests, same-box A/B against the round-6 evidence library, kernel times
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
O=gpurun_out/r06_g; mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -q -m gpu -x -k "rmsnorm or round6 or lm_ or full_depth or b32_step or recipe or three_training or fullft or full_ft or lora or two_models" > $O/pytest_sub.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sub.log
B="python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-logits-full"
for mode in f32 bf16; do
  out=$O/ab_$mode.txt; : > $out
  for i in 1 2 3; do
    for lib in base new; do
      if [ $lib = new ]; then unset TA355_LIB; else export TA355_LIB=$PWD/tiny_audio_amd/libta355_$lib.so; fi
      echo -n "$mode $lib run $i: " >> $out
      timeout 200 $B --streams $mode 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'], r['gemm_ms_per_step'], d['final_loss'])" >> $out 2>&1
    done
  done
  unset TA355_LIB
  cat $out
done
for mode in f32 bf16; do
OUT=/tmp/prof_$mode
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $REPO/bench.py --streams $mode --steps 4 --warmup 2 --no-cpu-baseline --no-logits-full --no-roofline > $OUT.log 2>&1 < /dev/null)
T=$(find $OUT -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python scripts/summarize_trace_steps.py "$T" $O/kernel_steps_$mode.md --skip 2 --note "bench.py --streams $mode under rocprofv3 --kernel-trace, library = new" | tail -1
grep -E "rmsnorm|kernel time" $O/kernel_steps_$mode.md
done
