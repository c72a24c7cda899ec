#!/bin/bash
# round 6, visit f: no head-major V copy in the LM attention forward (backward reads V in place) + f32-input half-wave RMSNorm forward:
# tests, same-box A/B in both stream modes (base = commit b1c84a6), kernel table
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
O=gpurun_out/r06_f; mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -q -m gpu -x -k "attention or attn or lm_ or full_depth or b32_step or recipe or generate or decode or posids or position or rmsnorm or round6 or three_training or fullft or full_ft or lora" > $O/pytest_sub.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sub.log
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
OUT=/tmp/prof_new
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-logits-full --no-roofline > $OUT.log 2>&1 < /dev/null)
T=$(find $OUT -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python scripts/summarize_trace_steps.py "$T" $O/kernel_steps_new_f32.md --skip 2 --note "bench.py (f32 streams) under rocprofv3 --kernel-trace, library = new" | tail -1
grep -E "attn_|rmsnorm|kernel time" $O/kernel_steps_new_f32.md
timeout 300 python scripts/attn_stamps.py --streams f32 --out $O/attn_stamps_f32.txt 2>&1 | grep -v Warning | sed -n 3,18p
