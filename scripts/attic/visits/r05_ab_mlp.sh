#!/bin/bash
# same-box A/B of the default step against the previous library (tiny_audio_amd/libta355_prev.so)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-logits-full"
out=gpurun_out/${1:-r05_ab}.txt
: > $out
for i in 1 2 3; do
  for lib in prev new; do
    if [ $lib = prev ]; then export TA355_LIB=$PWD/tiny_audio_amd/libta355_prev.so; else unset TA355_LIB; fi
    echo -n "$lib run $i: " >> $out
    timeout 200 $B 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'], r['gemm_ms_per_step'], d['final_loss'])" >> $out 2>&1
  done
done
cat $out
