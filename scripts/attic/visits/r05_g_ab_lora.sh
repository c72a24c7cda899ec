#!/bin/bash
# round 5 visit g: LoRA step A/B vs the previous library (TA355_LIB), same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
B="python bench.py --lora --steps 12 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline"
out=gpurun_out/${1:-r05_g_ab_lora}.txt
: > $out
for i in 1 2 3; do
  for lib in prev new; do
    if [ $lib = prev ]; then export TA355_LIB=$PWD/tiny_audio_amd/libta355_prev.so; else unset TA355_LIB; fi
    echo -n "$lib run $i: " >> $out
    timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> $out 2>&1
  done
done
cat $out
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -m gpu -k "lora" 2>&1 | tail -3
