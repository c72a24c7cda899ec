#!/bin/bash
# round 5 visit f: attention backward epilogue reads x-hat from the tape's q / k instead of qkv0: parity + same-box A/B vs previous library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "attention or attn or qkv or lm or asr or full or train" 2>&1 | tail -5
timeout 300 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "full_depth" 2>&1 | tail -3
B="python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline"
out=gpurun_out/r05_f_ab_attn_bwd_xhat.txt
: > $out
for i in 1 2 3; do
  for lib in prev new; do
    if [ $lib = prev ]; then export TA355_LIB=$PWD/tiny_audio_amd/libta355_prev.so; else unset TA355_LIB; fi
    echo -n "$lib run $i: " >> $out
    timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> $out 2>&1
  done
done
cat $out
