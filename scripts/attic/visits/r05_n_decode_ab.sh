#!/bin/bash
# decode: same-box A/B of the library against tiny_audio_amd/libta355_prev.so + the decode tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${1:-r05_n_decode_ab}.txt
: > $out
timeout 300 python -m pytest tests -x -q -m gpu -k "decode or generate or greedy" < /dev/null 2>&1 | tail -2 >> $out
for i in 1 2; do
  for lib in prev new; do
    if [ $lib = prev ]; then export TA355_LIB=$PWD/tiny_audio_amd/libta355_prev.so; else unset TA355_LIB; fi
    echo -n "$lib run $i: " >> $out
    timeout 200 python scripts/gen_bench.py 32 33 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['prompt_pass_ms'], d['per_token_ms'], d['tokens_per_s'])" >> $out 2>&1
  done
done
cat $out
