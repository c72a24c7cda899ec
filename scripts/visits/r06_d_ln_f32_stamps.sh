#!/bin/bash
# round 6, visit d: new round-6 GPU tests; same-box A/B of the f32-input half-wave LayerNorm (base = visit c's `new`); attention stamps (fixed)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
O=gpurun_out/r06_d; mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_parity.py -q -m gpu -x -rs > $O/pytest_r6.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_r6.log
B="python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-logits-full"
out=$O/ab_f32.txt; : > $out
for i in 1 2 3; do
  for lib in base new; do
    if [ $lib = new ]; then unset TA355_LIB; else export TA355_LIB=$PWD/tiny_audio_amd/libta355_$lib.so; fi
    echo -n "f32 $lib run $i: " >> $out
    timeout 200 $B 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; h=[v for k,v in r['hbm_kernels'].items() if k.startswith('layernorm')][0]; print(d['ms_per_step'], d['value'], r['frac'], r['gemm_ms_per_step'], d['final_loss'], 'LN us', h['avg_us'], 'frac', h['frac'])" >> $out 2>&1
  done
done
unset TA355_LIB
cat $out
timeout 300 python scripts/attn_stamps.py --streams f32 --out $O/attn_stamps_f32.txt 2>&1 | grep -v Warning | tail -70
