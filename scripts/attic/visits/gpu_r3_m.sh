#!/bin/bash
# round 3, visit m: residual as the accumulators' start value in EVERY tile variant -- full suite, same-box A/B against the library of visit k
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/r3m_pytest.log
echo "pytest rc=${PIPESTATUS[0]}"; tail -8 $OUT/r3m_pytest.log
echo "== A/B: library of visit k | new, residual in the epilogue | new"
for i in 1 2; do
  for cfg in "libta355_prev.so 1" "libta355.so 0" "libta355.so 1"; do
    set -- $cfg
    TA355_GEMM_RES_INIT=$2 TA355_LIB=$REPO/tiny_audio_amd/$1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 res_init=$2', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3m_ab_epilogue_lds.txt
for f in moe lora; do
  flag="--projector moe"; [ $f = lora ] && flag="--lora"
  timeout 300 python bench.py $flag --no-cpu-baseline --no-logits-full > $OUT/r3m_bench_$f.json 2>/dev/null; echo "$f rc=$?"
  python -c "import sys,json; d=json.loads(open('gpurun_out/r3m_bench_$f.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$f', d['ms_per_step'], d['value'], r.get('achieved'), r.get('frac'))"
done
