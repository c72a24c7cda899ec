#!/bin/bash
# round 3, visit f: single-pass log-mel (tests + rate), default bench line with roofline / hbm_kernels / parity, MoE and LoRA lines
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -k "logmel or collator or asr_model or smoke or b32 or full_depth" 2>&1 | tail -8 | tee $OUT/r3f_pytest.log
python scripts/logmel_bench.py 2>&1 | tail -4 | tee $OUT/r3f_logmel_bench.txt
TA355_LOGMEL_ONEPASS=0 python scripts/logmel_bench.py 2>&1 | tail -4 | tee -a $OUT/r3f_logmel_bench.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/r3f_bench_mlp.json 2> $OUT/r3f_bench_mlp.err; echo "mlp rc=$?"; cat $OUT/r3f_bench_mlp.json
timeout 300 python bench.py --projector moe --no-cpu-baseline --no-logits-full > $OUT/r3f_bench_moe.json 2>/dev/null; echo "moe rc=$?"
timeout 300 python bench.py --lora --no-cpu-baseline --no-logits-full > $OUT/r3f_bench_lora.json 2>/dev/null; echo "lora rc=$?"
for f in mlp moe lora; do python -c "import sys,json; d=json.loads(open('gpurun_out/r3f_bench_$f.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$f', d['ms_per_step'], d['value'], r.get('achieved'), r.get('frac'))"; done
