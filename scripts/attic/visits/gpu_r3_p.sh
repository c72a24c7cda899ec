#!/bin/bash
# round 3, visit p: LoRA -- K extension on the persistent ping-pong kernel, rank-space projections over the non-zero column blocks only
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -x -k "lora or gemm or generate or decode" 2>&1 | tail -6 | tee $OUT/r3p_pytest.log
echo "== A/B on bench.py --lora"
for i in 1 2; do
  for setting in "TA355_GEMM_PERSIST_KEXT=0 TA355_LORA_NT_FULL=1" "TA355_GEMM_PERSIST_KEXT=1 TA355_LORA_NT_FULL=1" "TA355_GEMM_PERSIST_KEXT=0 TA355_LORA_NT_FULL=0" "TA355_GEMM_PERSIST_KEXT=1 TA355_LORA_NT_FULL=0"; do
    env $setting python bench.py --lora --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$setting', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3p_ab_lora.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mlp', d['ms_per_step'], d['value'])" | tee -a $OUT/r3p_ab_lora.txt
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof_r3p; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python $REPO/bench.py --lora --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-logits-full > $P/run.log 2>&1; echo "rocprof rc=$?"
TR=$(find $P -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/summarize_trace_steps.py $TR $OUT/r3p_lora_kernel_steps.md --skip 1 --note "bench.py --lora --steps 4 --warmup 1 (B = 32), rocprofv3 --kernel-trace --stats; round 3 visit p" | grep -i "lora\|steps:\|v2<\|v4<256\|v4<320, 0, true, false"
find $P -name "*kernel_trace.csv" -delete
