#!/bin/bash
# round 3, visit l: ping-pong GEMM epilogue without global loads (bias / rope rows / GELU table DMA'd into the idle stage during the
# last K tile; bf16 residual as the accumulators' start value) -- GEMM + model tests, same-box A/B against the previous library build
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q -x -k "gemm or encoder or asr_model or b32 or full_depth or smoke or moe or lora" 2>&1 | tail -30 > $OUT/r3l_pytest.log
echo "pytest rc=${PIPESTATUS[0]}"; tail -8 $OUT/r3l_pytest.log
echo "== A/B: prev library | new, residual in the epilogue | new"
for i in 1 2 3; do
  for cfg in "libta355_prev.so 1" "libta355.so 0" "libta355.so 1"; do
    set -- $cfg
    TA355_GEMM_RES_INIT=$2 TA355_LIB=$REPO/tiny_audio_amd/$1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 res_init=$2', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3l_ab_epilogue_lds.txt
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof_r3l; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-logits-full > $P/run.log 2>&1; echo "rocprof rc=$?"
TR=$(find $P -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/summarize_trace_steps.py $TR $OUT/r3l_kernel_steps.md --skip 1 --note "bench.py --steps 4 --warmup 1 (configs[1], B = 32), rocprofv3 --kernel-trace --stats; round 3 visit l (epilogue constants through LDS, residual as accumulator start)" | head -24
find $P -name "*kernel_trace.csv" -delete
