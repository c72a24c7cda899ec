#!/bin/bash
# round-6 evidence visit: full GPU suite (with skip reasons), bench lines of every configuration, per-step kernel tables (both stream
# modes), PMC passes in situ (f32 streams = the default), decode bench + kernel stats
TAG=${1:-r06_final}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd $REPO
timeout 1800 python -m pytest tests -m gpu -q -rs > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_mlp.json 2> $OUT/bench_mlp.err; echo "mlp rc=$?"
timeout 600 python bench.py --projector moe --no-cpu-baseline > $OUT/bench_moe.json 2>/dev/null; echo "moe rc=$?"
timeout 600 python bench.py --lora --no-cpu-baseline > $OUT/bench_lora.json 2>/dev/null; echo "lora rc=$?"
timeout 600 python bench.py --lm 1.7b --no-cpu-baseline --no-logits-full > $OUT/bench_lm17.json 2>/dev/null; echo "lm17 rc=$?"
timeout 600 python bench.py --full-ft --no-cpu-baseline --no-logits-full > $OUT/bench_fullft.json 2>/dev/null; echo "fullft rc=$?"
timeout 600 python bench.py --projector qformer --no-cpu-baseline --no-logits-full --no-roofline > $OUT/bench_qformer.json 2>/dev/null; echo "qformer rc=$?"
timeout 600 python bench.py --projector mosa --no-cpu-baseline --no-logits-full --no-roofline > $OUT/bench_mosa.json 2>/dev/null; echo "mosa rc=$?"
for f in mlp moe lora lm17 fullft qformer mosa; do python -c "import sys,json; d=json.loads(open('$OUT/bench_$f.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; so=d.get('streams_other') or {}; print('$f', d['ms_per_step'], d['value'], r.get('achieved'), r.get('frac'), 'bf16 streams:', so.get('ms_per_step'), d.get('parity'))"; done
cd /tmp && export TMPDIR=/tmp
for mode in f32 bf16; do
  P=/tmp/prof_$mode; rm -rf $P
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python $REPO/bench.py --streams $mode --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-logits-full > $P.log 2>&1; echo "rocprof $mode rc=$?"
  TR=$(find $P -name "*kernel_trace.csv" | head -1)
  [ -n "$TR" ] && python $REPO/scripts/summarize_trace_steps.py $TR $OUT/kernel_steps_$mode.md --skip 2 --note "bench.py --streams $mode --steps 4 --warmup 2 (configs[1], B = 32), rocprofv3 --kernel-trace --stats; round-6 head" | head -12
  ST=$(find $P -name "*kernel_stats.csv" | head -1)
  [ -n "$ST" ] && [ $mode = f32 ] && python $REPO/scripts/summarize_rocprof.py $ST $OUT/kernel_stats.md --steps 6 --note "rocprofv3 --kernel-trace --stats of bench.py --steps 4 --warmup 2, f32 streams (includes model construction)" > /dev/null
  rm -rf $P
done
cd $REPO
PMC_CMD="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-logits-full" bash scripts/gpu_pmc.sh ${TAG}_in_situ 2>&1 | tail -6
python scripts/summarize_pmc.py $REPO/gpurun_out/pmc_${TAG}_in_situ $OUT/pmc_summary_in_situ.md --json $OUT/pmc_gemm_traffic_b32_mlp_f32.json --note "in situ: bench.py --steps 1 --warmup 1 (B=32, MLP projector, f32 streams = the default), every GEMM / attention launch of two training steps; round-6 head" | head -12
rm -rf $REPO/gpurun_out/pmc_${TAG}_in_situ
python scripts/gen_bench.py 32 64 2>/dev/null | tail -1 > $OUT/gen_bench_b32.json; cut -c1-200 $OUT/gen_bench_b32.json
cd /tmp
rm -rf /tmp/prof_dec
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o b -- python $REPO/scripts/gen_bench.py 32 33 > /tmp/prof_dec.log 2>&1
S=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
cd $REPO
[ -n "$S" ] && python scripts/summarize_rocprof.py "$S" $OUT/decode_kernel_stats.md --steps 1 --note "scripts/gen_bench.py 32 33 (4 x generate(1 token) + 4 x generate(33 tokens), B = 32: 128 decode steps of 28 layers = 3584 launches of each layer kernel; the gemm / attn_fwd / layernorm rows are the 8 prompt passes), fused decode step, rocprofv3 --kernel-trace --stats; round-6 head" > /dev/null
timeout 300 python scripts/attn_stamps.py --streams f32 --out $OUT/attn_stamps_f32.txt > /dev/null 2>&1; echo "stamps rc=$?"
ls $OUT
