#!/bin/bash
# round-4 evidence visit: full GPU suite, bench lines of every configuration, per-step kernel table, PMC passes in situ
TAG=${1:-r04_final}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd $REPO
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $OUT/${TAG}_pytest.log; echo "pytest rc=${PIPESTATUS[0]}"; tail -3 $OUT/${TAG}_pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_mlp.json 2> $OUT/${TAG}_bench_mlp.err; echo "mlp rc=$?"
timeout 600 python bench.py --projector moe --no-cpu-baseline --no-logits-full > $OUT/${TAG}_bench_moe.json 2>/dev/null; echo "moe rc=$?"
timeout 600 python bench.py --lora --no-cpu-baseline --no-logits-full > $OUT/${TAG}_bench_lora.json 2>/dev/null; echo "lora rc=$?"
timeout 600 python bench.py --lm 1.7b --no-cpu-baseline --no-logits-full > $OUT/${TAG}_bench_lm17.json 2>/dev/null; echo "lm17 rc=$?"
timeout 600 python bench.py --full-ft --no-cpu-baseline --no-logits-full > $OUT/${TAG}_bench_fullft.json 2>/dev/null; echo "fullft rc=$?"
timeout 600 python bench.py --projector qformer --no-cpu-baseline --no-logits-full --no-roofline > $OUT/${TAG}_bench_qformer.json 2>/dev/null; echo "qformer rc=$?"
timeout 600 python bench.py --projector mosa --no-cpu-baseline --no-logits-full --no-roofline > $OUT/${TAG}_bench_mosa.json 2>/dev/null; echo "mosa rc=$?"
for f in mlp moe lora lm17 fullft qformer mosa; do python -c "import sys,json; d=json.loads(open('gpurun_out/${TAG}_bench_$f.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$f', d['ms_per_step'], d['value'], r.get('achieved'), r.get('frac'), d.get('parity'))"; done
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof_$TAG; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-logits-full > $P/run.log 2>&1; echo "rocprof rc=$?"
TR=$(find $P -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/summarize_trace_steps.py $TR $OUT/${TAG}_kernel_steps.md --skip 1 --note "bench.py --steps 4 --warmup 1 (configs[1], B = 32), rocprofv3 --kernel-trace --stats; round-4 head" | head -30
ST=$(find $P -name "*kernel_stats.csv" | head -1)
python $REPO/scripts/summarize_rocprof.py $ST $OUT/${TAG}_kernel_stats.md --steps 5 --note "rocprofv3 --kernel-trace --stats of bench.py --steps 4 --warmup 1 (includes model construction)" > /dev/null
find $P -name "*kernel_trace.csv" -delete
cd $REPO
PMC_CMD="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-logits-full" bash scripts/gpu_pmc.sh ${TAG}_in_situ 2>&1 | tail -8
python scripts/summarize_pmc.py $OUT/pmc_${TAG}_in_situ $OUT/${TAG}_pmc_summary_in_situ.md --json $OUT/${TAG}_pmc_gemm_traffic.json --note "in situ: bench.py --steps 1 --warmup 1 (B=32, MLP projector), every GEMM / attention launch of two training steps; round-4 head" | head -40
rm -rf $OUT/pmc_${TAG}_in_situ
# decode: bench line with its roofline + kernel stats of the fused step
cd $REPO
python scripts/gen_bench.py 32 64 2>/dev/null | tail -1 > $OUT/${TAG}_gen_bench_b32.json; cut -c1-160 $OUT/${TAG}_gen_bench_b32.json
cd /tmp
rm -rf /tmp/prof_dec
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o b -- python $REPO/scripts/gen_bench.py 32 33 > /tmp/prof_dec.log 2>&1
S=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
cd $REPO
[ -n "$S" ] && python scripts/summarize_rocprof.py "$S" $OUT/${TAG}_decode_kernel_stats.md --steps 1 --note "scripts/gen_bench.py 32 33 (4 x generate(1 token) + 4 x generate(33 tokens), B = 32: 128 decode steps of 28 layers = 3584 launches of each layer kernel; the gemm / attn_fwd / layernorm rows are the 8 prompt passes), fused decode step with next-kernel prefetch, rocprofv3 --kernel-trace --stats; round-4 head" > /dev/null
