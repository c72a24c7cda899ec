#!/bin/bash
# round-2 evidence visit: bench lines of every configuration + per-step kernel table of the default one
TAG=${1:-r02_a}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
export PYTHONUNBUFFERED=1
cd $REPO
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_mlp.json 2> gpurun_out/bench_${TAG}_mlp.err; echo "mlp rc=$?"
timeout 600 python bench.py --projector moe --no-cpu-baseline --no-logits-full > gpurun_out/bench_${TAG}_moe.json 2>/dev/null; echo "moe rc=$?"
timeout 600 python bench.py --lora --no-cpu-baseline --no-logits-full > gpurun_out/bench_${TAG}_lora.json 2>/dev/null; echo "lora rc=$?"
timeout 600 python bench.py --lm 1.7b --no-cpu-baseline --no-logits-full > gpurun_out/bench_${TAG}_lm17.json 2>/dev/null; echo "lm17 rc=$?"
timeout 600 python bench.py --full-ft --no-cpu-baseline --no-logits-full > gpurun_out/bench_${TAG}_fullft.json 2>/dev/null; echo "fullft rc=$?"
timeout 600 python bench.py --projector qformer --no-cpu-baseline --no-logits-full --no-roofline > gpurun_out/bench_${TAG}_qformer.json 2>/dev/null; echo "qformer rc=$?"
timeout 600 python bench.py --projector mosa --no-cpu-baseline --no-logits-full --no-roofline > gpurun_out/bench_${TAG}_mosa.json 2>/dev/null; echo "mosa rc=$?"
for f in mlp moe lora lm17 fullft qformer mosa; do python -c "import sys,json; d=json.loads(open('gpurun_out/bench_${TAG}_$f.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$f', d['ms_per_step'], d['value'], r.get('achieved'), r.get('frac'))"; done
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-logits-full > $OUT/run.log 2>&1; echo "rocprof rc=$?"
TR=$(find $OUT -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/summarize_trace_steps.py $TR $REPO/gpurun_out/${TAG}_kernel_steps.md --skip 1 --note "bench.py --steps 4 --warmup 1 (configs[1], B = 32), rocprofv3 --kernel-trace --stats; round-2 head" | head -45
ST=$(find $OUT -name "*kernel_stats.csv" | head -1)
python $REPO/scripts/summarize_rocprof.py $ST $REPO/gpurun_out/${TAG}_kernel_stats.md --steps 5 --note "rocprofv3 --kernel-trace --stats of bench.py --steps 4 --warmup 1 (includes model construction)" > /dev/null
find $OUT -name "*kernel_trace.csv" -delete
