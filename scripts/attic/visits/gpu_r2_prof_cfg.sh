#!/bin/bash
# per-step kernel table of one bench configuration under rocprofv3: gpu_r2_prof_cfg.sh <tag> <bench args...>
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-logits-full "$@" > $OUT/run.log 2>&1; echo "rocprof rc=$?"
TR=$(find $OUT -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/summarize_trace_steps.py $TR $REPO/gpurun_out/${TAG}_kernel_steps.md --skip 1 --note "bench.py --steps 4 --warmup 1 $* (B = 32), rocprofv3 --kernel-trace --stats; round-2 head" | head -60
find $OUT -name "*kernel_trace.csv" -delete
