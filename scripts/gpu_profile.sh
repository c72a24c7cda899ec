#!/bin/bash
# rocprofv3 kernel-trace + stats of a short bench run; summary CSVs are merged back under gpurun_out/prof_<tag>/.
TAG=${1:-r1}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- \
  python $REPO/bench.py ${PROF_BENCH_ARGS:---steps 3 --warmup 1 --batch 32 --no-cpu-baseline --no-roofline} > $OUT/run.log 2>&1
echo "rocprof rc=$?"
find $OUT -name "*.csv" | head -20
# keep the merge small: drop the per-dispatch trace if it is huge, keep stats
find $OUT -name "*kernel_trace.csv" -size +40M -delete
STATS=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$STATS" ] && head -40 "$STATS"
tail -3 $OUT/run.log
