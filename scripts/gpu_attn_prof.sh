#!/bin/bash
# kernel-time A/B of the LM attention forward variants under rocprofv3 (GPU durations, not host launch rate)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  TA355_ATTN_GQA=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_attn$f -o a -- python $REPO/scripts/gemm_bench.py --only attn --reps 30 > /dev/null 2>&1
  echo "gqa=$f"; grep -E "attn_fwd" $REPO/gpurun_out/prof_attn$f/a_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,200-400 | sed 's/"//g' | awk -F, '{print $1, $(NF-2), $(NF-1), $NF}' | cut -c1-160
  rm -f $REPO/gpurun_out/prof_attn$f/a_kernel_trace.csv
done
