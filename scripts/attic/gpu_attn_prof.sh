#!/bin/bash
# kernel-time A/B of attention variants under rocprofv3 (GPU durations, not host launch rate): VAR=ENVNAME VALS="a b"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
VAR=${VAR:-TA355_ATTN_GQA}; VALS=${VALS:-"1 0"}
cd /tmp && export TMPDIR=/tmp
for f in $VALS; do
  env $VAR=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_attn_$f -o a -- python $REPO/scripts/gemm_bench.py --only attn --reps 30 > /dev/null 2>&1
  python - <<PY
import csv
for r in csv.DictReader(open("$REPO/gpurun_out/prof_attn_$f/a_kernel_stats.csv")):
    if "attn_fwd" in r["Name"]: print("$VAR=$f", r["Name"][:44], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
  rm -rf $REPO/gpurun_out/prof_attn_$f
done
