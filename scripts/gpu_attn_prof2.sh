#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for sc in 12345 54321 0.088; do
  ATTN_SCALE=$sc timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_attnx -o a -- python $REPO/scripts/gemm_bench.py --only attn --reps 30 > /dev/null 2>&1
  echo -n "scale=$sc "; python - <<PY
import csv
for r in csv.DictReader(open("$REPO/gpurun_out/prof_attnx/a_kernel_stats.csv")):
    if "attn_fwd_gqa" in r["Name"]: print(r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
  rm -rf $REPO/gpurun_out/prof_attnx
done
