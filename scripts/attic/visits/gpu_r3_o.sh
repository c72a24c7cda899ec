#!/bin/bash
# round 3, visit o: which hipBLASLt kernels (macro tiles) the vendor library picks on the step's shapes (calibration only)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof_r3o; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python $REPO/scripts/blaslt_kernel_names.py > $P/run.log 2>&1; echo "rc=$?"
cat $P/run.log | tail -15
ST=$(find $P -name "*kernel_stats.csv" | head -1)
python - "$ST" <<'PY' | tee $OUT/r3o_blaslt_kernel_names.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = int(r["Calls"])
    if 10 <= n <= 22:
        print(n, "%.1f us" % (float(r["AverageNs"]) / 1e3), r["Name"][:400])
PY
find $P -name "*kernel_trace.csv" -delete
