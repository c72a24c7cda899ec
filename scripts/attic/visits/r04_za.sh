#!/bin/bash
# round 4, visit za: log-mel KERNEL times (rocprofv3) by form and debug bits (the microbench is host-bound near 60 us per call)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_za
export TMPDIR=/tmp
for m in 1 0; do for d in 0 1 2 3; do
  rm -rf /tmp/prof_lm
  (cd /tmp && TA355_LOGMEL_MFMA=$m TA355_LOGMEL_DEBUG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lm -o b -- python $REPO/scripts/logmel_bench.py > /dev/null 2>&1)
  S=$(find /tmp/prof_lm -name "*kernel_stats.csv" | head -1)
  echo -n "MFMA=$m dbg=$d  "; python - "$S" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "logmel" in r["Name"]]
print("  ".join(f"{r['Name'].split('(')[0][-34:]}: {float(r['TotalDurationNs']) / int(r['Calls']) / 1e3:.1f} us x{r['Calls']}" for r in rows))
PY
done; done | tee gpurun_out/r04_za/logmel_kernel_times.txt
