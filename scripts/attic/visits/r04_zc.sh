#!/bin/bash
# round 4, visit zc: log-mel kernel floor: dbg 3 (no transform, no mel taps), 11 (+ no write-out), 27 (+ no mel stage at all), 31
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_zc
export TMPDIR=/tmp
for d in 3 11 19 27 8; do
  rm -rf /tmp/prof_lm
  (cd /tmp && TA355_LOGMEL_DEBUG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lm -o b -- python $REPO/scripts/logmel_bench.py > /dev/null 2>&1)
  S=$(find /tmp/prof_lm -name "*kernel_stats.csv" | head -1)
  echo -n "dbg=$d  "; python - "$S" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "logmel_fft" in r["Name"]]
print("  ".join(f"{float(r['TotalDurationNs']) / int(r['Calls']) / 1e3:.1f} us x{r['Calls']}" for r in rows))
PY
done | tee gpurun_out/r04_zc/logmel_floor.txt
