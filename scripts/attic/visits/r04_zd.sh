#!/bin/bash
# round 4, visit zd: log-mel with one clip-maximum atomic per workgroup: parity + kernel times
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_zd
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round2.py -x -q -k "logmel or collator_end_to_end" 2>&1 | tail -2
for cfg in "1 0" "0 0" "1 3" "1 19"; do
  set -- $cfg
  rm -rf /tmp/prof_lm
  (cd /tmp && TA355_LOGMEL_PERSIST=$1 TA355_LOGMEL_DEBUG=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lm -o b -- python $REPO/scripts/logmel_bench.py > /dev/null 2>&1)
  S=$(find /tmp/prof_lm -name "*kernel_stats.csv" | head -1)
  echo -n "persist=$1 dbg=$2  "; python - "$S" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "logmel" in r["Name"]]
print("  ".join(f"{r['Name'].split('(')[0][-24:]}: {float(r['TotalDurationNs']) / int(r['Calls']) / 1e3:.1f} us" for r in rows))
PY
done | tee gpurun_out/r04_zd/logmel_kernel_times.txt
