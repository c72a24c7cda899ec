#!/bin/bash
# round 4, visit zb: persistent log-mel kernel: parity tests + kernel times (rocprofv3) persistent vs one tile per workgroup
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_zb
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round2.py -x -q -k "logmel or collator_end_to_end" 2>&1 | tail -3
TA355_LOGMEL_MFMA=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "logmel" 2>&1 | tail -1
TA355_LOGMEL_PERSIST=0 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "logmel" 2>&1 | tail -1
for cfg in "1 0 0" "0 0 0" "1 0 3" "1 1 0"; do
  set -- $cfg
  rm -rf /tmp/prof_lm
  (cd /tmp && TA355_LOGMEL_PERSIST=$1 TA355_LOGMEL_MFMA=$2 TA355_LOGMEL_DEBUG=$3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lm -o b -- python $REPO/scripts/logmel_bench.py > /dev/null 2>&1)
  S=$(find /tmp/prof_lm -name "*kernel_stats.csv" | head -1)
  echo -n "persist=$1 mfma=$2 dbg=$3  "; python - "$S" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "logmel" in r["Name"]]
print("  ".join(f"{r['Name'].split('(')[0][-34:]}: {float(r['TotalDurationNs']) / int(r['Calls']) / 1e3:.1f} us x{r['Calls']}" for r in rows))
PY
done | tee gpurun_out/r04_zb/logmel_kernel_times.txt
