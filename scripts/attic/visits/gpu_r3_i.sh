#!/bin/bash
# round 3, visit i: batched epilogue loads -- full suite, same-box A/B against the previous build of the library (TA355_LIB), step table
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $OUT/r3i_pytest.log
echo "pytest rc=${PIPESTATUS[0]}"; tail -6 $OUT/r3i_pytest.log
echo "== A/B library builds: prev = per-fragment epilogue loads, new = batched"
for i in 1 2 3; do
  for lib in libta355_prev.so libta355.so; do
    TA355_LIB=$REPO/tiny_audio_amd/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3i_ab_epilogue.txt
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof_r3i; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-logits-full > $P/run.log 2>&1; echo "rocprof rc=$?"
TR=$(find $P -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/summarize_trace_steps.py $TR $OUT/r3i_kernel_steps.md --skip 1 --note "bench.py --steps 4 --warmup 1 (configs[1], B = 32), rocprofv3 --kernel-trace --stats; round 3 visit i (batched epilogue loads)" | head -24
python - "$TR" <<'PY' | tee $OUT/r3i_alternating_split.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in by.items():
    if "gemm_nt_kernel_v4<320, 0, true, true" in k or "gemm_nt_kernel_v5<0, true, true" in k:
        v = v[len(v) // 4:]
        print(k[:60], "n", len(v), "even %.1f us  odd %.1f us  min %.1f max %.1f" % (sum(v[0::2]) / len(v[0::2]), sum(v[1::2]) / len(v[1::2]), min(v), max(v)))
PY
find $P -name "*kernel_trace.csv" -delete
