#!/bin/bash
# round 3, visit e: full -m gpu suite (GELU table in every tile variant; DMA double-buffered attention backward); step table
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/r3e_pytest.log
echo "pytest rc=${PIPESTATUS[0]}"; tail -8 $OUT/r3e_pytest.log
for i in 1 2 3; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mlp', d['ms_per_step'], d['value'])"; done | tee $OUT/r3e_bench3.txt
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof_r3e; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-logits-full > $P/run.log 2>&1; echo "rocprof rc=$?"
TR=$(find $P -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/summarize_trace_steps.py $TR $OUT/r3e_kernel_steps.md --skip 1 --note "bench.py --steps 4 --warmup 1 (configs[1], B = 32), rocprofv3 --kernel-trace --stats; round 3 visit e" | head -34
python - "$TR" <<'PY' | tee $OUT/r3e_alternating_split.txt
# launches of one kernel name alternate between two call sites in the encoder layer (o_proj, fc2): even / odd means
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in by.items():
    if "gemm_nt_kernel_v4<320, 0, true, true" in k or "gemm_nt_kernel_v4<320, 1" in k:
        v = v[len(v) // 4:]          # skip the warm-up step
        print(k[:60], "n", len(v), "even %.1f us  odd %.1f us  min %.1f max %.1f" % (sum(v[0::2]) / len(v[0::2]), sum(v[1::2]) / len(v[1::2]), min(v), max(v)))
PY
find $P -name "*kernel_trace.csv" -delete
