#!/bin/bash
# round 4, visit o: decode baseline (scripts/gen_bench.py, B = 32) + its kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_o
export TMPDIR=/tmp
python scripts/gen_bench.py 32 64 2>/dev/null | tail -1 | tee gpurun_out/r04_o/gen_bench_b32.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o b -- python $REPO/scripts/gen_bench.py 32 33 > /tmp/prof_dec.log 2>&1)
S=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && python scripts/summarize_rocprof.py "$S" gpurun_out/r04_o/decode_kernel_stats_before.md --steps 1 --note "scripts/gen_bench.py 32 33 (4 x generate(1 token) + 4 x generate(33 tokens), B = 32) under rocprofv3 --kernel-trace --stats" | tail -30
