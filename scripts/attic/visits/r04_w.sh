#!/bin/bash
# round 4, visit w: decode q|k|v columns per workgroup x prefetch, in situ
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_w
for rep in 1 2; do
for cfg in "16 0" "16 1" "32 0" "32 1"; do
  set -- $cfg
  TA355_DEC_QKV_COLS=$1 TA355_DECODE_PREFETCH=$2 python scripts/gen_bench.py 32 64 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qkv_cols=$1 prefetch=$2', d['per_token_ms'], d['roofline']['frac'])"
done
done | tee gpurun_out/r04_w/gen_bench_cols_pf.txt
