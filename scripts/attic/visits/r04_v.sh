#!/bin/bash
# round 4, visit v: decode prefetch workgroup count in situ
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_v
for rep in 1 2; do
for cfg in "0 128" "1 32" "1 64" "1 128" "1 256" "1 512"; do
  set -- $cfg
  TA355_DECODE_PREFETCH=$1 TA355_DECODE_PF_WGS=$2 python scripts/gen_bench.py 32 64 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prefetch=$1 wgs=$2', d['per_token_ms'], d['roofline']['frac'])"
done
done | tee gpurun_out/r04_v/gen_bench_pf_wgs.txt
