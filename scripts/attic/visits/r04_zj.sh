#!/bin/bash
# round 4, visit zj: the Delta fold re-measured against the LEAN tiled backward (the fold had raised its spill 20 -> 140 B/lane and the
# first A/B compared two paths through the same spilling binary): kernel tests, in-step A/B, per-step kernel times of both
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_zj
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention_bwd or attn_bwd" 2>&1 | tail -2
for i in 1 2 3; do
  for v in 0 1; do
    TA355_ATTN_DELTA_FUSED=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TA355_ATTN_DELTA_FUSED=$v', d['ms_per_step'], d['value'])"
  done
done | tee gpurun_out/r04_zj/ab_delta_fused_lean.txt
for v in 0 1; do
  rm -rf /tmp/prof_a
  (cd /tmp && TA355_ATTN_DELTA_FUSED=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o b -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-logits-full --no-roofline > /dev/null 2>&1)
  T=$(find /tmp/prof_a -name "*kernel_trace.csv" | head -1)
  python scripts/summarize_trace_steps.py "$T" gpurun_out/r04_zj/kernel_steps_delta_fused_$v.md --skip 2 --note "TA355_ATTN_DELTA_FUSED=$v" > /dev/null
  echo "TA355_ATTN_DELTA_FUSED=$v"; grep -E "attn_bwd|kernel time" gpurun_out/r04_zj/kernel_steps_delta_fused_$v.md | cut -c1-140
done | tee -a gpurun_out/r04_zj/ab_delta_fused_lean.txt
