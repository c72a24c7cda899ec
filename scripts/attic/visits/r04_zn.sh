#!/bin/bash
# round 4, visit zn: MoE on one rank back-propagates aux * N directly (no shadow recomputation): parity tests + step time
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_zn
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -k "moe" 2>&1 | tail -3
for i in 1 2 3; do
  python bench.py --projector moe --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('moe', d['ms_per_step'], d['value'])"
done | tee gpurun_out/r04_zn/moe.txt
python bench.py --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mlp', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r04_zn/moe.txt
