#!/bin/bash
# round 5 visit b: the two-workgroups-per-CU GEMM (variants 16 / 17): parity, microbench on the step's shapes, in-step masks
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "v7" 2>&1 | tail -8 > gpurun_out/r05_b_pytest_v7.log
tail -4 gpurun_out/r05_b_pytest_v7.log
timeout 400 python scripts/gemm_v7_ab.py --reps 10 --variants ,16,17 > gpurun_out/r05_b_gemm_2wg_ab.txt 2>&1
cat gpurun_out/r05_b_gemm_2wg_ab.txt | head -20
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-logits-full --no-roofline"
for m in 0 1 4 5 8 16 32 261 13; do
  echo "== TA355_V9_MASK=$m" >> gpurun_out/r05_b_instep.txt
  TA355_V9_MASK=$m timeout 200 $B 2>&1 | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/r05_b_instep.txt 2>&1
done
cat gpurun_out/r05_b_instep.txt
timeout 400 python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "stream_modes or recipe_fixture or aux" 2>&1 | tail -5
