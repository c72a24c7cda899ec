#!/bin/bash
# round 3, visit ad: row-merged stores, single code path -- libraries: base (pairs), m256 (merged on the 256-column tiles), full (also on 256x320)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
for i in 1 2; do
  for lib in libta355_base.so libta355_m256.so libta355.so; do
    TA355_LIB=$REPO/tiny_audio_amd/$lib python scripts/gemm_lib_probe.py 2>/dev/null | tail -1
  done
done | tee $OUT/r3ad_gemm_lib_probe.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "gemm or b32 or smoke" 2>&1 | tail -3 | tee $OUT/r3ad_pytest.log
for i in 1 2 3; do
  for lib in libta355_base.so libta355_m256.so libta355.so; do
    TA355_LIB=$REPO/tiny_audio_amd/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-logits-full --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/r3ad_ab_store_merge_libs.txt
