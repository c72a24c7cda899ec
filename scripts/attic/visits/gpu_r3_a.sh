#!/bin/bash
# round 3, visit a: the new full-depth / bench-shape parity tests, VALU issue-cost probe, vendor-library calibration, bench line
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_round3.py -q -x -s 2>&1 | tail -40 > $OUT/r3a_pytest.log
echo "pytest rc=$?"; tail -15 $OUT/r3a_pytest.log
timeout 120 scripts/probe/valu_rate > $OUT/r3a_valu_rate.txt 2>&1; cat $OUT/r3a_valu_rate.txt
timeout 300 python scripts/blaslt_calibration.py > $OUT/r3a_blaslt_calibration.txt 2>&1; cat $OUT/r3a_blaslt_calibration.txt
timeout 400 python bench.py > $OUT/r3a_bench_mlp.json 2> $OUT/r3a_bench_mlp.err; cat $OUT/r3a_bench_mlp.json
