#!/bin/bash
# round 5 visit c: after the prune (ABI 3): the whole -m gpu suite + the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r05_c_pytest_gpu.log
tail -12 gpurun_out/r05_c_pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r05_c_bench_mlp.json 2> gpurun_out/r05_c_bench_mlp.err
python -c "
import json; d=json.load(open('gpurun_out/r05_c_bench_mlp.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['logits_full']['ms_per_step'], d['streams_other']['ms_per_step'])"
