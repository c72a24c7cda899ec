#!/bin/bash
# round 4, visit f: the recorded-and-replayed step GEMMs, full-vocabulary logits vs the oracle, streaming penalties
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_f
timeout 1500 python -m pytest tests/test_gpu_round4.py -x -q -k "real_b32_step" -s 2>&1 | tail -15 > gpurun_out/r04_f/pytest_step_gemms.log
tail -6 gpurun_out/r04_f/pytest_step_gemms.log
timeout 1500 python -m pytest "tests/test_gpu_round3.py::test_full_depth_one_clip_vs_oracle[mlp]" tests/test_gpu_parity.py -x -q -k "full_depth or streaming_penalties or repetition_penalty" -s 2>&1 | tail -15 > gpurun_out/r04_f/pytest_logits.log
tail -8 gpurun_out/r04_f/pytest_logits.log
cp gpurun_out/r03_full_depth_drift.json gpurun_out/r04_f/ 2>/dev/null
