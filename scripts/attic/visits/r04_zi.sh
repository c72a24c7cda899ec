#!/bin/bash
# round 4, visit zi: sampling kernels + generate(do_sample) on the GPU
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -k "warp or sample or do_sample" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "generate" 2>&1 | tail -2
