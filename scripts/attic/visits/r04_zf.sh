#!/bin/bash
# round 4, visit zf: min_new_tokens on the GPU + smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_zf
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "generate" 2>&1 | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
