#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -s -k "true_vocabulary or warp or sample" 2>&1 | grep -E "warp \+ sample|passed|failed|Error|assert" | head
