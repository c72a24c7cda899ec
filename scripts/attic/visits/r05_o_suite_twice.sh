#!/bin/bash
# flake hunt: the whole -m gpu suite twice on one box, then smoke
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2; do
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider < /dev/null > gpurun_out/r05_o_suite_$i.log 2>&1
  tail -3 gpurun_out/r05_o_suite_$i.log
done
timeout 300 python __graft_entry__.py --smoke < /dev/null 2>&1 | tail -2
