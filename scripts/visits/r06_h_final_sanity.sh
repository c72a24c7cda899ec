#!/bin/bash
# round 6, last visit: what the driver runs at round end, on the final commit -- build check, smoke(), the full GPU suite, the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_h; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 1800 python -m pytest tests -m gpu -q -rs > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -1 $O/bench_default.json | cut -c1-420
