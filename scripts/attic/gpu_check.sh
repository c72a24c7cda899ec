#!/bin/bash
# One GPU-box visit: kernel tests, parity tests, smoke, short bench.  Full logs land in gpurun_out/.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
STAGES=${STAGES:-"kernels parity smoke bench"}
for s in $STAGES; do
  case $s in
    kernels) timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -v --tb=short -p no:cacheprovider > gpurun_out/kernels.log 2>&1; echo "kernels rc=$?";;
    parity)  timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -v --tb=short -p no:cacheprovider > gpurun_out/parity.log 2>&1; echo "parity rc=$?";;
    smoke)   timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?";;
    bench)   timeout 900 python bench.py ${BENCH_ARGS:---steps 4 --warmup 2 --batch 8} > gpurun_out/bench.log 2>&1; echo "bench rc=$?";;
  esac
done
for f in kernels parity; do [ -f gpurun_out/$f.log ] && { echo "== $f"; grep -E "PASSED|FAILED|ERROR|passed|failed|error" gpurun_out/$f.log | tail -60; }; done
for f in smoke bench; do [ -f gpurun_out/$f.log ] && { echo "== $f"; tail -5 gpurun_out/$f.log; }; done
