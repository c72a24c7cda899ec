#!/bin/bash
# round 4, visit y: log-mel with the sub-transforms on the f32 matrix cores: parity tests + microbench vs the register FFT
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_y
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round2.py -x -q -k "logmel or collator_end_to_end" 2>&1 | tail -15 > gpurun_out/r04_y/pytest.log
tail -6 gpurun_out/r04_y/pytest.log
for rep in 1 2; do
TA355_LOGMEL_DFT=1 python scripts/logmel_bench.py 2>/dev/null
TA355_LOGMEL_MFMA=0 python scripts/logmel_bench.py 2>/dev/null
python scripts/logmel_bench.py 2>/dev/null
done | tee gpurun_out/r04_y/logmel_bench.txt
