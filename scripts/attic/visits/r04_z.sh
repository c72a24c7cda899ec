#!/bin/bash
# round 4, visit z: log-mel: what the kernel costs without its transform (dbg 1), without the mel stage (dbg 2), without both (dbg 3)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_z
for m in 1 0; do for d in 0 1 2 3; do
echo -n "MFMA=$m dbg=$d  "; TA355_LOGMEL_MFMA=$m TA355_LOGMEL_DEBUG=$d python scripts/logmel_bench.py 2>/dev/null | head -1
done; done | tee gpurun_out/r04_z/logmel_dbg.txt
