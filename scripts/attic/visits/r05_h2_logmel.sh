#!/bin/bash
# round 5 visit h2: log-mel kernel times (rocprofv3 --kernel-trace --stats) + parity
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
cat > /tmp/lm_rate.py <<PY
import torch, sys
sys.path.insert(0, '$REPO')
from tiny_audio_amd.asr_processing import LogMelFeatureExtractor
dev = "cuda"
fe = LogMelFeatureExtractor(128, dev)
wav = 0.1 * torch.randn(32, 160000, device=dev)
lens = torch.full((32,), 160000, device=dev, dtype=torch.int64)
for _ in range(30): fe.extract(wav, lens)
torch.cuda.synchronize()
PY
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lmprof -o b -- python /tmp/lm_rate.py > /tmp/lmprof.log 2>&1 < /dev/null)
S=$(find /tmp/lmprof -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$S" ]; then python -c "
import csv,sys
for r in csv.DictReader(open('$S')):
    if 'logmel' in r['Name']: print(r['Name'][:60], 'calls', r['Calls'], 'avg_ns', r['AverageNs'], 'min', r['MinNs'], 'max', r['MaxNs'])
" > gpurun_out/r05_h_logmel_kernel_stats.txt 2>&1; else tail -5 /tmp/lmprof.log > gpurun_out/r05_h_logmel_kernel_stats.txt; fi
cat gpurun_out/r05_h_logmel_kernel_stats.txt

