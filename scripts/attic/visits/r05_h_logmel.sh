#!/bin/bash
# round 5 visit h: log-mel with per-workgroup maxima (no init launch, no atomics): parity + rate
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu -k "logmel or collator or mel or full_depth_one_clip_vs_oracle or smoke" 2>&1 | tail -4
python - <<'PY' > gpurun_out/r05_h_logmel_rate.txt 2>&1
import torch, numpy as np, sys
sys.path.insert(0, '.')
from tiny_audio_amd.asr_processing import LogMelFeatureExtractor
dev = "cuda"
fe = LogMelFeatureExtractor(128, dev)
for B, n in ((32, 160000), (32, 480000), (5, 160077), (1, 16000), (64, 160000)):
    wav = 0.1 * torch.randn(B, n, device=dev)
    lens = torch.full((B,), n, device=dev, dtype=torch.int64)
    filler = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
    for _ in range(3): fe.extract(wav, lens)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(4): torch.mm(filler, filler)
    a.record()
    for _ in range(20): fe.extract(wav, lens)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    nbytes = B * (n * 4 + 128 * (n // 160) * 4)
    print(f"B={B} samples={n}: {us:.2f} us per call (both launches), {nbytes / us / 1e3:.0f} GB/s = {nbytes / us / 1e3 / 8000:.4f} of 8 TB/s")
PY
cat gpurun_out/r05_h_logmel_rate.txt
