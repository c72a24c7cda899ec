#!/usr/bin/env python3
"""log-mel kernel: FFT form vs the exact DFT-as-GEMM form (TA355_LOGMEL_DFT=1 must be set BEFORE the first call of a process:
run this script twice).  B = 32 clips of 10 s; bytes = 640 KB in + 512 KB out per clip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd.asr_processing import LogMelFeatureExtractor
fe = LogMelFeatureExtractor(128, "cuda")
B = 32
wav = 0.1 * torch.randn(B, 160000, device="cuda"); lens = torch.full((B,), 160000, device="cuda", dtype=torch.int64)
for _ in range(3):
    f, m = fe.extract(wav, lens)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    f, m = fe.extract(wav, lens)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 20 * 1e3
print(f"TA355_LOGMEL_DFT={os.environ.get('TA355_LOGMEL_DFT', '0')}: {us:.1f} us per {B} clips = {B * 1.152e6 / us / 1e3:.1f} GB/s; checksum {float(f.double().sum()):.6f} max {float(f.max()):.5f}")
torch.save(f.cpu(), f"/tmp/logmel_{os.environ.get('TA355_LOGMEL_DFT', '0')}.pt")
if os.path.exists("/tmp/logmel_0.pt") and os.path.exists("/tmp/logmel_1.pt"):
    x, y = torch.load("/tmp/logmel_0.pt"), torch.load("/tmp/logmel_1.pt")
    print("FFT vs DFT: max abs diff", float((x - y).abs().max()), "mean", float((x - y).abs().mean()))
