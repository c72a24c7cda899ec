#!/usr/bin/env python3
"""log-mel kernel: the MFMA sub-transform form (default), the register-FFT form (TA355_LOGMEL_MFMA=0) and the exact DFT-as-GEMM form
(TA355_LOGMEL_DFT=1); the knobs are read at the first call of a process: run this script once per form.  B = 32 clips of 10 s; bytes = 640 KB in + 512 KB out per clip."""
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
tag = "dft" if os.environ.get("TA355_LOGMEL_DFT", "0") == "1" else ("fft" if os.environ.get("TA355_LOGMEL_MFMA", "1") == "0" else "mfma")
print(f"form={tag}: {us:.1f} us per {B} clips = {B * 1.152e6 / us / 1e3:.1f} GB/s; checksum {float(f.double().sum()):.6f} max {float(f.max()):.5f}")
torch.save(f.cpu(), f"/tmp/logmel_{tag}.pt")
for other in ("dft", "fft", "mfma"):
    if other != tag and os.path.exists(f"/tmp/logmel_{other}.pt"):
        y = torch.load(f"/tmp/logmel_{other}.pt")
        print(f"  {tag} vs {other}: max abs diff", float((f.cpu() - y).abs().max()), "mean", float((f.cpu() - y).abs().mean()))
