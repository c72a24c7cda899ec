#!/usr/bin/env python3
"""log-mel kernel (the one form the library has since round 5: mixed-radix register FFT, persistent workgroups, per-workgroup clip
maxima).  B = 32 clips of 10 s; bytes = 640 KB in + 512 KB out per clip.  Same-box A/B of two builds: TA355_LIB=<other libta355.so>
and a TAG for the saved output (TAG=old python scripts/logmel_bench.py; TAG=new ...: the second run prints the difference)."""
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
tag = os.environ.get("TAG", "cur")
print(f"form={tag}: {us:.1f} us per {B} clips = {B * 1.152e6 / us / 1e3:.1f} GB/s; checksum {float(f.double().sum()):.6f} max {float(f.max()):.5f}")
torch.save(f.cpu(), f"/tmp/logmel_{tag}.pt")
import glob
for path in sorted(glob.glob("/tmp/logmel_*.pt")):
    other = os.path.basename(path)[len("logmel_"):-3]
    if other != tag:
        y = torch.load(f"/tmp/logmel_{other}.pt")
        print(f"  {tag} vs {other}: max abs diff", float((f.cpu() - y).abs().max()), "mean", float((f.cpu() - y).abs().mean()))
