#!/usr/bin/env python3
"""K sweep of one GEMM shape: time(K) = fixed (prologue + epilogue + launch) + per-K-tile cost.
usage: gemm_ksweep.py M N variant [f32res]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops
M, N, var = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
f32res = len(sys.argv) > 4
os.environ["TA355_GEMM_VARIANT"] = var
BF16, F32 = torch.bfloat16, torch.float32
prev = None
for K in (64, 128, 256, 512, 1024, 1280, 2560, 5120, 10240):
    A = torch.randn(M, K, device="cuda").to(BF16); W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(BF16)
    out = torch.zeros(M, N, device="cuda", dtype=F32 if f32res else BF16)
    fn = lambda: ops.gemm_nt(A, W, out=out, residual=out if f32res else None)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20 * 1e3
    line = f"M={M} N={N} K={K:6d} v{var} {'f32+res' if f32res else 'bf16'}: {t:8.1f} us  {2.0*M*N*K/t/1e6:7.1f} TF/s"
    if prev: line += f"   d/ktile = {(t - prev[1]) / ((K - prev[0]) / 64):.3f} us"
    print(line, flush=True); prev = (K, t)
