#!/usr/bin/env python3
"""Which operand makes a cold GEMM slow?  Flush the caches (1 GB written), then optionally re-touch W or A (a read that leaves
them in L2 / the infinity cache), then time ONE launch.  usage: gemm_cold_which.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops

DEV, BF16 = "cuda", torch.bfloat16
flush = torch.empty(256 * 1024 * 1024, device=DEV, dtype=torch.int32)
shapes = [("enc_o", 16000, 1280, 1280, True), ("enc_fc1", 16000, 5120, 1280, False), ("enc_fc2", 16000, 1280, 5120, True),
          ("lm_down", 6144, 1024, 3072, True), ("lm_gu", 6144, 6144, 1024, False)]
print(f"{'shape':9s} {'warm':>8s} {'cold':>8s} {'cold+W':>8s} {'cold+A':>8s} {'cold+A+W':>9s}   us per launch (median of 7)")
for name, M, N, K, hasres in shapes:
    A = torch.randn(M, K, device=DEV).to(BF16)
    W = (torch.randn(N, K, device=DEV) / K ** 0.5).to(BF16)
    out = torch.randn(M, N, device=DEV).to(BF16)
    fn = lambda: ops.gemm_nt(A, W, M, N, K, out=out, residual_bf16=out if hasres else None)
    res = []
    for mode in ("warm", "cold", "W", "A", "AW"):
        ts = []
        for _ in range(7):
            if mode != "warm":
                flush.fill_(1)
                if "W" in mode:
                    W.float().sum()
                if "A" in mode:
                    A.float().sum()
            else:
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort(); res.append(ts[3])
    print(f"{name:9s} " + " ".join(f"{t:8.1f}" for t in res[:4]) + f" {res[4]:9.1f}", flush=True)
