#!/usr/bin/env python3
"""Does the layout of the A operand decide what a COLD ping-pong GEMM costs?  The same product with A row-major [M, K] and with A in
[M/64][K/64][64][64] blocks (8 KB contiguous per 64 x 64 block: the DMA of a K tile then reads 1 KB runs instead of 128-B pieces
K * 2 bytes apart), warm and cold (1 GB written between launches), on the encoder's shapes.  Experiment: TA355_GEMM_DEBUG bit 10."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops

SHAPES = [("enc fc2", 16000, 1280, 5120), ("enc o_proj", 16000, 1280, 1280), ("enc fc1", 16000, 5120, 1280), ("enc q|k|v", 16000, 3840, 1280),
          ("lm q|k|v", 6016, 4096, 1024), ("lm gate|up", 6016, 6144, 1024)]
flush = torch.empty(256 * 1024 * 1024, device="cuda", dtype=torch.float32)

def t(fn, cold, reps=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        if cold: flush.fill_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / reps * 1e3

print(f"{'shape':<12}{'M':>6}{'N':>6}{'K':>6} | row-major warm / cold us | blocked warm / cold us | max |diff|")
for name, M, N, K in SHAPES:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    Ab = A.view(M // 64, 64, K // 64, 64).permute(0, 2, 1, 3).contiguous()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); out2 = torch.empty_like(out)
    os.environ["TA355_GEMM_DEBUG"] = "0"
    f0 = lambda: ops.gemm_nt(A, W, out=out)
    w0, c0 = t(f0, False), t(f0, True)
    os.environ["TA355_GEMM_DEBUG"] = "1024"
    f1 = lambda: ops.gemm_nt(Ab.view(M, K), W, out=out2)
    w1, c1 = t(f1, False), t(f1, True)
    os.environ["TA355_GEMM_DEBUG"] = "0"
    print(f"{name:<12}{M:>6}{N:>6}{K:>6} | {w0:8.1f} / {c0:8.1f} | {w1:8.1f} / {c1:8.1f} | {float((out.float() - out2.float()).abs().max()):.3g}")
