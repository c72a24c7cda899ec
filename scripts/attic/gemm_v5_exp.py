#!/usr/bin/env python3
"""What bounds the 192x128 one-tile-per-CU GEMM (v5): times its main loop with parts removed (results are wrong on purpose).
TA355_GEMM_DEBUG = 16 * EXP: 1 no W fragment reads, 2 no A fragment reads, 3 neither, 4 no DMA after the prologue, 5 twelve 32x32x16 MFMAs per half-step instead of 24 16x16x32."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops

DEV, BF16 = "cuda", torch.bfloat16


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


os.environ["TA355_GEMM_VARIANT"] = "10"
names = ["all", "no W reads", "no A reads", "no reads", "no DMA", "32x32x16"]
print(f"{'shape':12s} " + " ".join(f"{n:>11s}" for n in names) + "   us per launch; half-steps; cycles per half-step at 1.7 GHz from the K slope")
res = {}
for name, M, N, K in [("lm_o", 6144, 1024, 2048), ("lm_dxn_gu", 6144, 1024, 6144)]:
    A = torch.randn(M, K, device=DEV).to(BF16)
    W = (torch.randn(N, K, device=DEV) / K ** 0.5).to(BF16)
    out = torch.empty(M, N, device=DEV, dtype=BF16)
    ts = []
    for ex in range(6):
        os.environ["TA355_GEMM_DEBUG"] = str(16 * ex)
        ts.append(timeit(lambda: ops.gemm_nt(A, W, M, N, K, out=out)))
    res[name] = (K, ts)
    print(f"{name:12s} " + " ".join(f"{t:11.1f}" for t in ts))
(k0, t0), (k1, t1) = res["lm_o"], res["lm_dxn_gu"]
print("cycles/half-step " + " ".join(f"{(b - a) / ((k1 - k0) / 32) * 1700:11.0f}" for a, b in zip(t0, t1)))
os.environ["TA355_GEMM_DEBUG"] = "0"; os.environ["TA355_GEMM_VARIANT"] = ""
