#!/usr/bin/env python3
"""Weight-gradient product dW = dY^T X at the LM's shapes (M = 6144 tokens): the TN kernel (csrc/gemm_tn.hip) against two
transposes + the NT GEMM with split-K (what ta_lm_backward used before)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops
DEV, BF16, F32 = "cuda", torch.bfloat16, torch.float32


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


M = 6144
for name, Ny, Nx in (("dWqkv", 4096, 1024), ("dWo", 1024, 2048), ("dWgu", 6144, 1024), ("dWd", 1024, 3072)):
    Y = torch.randn(M, Ny, device=DEV).to(BF16); X = torch.randn(M, Nx, device=DEV).to(BF16)
    out = torch.zeros(Ny, Nx, device=DEV, dtype=F32)
    t_tn = timeit(lambda: ops.gemm_tn(Y, X, out=out, accumulate=True))

    def old():
        yt = ops.transpose_to_bf16(Y, ld_out=M); xt = ops.transpose_to_bf16(X, ld_out=M)
        ops.gemm_nt(yt, xt, Ny, Nx, M, out=out, residual=out, splits=2)
    t_old = timeit(old)
    fl = 2.0 * M * Ny * Nx
    print(f"{name:6s} [{Ny}x{Nx}]  TN {t_tn:7.1f} us ({fl / t_tn / 1e6:6.0f} TF/s)   transposes + NT {t_old:7.1f} us ({fl / t_old / 1e6:6.0f} TF/s)", flush=True)
