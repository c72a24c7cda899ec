#!/usr/bin/env python3
"""What does the vendor library reach on the step's GEMM shapes?  torch.matmul (hipBLASLt / rocBLAS behind it) against
ta_gemm_bf16_nt on every (M, N, K) the B = 32 step launches, both WARM (back-to-back launches on the same operands) and
COLD (a 1-GB fill between launches, so neither L2 nor the infinity cache holds an operand), plain bf16 in / bf16 out.
The library is NOT on the product path (no fused epilogues, no row maps): this is a calibration of how far the hand-written
kernels are from what the best available generic kernel does on this chip at these shapes.

    python scripts/blaslt_calibration.py [--reps 20]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from tiny_audio_amd import ops

SHAPES = [("enc q|k", 16000, 2560, 1280), ("enc V^T", 1280, 16000, 1280), ("enc o_proj", 16000, 1280, 1280),
          ("enc fc1", 16000, 5120, 1280), ("enc fc2", 16000, 1280, 5120), ("enc conv2", 16000, 1280, 3840),
          ("lm q|k|v", 6144, 4096, 1024), ("lm o", 6144, 1024, 2048), ("lm gate|up", 6144, 6144, 1024),
          ("lm down", 6144, 1024, 3072), ("lm d(act)", 6144, 3072, 1024), ("lm d(xn) gu", 6144, 1024, 6144),
          ("lm d(ao)", 6144, 2048, 1024), ("lm d(xn) qkv", 6144, 1024, 4096), ("sq 8192", 8192, 8192, 8192)]


def timeit(fn, reps, flush=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if flush is None:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3
    tot = 0.0
    for _ in range(reps):
        flush.fill_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = "cuda"
    flush = torch.empty(256 * 1024 * 1024, device=dev, dtype=torch.float32)
    print(f"{'shape':<14}{'M':>7}{'N':>7}{'K':>7} | {'lib warm':>9}{'TF/s':>7} {'lib cold':>9}{'TF/s':>7} | {'ta355 warm':>11}{'TF/s':>7} {'ta355 cold':>11}{'TF/s':>7}")
    for name, M, N, K in SHAPES:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        lib = lambda: torch.matmul(A, W.t(), out=out)
        mine = lambda: ops.gemm_nt(A, W, out=out)
        fl = 2.0 * M * N * K
        r = []
        for fn in (lib, mine):
            w = timeit(fn, a.reps)
            c = timeit(fn, max(4, a.reps // 4), flush)
            r += [w, fl / w / 1e6, c, fl / c / 1e6]
        print(f"{name:<14}{M:>7}{N:>7}{K:>7} | {r[0]:>9.1f}{r[1]:>7.0f} {r[2]:>9.1f}{r[3]:>7.0f} | {r[4]:>11.1f}{r[5]:>7.0f} {r[6]:>11.1f}{r[7]:>7.0f}")


if __name__ == "__main__":
    main()
