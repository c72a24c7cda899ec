#!/usr/bin/env python3
"""Experiment: start the first-round workgroups of every other CU late (TA355_GEMM_DEBUG = cycles << 8 | 0), so that the
rounds of a multi-round K-short GEMM stop finishing in lock step.  usage: gemm_stagger.py [--reps 30]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops

DEV, BF16 = "cuda", torch.bfloat16
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 30
shapes = [("enc_qk", 16000, 2560, 1280), ("enc_qkv", 16000, 3840, 1280), ("enc_fc1", 16000, 5120, 1280), ("enc_o", 16000, 1280, 1280),
          ("lm_gu", 6144, 6144, 1024), ("lm_qkv", 6144, 4096, 1024), ("big_fc1", 64000, 5120, 1280), ("sq8192", 8192, 8192, 8192)]
waits_us = [0, 5, 10, 15, 20, 30]


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


os.environ["TA355_GEMM_VARIANT"] = "4"
print(f"{'shape':10s} {'tiles':>6s} " + " ".join(f"{str(w) + 'us':>8s}" for w in waits_us) + "   (us per launch; late start of every other CU's first workgroup)")
for name, M, N, K in shapes:
    A = torch.randn(M, K, device=DEV).to(BF16)
    W = (torch.randn(N, K, device=DEV) / K ** 0.5).to(BF16)
    out = torch.empty(M, N, device=DEV, dtype=BF16)
    cells = []
    for w in waits_us:
        os.environ["TA355_GEMM_DEBUG"] = str((int(w * 1700) >> 8) << 8)
        cells.append(f"{timeit(lambda: ops.gemm_nt(A, W, M, N, K, out=out)):8.1f}")
    os.environ["TA355_GEMM_DEBUG"] = "0"
    print(f"{name:10s} {((M + 255) // 256) * ((N + 319) // 320):6d} " + " ".join(cells), flush=True)
