#!/usr/bin/env python3
"""Row-merged stores of the 256-column ping-pong tiles (TA355_GEMM_DEBUG bit 11 = the pair form) on single launches: results must be
bit-identical, times are back-to-back averages."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops

SHAPES = [("enc fc1 (auto: 256x320)", 16000, 5120, 1280, ""), ("enc q|k|v (auto)", 16000, 3840, 1280, ""), ("enc fc2 (auto)", 16000, 1280, 5120, ""),
          ("lm gate|up (auto)", 6016, 6144, 1024, ""), ("lm d(act) (auto)", 6016, 3072, 1024, ""), ("ragged 1000x700x192 (variant 4)", 1000, 704, 192, "4"),
          ("enc fc1 (variant 3)", 16000, 5120, 1280, "3"), ("enc q|k|v (variant 3)", 16000, 3840, 1280, "3"), ("lm q|k|v (auto: 192x256)", 6016, 4096, 1024, ""),
          ("lm d(attn-out) (auto)", 6016, 2048, 1024, ""), ("sq 8192 (variant 3)", 8192, 8192, 8192, "3"), ("ragged 1000x700x192 (variant 3)", 1000, 704, 192, "3"),
          ("ragged 777x264x64 (variant 12)", 777, 264, 64, "12")]

def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for name, M, N, K, var in SHAPES:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    o0 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); o1 = torch.empty_like(o0)
    if var: os.environ["TA355_GEMM_VARIANT"] = var
    else: os.environ.pop("TA355_GEMM_VARIANT", None)
    os.environ["TA355_GEMM_DEBUG"] = "2048"
    f0 = lambda: ops.gemm_nt(A, W, out=o0, bias=bias, act=1)
    t0 = t(f0)
    os.environ["TA355_GEMM_DEBUG"] = "0"
    f1 = lambda: ops.gemm_nt(A, W, out=o1, bias=bias, act=1)
    t1 = t(f1)
    print(f"{name:<34} pairs {t0:8.1f} us   row-merged {t1:8.1f} us   identical {bool(torch.equal(o0, o1))}")
os.environ.pop("TA355_GEMM_VARIANT", None)
