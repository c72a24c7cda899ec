#!/usr/bin/env python3
"""Back-to-back times of the step's ping-pong GEMM shapes (bias + GELU epilogue) for whatever library TA355_LIB names: run once per
build and compare (the library is bound at import)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops

SHAPES = [("enc fc1", 16000, 5120, 1280, ""), ("enc q|k|v", 16000, 3840, 1280, ""), ("enc fc2", 16000, 1280, 5120, ""), ("enc o_proj", 16000, 1280, 1280, ""),
          ("lm gate|up", 6016, 6144, 1024, ""), ("lm d(act)", 6016, 3072, 1024, ""), ("lm q|k|v", 6016, 4096, 1024, ""), ("lm d(attn-out)", 6016, 2048, 1024, ""),
          ("enc fc1 (variant 3)", 16000, 5120, 1280, "3"), ("head 1184 x 151680", 1184, 151680, 1024, "")]

def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

out = []
for name, M, N, K, var in SHAPES:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    if var: os.environ["TA355_GEMM_VARIANT"] = var
    else: os.environ.pop("TA355_GEMM_VARIANT", None)
    out.append("%s %.1f" % (name, t(lambda: ops.gemm_nt(A, W, out=o, bias=bias, act=1))))
    del A, W, o
os.environ.pop("TA355_GEMM_VARIANT", None)
print(os.path.basename(os.environ.get("TA355_LIB", "libta355.so")), "|", " | ".join(out))
