#!/usr/bin/env python3
"""Round 4: the one-wave-per-SIMD AGPR GEMM (variants 13 = 256x256, 14 = 256x320, 15 = 192x256) against the tile the launch-time
model picks today, on the B = 32 step's shapes, cold (1 GB written before every launch) and warm (back to back).
usage: gemm_v7_ab.py [--reps N] [--variants ,13,14,15]   ('' = the automatic choice)"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops

DEV, BF16, F32 = "cuda", torch.bfloat16, torch.float32
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 10
variants = sys.argv[sys.argv.index("--variants") + 1].split(",") if "--variants" in sys.argv else ["", "13", "14", "15"]
_flush = torch.empty(512 * 1024 * 1024, device=DEV, dtype=torch.int16)


def time_cold(fn):
    tot = 0.0
    for _ in range(reps):
        _flush.fill_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / reps * 1e3


def time_warm(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(2 * reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (2 * reps) * 1e3


# name, M, N, K, epilogue: the launches of the B = 32 step (tests/test_gpu_round3.py STEP_GEMMS) + two squares
SHAPES = [("enc.qkv rope", 16000, 3840, 1280, "rope"), ("enc.o_proj", 16000, 1280, 1280, "bias_res"), ("enc.fc1", 16000, 5120, 1280, "gelu"),
          ("enc.fc2", 16000, 1280, 5120, "bias_res"), ("conv2-like", 16000, 1280, 3840, "gelu"),
          ("lm.qkv", 6144, 4096, 1024, "plain"), ("lm.o", 6144, 1024, 2048, "res"), ("lm.gate|up", 6144, 6144, 1024, "plain"),
          ("lm.down", 6144, 1024, 3072, "res"), ("lm.d(act)", 6144, 3072, 1024, "plain"), ("lm.d(xn) gu", 6144, 1024, 6144, "plain"),
          ("lm.d(attn-out)", 6144, 2048, 1024, "plain"), ("lm.d(xn) qkv", 6144, 1024, 4096, "plain"),
          ("sq4096", 4096, 4096, 4096, "plain"), ("sq8192", 8192, 8192, 8192, "plain")]
if "--match" in sys.argv:
    SHAPES = [s for s in SHAPES if sys.argv[sys.argv.index("--match") + 1] in s[0]]
res = {}
print(f"{'shape':16s} {'M':>6s} {'N':>6s} {'K':>5s}  " + "  ".join(f"{('v' + v) if v else 'auto':>18s}" for v in variants) + "   (cold us / warm us / warm TF/s)")
for name, M, N, K, epi in SHAPES:
    A = torch.randn(M, K, device=DEV).to(BF16)
    W = (torch.randn(N, K, device=DEV) / K ** 0.5).to(BF16)
    out = torch.zeros(M, N, device=DEV, dtype=BF16)
    bias = torch.randn(N, device=DEV)
    kw = {}
    if epi == "bias_res":
        kw = dict(bias=bias, residual_bf16=out)
    elif epi == "res":
        kw = dict(residual_bf16=torch.zeros(M, N, device=DEV, dtype=BF16))
    elif epi == "gelu":
        kw = dict(bias=bias, act=1)
    elif epi == "rope":
        tab = torch.randn(1500, 16, 2, device=DEV)
        kw = dict(bias=bias, act=2, rope=(tab, 500, 2560))
    row = []
    for v in variants:
        os.environ["TA355_GEMM_VARIANT"] = v
        fn = lambda: ops.gemm_nt(A, W, M, N, K, out=out, **kw)
        c, w = time_cold(fn), time_warm(fn)
        res[f"{name}:{v or 'auto'}"] = (round(c, 1), round(w, 1))
        row.append(f"{c:6.1f}/{w:6.1f}/{2.0 * M * N * K / w / 1e6:5.0f}")
    os.environ["TA355_GEMM_VARIANT"] = ""
    print(f"{name:16s} {M:6d} {N:6d} {K:5d}  " + "  ".join(f"{r:>18s}" for r in row), flush=True)
    del A, W, out
print(json.dumps(res))
