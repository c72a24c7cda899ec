#!/usr/bin/env python3
"""GEMM / attention micro-benchmarks on the step's real shapes (B=32): TF/s per shape, random bf16 data.
usage: gemm_bench.py [--reps N] [--only gemm|attn]"""
import sys
import os
import json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops

DEV = "cuda"
BF16, F32 = torch.bfloat16, torch.float32
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else "all"


def timeit(fn, reps=reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


_flush = None


def time_cold(fn, reps=10):
    """each launch timed alone after a 1 GB write pass (L2 + infinity cache hold none of the operands)"""
    global _flush
    if _flush is None:
        _flush = torch.empty(512 * 1024 * 1024, device=DEV, dtype=torch.int16)
    tot = 0.0
    for _ in range(reps):
        _flush.fill_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / reps * 1e-3


if "--cold" in sys.argv:
    timeit = lambda fn: time_cold(fn)

res = {}
if only in ("all", "gemm"):
    shapes = [  # name, M, N, K, out_bf16, act, residual
        ("enc_qkv", 16000, 3840, 1280, True, 0, False), ("enc_o", 16000, 1280, 1280, False, 0, True),
        ("enc_fc1", 16000, 5120, 1280, True, 1, False), ("enc_fc2", 16000, 1280, 5120, False, 0, True),
        ("conv2", 16000, 1280, 3840, False, 1, False),
        ("lm_qkv", 6144, 4096, 1024, True, 0, False), ("lm_o", 6144, 1024, 2048, False, 0, True),
        ("lm_gu", 6144, 6144, 1024, True, 0, False), ("lm_down", 6144, 1024, 3072, False, 0, True),
        ("lm_dact", 6144, 3072, 1024, True, 0, False), ("lm_dxn_gu", 6144, 1024, 6144, False, 0, False),
        # the LM's N <= 2048 GEMMs as the step runs them now (bf16 stream: bf16 out, bf16 residual in place)
        ("lmb_o", 6144, 1024, 2048, True, 0, "bf16"), ("lmb_down", 6144, 1024, 3072, True, 0, "bf16"),
        ("lmb_dao", 6144, 2048, 1024, True, 0, False), ("lmb_dxn_qkv", 6144, 1024, 4096, True, 0, False),
        ("lmb_dxn_gu", 6144, 1024, 6144, True, 0, False),
        ("head_fwd", 1152, 151680, 1024, False, 0, False), ("sq4096", 4096, 4096, 4096, True, 0, False),
        ("sq8192", 8192, 8192, 8192, True, 0, False),
    ]
    variants = sys.argv[sys.argv.index("--variants") + 1].split(",") if "--variants" in sys.argv else [""]
    if "--match" in sys.argv:
        shapes = [s_ for s_ in shapes if sys.argv[sys.argv.index("--match") + 1] in s_[0]]
    for name, M, N, K, obf, act, hasres in shapes:
      for var in variants:
        os.environ["TA355_GEMM_VARIANT"] = var                    # the one tile knob the library has (0-5, 10, 12; "" = automatic)
        name_v = name + (":v" + var if var else "")
        A = (torch.randn(M, K, device=DEV) * 1.0).to(BF16)
        W = (torch.randn(N, K, device=DEV) / K ** 0.5).to(BF16)
        out = torch.empty(M, N, device=DEV, dtype=BF16 if obf else F32)
        bias = torch.randn(N, device=DEV) if act else None
        resid = out if hasres is True else None
        resid_bf = out if hasres == "bf16" else None
        if hasres:
            out.zero_()
        wb = "--wblocked" in sys.argv
        Wx = W.view(N // 64, 64, K // 64, 64).permute(0, 2, 1, 3).contiguous() if wb else W
        t = timeit(lambda: ops.gemm_nt(A, Wx, M, N, K, out=out, bias=bias, residual=resid, residual_bf16=resid_bf, act=act, w_blocked=wb))
        res[name_v] = round(2.0 * M * N * K / t / 1e12, 1)
        print(f"{name_v:14s} M={M:6d} N={N:6d} K={K:5d}  {t * 1e6:8.1f} us  {res[name_v]:7.1f} TF/s", flush=True)
        del A, W, out
    os.environ["TA355_GEMM_VARIANT"] = ""
if only in ("all", "attn"):
    for name, B, Hq, Hkv, L, hd, causal in [("enc_attn", 32, 20, 20, 500, 64, False), ("lm_attn_fwd", 32, 16, 8, 192, 128, True)]:
        Q = torch.randn(B, Hq, L, hd, device=DEV).to(BF16)
        K_ = torch.randn(B, Hkv, L, hd, device=DEV).to(BF16)
        Lp = ops.pad64(L)
        VT = torch.zeros(B, Hkv, hd, Lp, device=DEV, dtype=BF16); VT[..., :L] = torch.randn(B, Hkv, hd, L, device=DEV).to(BF16)
        sc = float(os.environ.get('ATTN_SCALE', hd ** -0.5))
        t = timeit(lambda: ops.attention_fwd(Q, K_, VT, L, causal, sc, None, want_lse=causal))
        fl = 4.0 * B * Hq * L * L * hd * (0.5 if causal else 1.0)
        res[name] = round(fl / t / 1e12, 1)
        print(f"{name:12s} {t * 1e6:8.1f} us  {res[name]:7.1f} TF/s", flush=True)
print(json.dumps(res))
