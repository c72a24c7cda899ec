#!/usr/bin/env python3
"""Experiment: are the epilogues of a multi-round GEMM slow because every CU stores at the same moment?  The same GEMM as
ONE launch vs TWO concurrent launches over row halves with different tile widths (their rounds drift apart), and the
epilogue cost of a launch that occupies only 32 CUs (HBM far from saturated)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops

DEV, BF16 = "cuda", torch.bfloat16
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def gemm(A, W, out, bias, act, variant, dbg=0):
    os.environ["TA355_GEMM_VARIANT"] = variant
    os.environ["TA355_GEMM_DEBUG"] = str(dbg)
    ops.gemm_nt(A, W, A.shape[0], W.shape[0], W.shape[1], out=out, bias=bias, act=act)


for name, M, N, K, act in [("enc_fc1", 16000, 5120, 1280, 1), ("enc_qk", 16000, 2560, 1280, 0), ("lm_gu", 6144, 6144, 1024, 0)]:
    A = torch.randn(M, K, device=DEV).to(BF16)
    W = (torch.randn(N, K, device=DEV) / K ** 0.5).to(BF16)
    bias = torch.randn(N, device=DEV) if act else None
    out = torch.empty(M, N, device=DEV, dtype=BF16)
    one = run(lambda: gemm(A, W, out, bias, act, "4"))
    one1 = run(lambda: gemm(A, W, out, bias, act, "4", 1))
    res = {}
    for split, va, vb in ((0.5, "4", "4"), (0.5, "4", "3"), (0.55, "4", "3"), (0.5, "7", "6")):
        h = int(M * split) // 256 * 256
        cur = torch.cuda.current_stream()

        def two():
            e = torch.cuda.Event(); e.record(cur)
            s1.wait_event(e); s2.wait_event(e)
            with torch.cuda.stream(s1):
                gemm(A[:h], W, out[:h], bias, act, va)
            with torch.cuda.stream(s2):
                gemm(A[h:], W, out[h:], bias, act, vb)
            cur.wait_stream(s1); cur.wait_stream(s2)
        res[(split, va, vb)] = run(two)
    print(f"{name}: one launch {one:.1f} us (no epilogue stores: {one1:.1f}); two concurrent launches: " +
          ", ".join(f"{k}: {v:.1f}" for k, v in res.items()), flush=True)

# epilogue of a launch on 32 CUs only (M = 2048, N = 1280: 8 x 4 tiles of 256 x 320)
for name, M, N, K in [("o32", 2048, 1280, 1280), ("o252", 16000, 1280, 1280)]:
    A = torch.randn(M, K, device=DEV).to(BF16)
    W = (torch.randn(N, K, device=DEV) / K ** 0.5).to(BF16)
    out = torch.empty(M, N, device=DEV, dtype=BF16)
    t = [run(lambda: gemm(A, W, out, None, 0, "4", d)) for d in (0, 1, 2)]
    print(f"{name}: full {t[0]:.1f} us, no epilogue {t[1]:.1f}, one K tile {t[2]:.1f}", flush=True)
os.environ["TA355_GEMM_VARIANT"] = ""; os.environ["TA355_GEMM_DEBUG"] = "0"
