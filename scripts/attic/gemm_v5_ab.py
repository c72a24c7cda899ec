#!/usr/bin/env python3
"""A/B of the one-192x128-tile-per-CU GEMM (v5, TA355_GEMM_VARIANT=10) against the automatic choice's candidates on the
step's shapes; checks bit-identity.  usage: gemm_v5_ab.py [--reps 30] [--cold]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops

DEV, BF16 = "cuda", torch.bfloat16
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 30
cold = "--cold" in sys.argv
_flush = torch.empty(256 * 1024 * 1024, device=DEV, dtype=torch.int32) if cold else None


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if cold:
        tot, n = 0.0, max(reps // 2, 5)
        for _ in range(n):
            _flush.fill_(1)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            tot += a.elapsed_time(b)
        return tot / n * 1e3
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


shapes = [  # name, M, N, K, act, residual(bf16, in place)
    ("lm_o", 6144, 1024, 2048, 0, True), ("lm_down", 6144, 1024, 3072, 0, True), ("lm_dxn_qkv", 6144, 1024, 4096, 0, False),
    ("lm_dxn_gu", 6144, 1024, 6144, 0, False), ("lm_dao", 6144, 2048, 1024, 0, False), ("lm_dact", 6144, 3072, 1024, 0, False),
    ("lm_qkv", 6144, 4096, 1024, 0, False), ("lm_gu", 6144, 6144, 1024, 0, False), ("enc_o", 16000, 1280, 1280, 0, True),
    ("enc_fc1", 16000, 5120, 1280, 1, False), ("proj_fc1", 4000, 1024, 5120, 1, False), ("proj_fc2", 4000, 1024, 1024, 0, False),
    ("ragged", 5000, 1000, 1280, 1, True), ("lm17_o", 6144, 2048, 2048, 0, True), ("lm17_down", 6144, 2048, 6144, 0, True),
]
variants = ["0", "5", "3", "4", "10", "11"]
print(f"{'shape':11s} {'M':>6s} {'N':>6s} {'K':>5s} " + " ".join(f"{'v' + v:>8s}" for v in variants) + "   us per launch" + (" (cold operands)" if cold else ""))
for name, M, N, K, act, hasres in shapes:
    torch.manual_seed(0)
    A = torch.randn(M, K, device=DEV).to(BF16)
    W = (torch.randn(N, K, device=DEV) / K ** 0.5).to(BF16)
    bias = torch.randn(N, device=DEV) if act else None
    res0 = torch.randn(M, N, device=DEV).to(BF16) if hasres else None
    ts, outs = [], []
    for v in variants:
        os.environ["TA355_GEMM_VARIANT"] = v
        out = res0.clone() if hasres else torch.full((M, N), float("nan"), device=DEV, dtype=BF16)
        fn = lambda: ops.gemm_nt(A, W, M, N, K, out=out, bias=bias, residual_bf16=out if hasres else None, act=act)
        fn(); torch.cuda.synchronize(); outs.append(out.clone())
        if hasres:
            out.copy_(res0)
        ts.append(timeit(fn))
    ok = not bool(torch.isnan(outs[-1].float()).any())
    same = torch.equal(outs[-2], outs[1])
    md = float((outs[-1].float() - outs[1].float()).abs().max())
    best = min(range(len(ts)), key=lambda i: ts[i])
    print(f"{name:11s} {M:6d} {N:6d} {K:5d} " + " ".join(f"{t:8.1f}" for t in ts) + f"   best v{variants[best]}  v10 vs v5 identical={same}; v11 vs v5 maxdiff={md:.3g} finite={ok}", flush=True)
os.environ["TA355_GEMM_VARIANT"] = ""
