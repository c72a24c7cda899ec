#!/usr/bin/env python3
"""A/B of the persistent ping-pong GEMM (v4: TA355_GEMM_PERSIST unset) against the one-tile-per-workgroup v2 (=0) on the
step's shapes; checks bit-identity.  usage: gemm_persist_ab.py [--reps 30] [--cold]"""
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


shapes = [  # name, M, N, K, act, residual(bf16, in place), f32 out
    ("enc_qk", 16000, 2560, 1280, 0, False, False), ("enc_vT", 1280, 16000, 1280, 0, False, False), ("enc_o", 16000, 1280, 1280, 0, True, False),
    ("enc_fc1", 16000, 5120, 1280, 1, False, False), ("enc_fc2", 16000, 1280, 5120, 0, True, False), ("conv2", 16000, 1280, 3840, 1, False, False),
    ("lm_qkv", 6144, 4096, 1024, 0, False, False), ("lm_gu", 6144, 6144, 1024, 0, False, False), ("lm_dact", 6144, 3072, 1024, 0, False, False),
    ("lm_dao", 6144, 2048, 1024, 0, False, False), ("ragged", 5000, 1000, 1280, 1, True, False), ("f32out", 4000, 2560, 1024, 0, False, True),
    ("sq4096", 4096, 4096, 4096, 0, False, False), ("big_fc1", 64000, 5120, 1280, 1, False, False),
]
print(f"{'shape':10s} {'M':>6s} {'N':>6s} {'K':>5s}   v2 us   v4 us   ratio   TF/s(v4)")
for name, M, N, K, act, hasres, f32 in shapes:
    torch.manual_seed(0)
    A = torch.randn(M, K, device=DEV).to(BF16)
    W = (torch.randn(N, K, device=DEV) / K ** 0.5).to(BF16)
    bias = torch.randn(N, device=DEV) if act else None
    res0 = torch.randn(M, N, device=DEV).to(BF16) if hasres else None
    ts, outs = [], []
    for persist in ("0", ""):
        os.environ["TA355_GEMM_PERSIST"] = persist
        for variant in (("4",) if name != "ragged" else ("4", "3")):
            os.environ["TA355_GEMM_VARIANT"] = variant
            out = res0.clone() if hasres else torch.full((M, N), float("nan"), device=DEV, dtype=torch.float32 if f32 else BF16)
            fn = lambda: ops.gemm_nt(A, W, M, N, K, out=out, bias=bias, residual_bf16=out if hasres else None, act=act)
            fn(); torch.cuda.synchronize(); outs.append(out.clone())
            if hasres:
                out.copy_(res0)
            ts.append(timeit(fn))
    half = len(ts) // 2
    same = all(torch.equal(outs[i], outs[i + half]) for i in range(half)) and not any(bool(torch.isnan(o.float()).any()) for o in outs)
    if not same:
        for i in range(half):
            d = (outs[i].float() - outs[i + half].float()).abs()
            bad = torch.nonzero(d > 0)
            print(f"   mismatch pair {i}: {bad.shape[0]} elements, rows {bad[:, 0].min().item()}..{bad[:, 0].max().item()}, cols {bad[:, 1].min().item()}..{bad[:, 1].max().item()}, max {d.max().item():.3g}"
                  if bad.numel() else f"   pair {i} identical")
    print(f"{name:10s} {M:6d} {N:6d} {K:5d} {ts[0]:7.1f} {ts[half]:7.1f} {ts[half] / ts[0]:7.3f} {2.0 * M * N * K / ts[half] / 1e6:8.0f}   identical={same}", flush=True)
os.environ["TA355_GEMM_VARIANT"] = ""; os.environ["TA355_GEMM_PERSIST"] = ""
