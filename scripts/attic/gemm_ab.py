#!/usr/bin/env python3
"""Same-process A/B of GEMM tile variants on the step's shapes (B = 32), with the experiment modes of csrc/gemm.hip:
TA355_GEMM_DEBUG=1 (no epilogue stores: prologue + main loop), =2 (one K tile: prologue + epilogue), =0 (all).
usage: gemm_ab.py [--variants 4,7] [--reps 20] [--match name] [--cold]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops

DEV, BF16, F32 = "cuda", torch.bfloat16, torch.float32
arg = lambda k, d: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
reps = int(arg("--reps", "20"))
variants = arg("--variants", "4,7").split(",")
match = arg("--match", "")
modes = [int(x) for x in arg("--modes", "0,1,2").split(",")]
cold = "--cold" in sys.argv
_flush = torch.empty(256 * 1024 * 1024, device=DEV, dtype=torch.int32) if cold else None


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if cold:
        tot = 0.0
        for _ in range(max(reps // 2, 5)):
            _flush.fill_(1)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            tot += a.elapsed_time(b)
        return tot / max(reps // 2, 5) * 1e-3
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


shapes = [  # name, M, N, K, act, residual(bf16, in place)
    ("enc_qk", 16000, 2560, 1280, 0, False), ("enc_vT", 1280, 16000, 1280, 0, False), ("enc_o", 16000, 1280, 1280, 0, True),
    ("enc_fc1", 16000, 5120, 1280, 1, False), ("enc_fc2", 16000, 1280, 5120, 0, True), ("conv2", 16000, 1280, 3840, 1, False),
    ("lm_qkv", 6144, 4096, 1024, 0, False), ("lm_o", 6144, 1024, 2048, 0, True), ("lm_gu", 6144, 6144, 1024, 0, False),
    ("lm_down", 6144, 1024, 3072, 0, True), ("lm_dact", 6144, 3072, 1024, 0, False), ("lm_dao", 6144, 2048, 1024, 0, False),
    ("lm_dxn_qkv", 6144, 1024, 4096, 0, False), ("lm_dxn_gu", 6144, 1024, 6144, 0, False), ("sq4096", 4096, 4096, 4096, 0, False),
]
print(f"{'shape':12s} {'M':>6s} {'N':>6s} {'K':>5s} " + " ".join(f"{'v' + v + ':' + str(m):>10s}" for v in variants for m in modes) + "   (us; TF/s of mode 0)")
for name, M, N, K, act, hasres in shapes:
    if match and match not in name:
        continue
    torch.manual_seed(0)
    A = torch.randn(M, K, device=DEV).to(BF16)
    W = (torch.randn(N, K, device=DEV) / K ** 0.5).to(BF16)
    bias = torch.randn(N, device=DEV) if act else None
    res0 = torch.randn(M, N, device=DEV).to(BF16) if hasres else None
    cells, outs = [], {}
    for v in variants:
        os.environ["TA355_GEMM_VARIANT"] = v
        for mode in modes:
            os.environ["TA355_GEMM_DEBUG"] = str(mode)
            out = res0.clone() if hasres else torch.empty(M, N, device=DEV, dtype=BF16)
            fn = lambda: ops.gemm_nt(A, W, M, N, K, out=out, bias=bias, residual_bf16=out if hasres else None, act=act)
            if mode == 0:
                fn(); torch.cuda.synchronize(); outs[v] = out.clone()
                if hasres:
                    out.copy_(res0)
            t = timeit(fn)
            cells.append(f"{t * 1e6:10.1f}")
            if mode == 0:
                cells[-1] = f"{t * 1e6:6.1f}/{2.0 * M * N * K / t / 1e12:4.0f}"
    os.environ["TA355_GEMM_DEBUG"] = "0"
    ref = outs[variants[0]]
    same = all(torch.equal(ref, o) for o in outs.values())
    md = max(float((ref.float() - o.float()).abs().max()) for o in outs.values())
    print(f"{name:12s} {M:6d} {N:6d} {K:5d} " + " ".join(f"{c:>10s}" for c in cells) + f"   identical={same} maxdiff={md:.3g}", flush=True)
os.environ["TA355_GEMM_VARIANT"] = ""
