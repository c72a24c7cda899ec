#!/usr/bin/env python3
"""LM attention backward at the step's shape (B = 32, 16 q / 8 kv heads, L = 192, head_dim 128): the tiled kernels (+ the Delta pass)
against the one-workgroup-per-(clip, kv head) kernel of round 4; TA355_ATTN_BWD_GQA_DEBUG splits the latter's time."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops
from tests.test_gpu_kernels import rope_tables

DEV, BF16 = "cuda", torch.bfloat16
B, Hq, Hkv, hd, L = 32, 16, 8, 128, 192
NQKV = (Hq + 2 * Hkv) * hd
g = torch.Generator(device=DEV).manual_seed(0)
x0 = torch.randn(B * L, NQKV, device=DEV, generator=g).to(BF16)
qn, kn = 1 + 0.1 * torch.randn(hd, device=DEV, generator=g), 1 + 0.1 * torch.randn(hd, device=DEV, generator=g)
cos, sin = rope_tables(256, hd, 1e6)
scale = hd ** -0.5
O, lse, Q, K, V, rq, rk = ops.attention_fwd_qkv(x0, qn, kn, cos, sin, B, Hq, Hkv, L, scale)
dO = torch.randn(B * L, Hq * hd, device=DEV, generator=g).to(BF16)
flush = torch.empty(256 * 1024 * 1024, device=DEV, dtype=torch.int16)


def timed(fn, reps=20, cold=False):
    for _ in range(3):
        fn()
    tot = 0.0
    for _ in range(reps):
        if cold:
            flush.fill_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / reps * 1e3


def old():
    delta, _ = ops.attn_bwd_prep(dO, O, B, Hq, L)
    return ops.attention_bwd_qkv(Q, K, V, dO, lse, delta, x0, rq, rk, qn, kn, cos, sin, L, scale)


def new():
    return ops.attention_bwd_gqa(Q, K, V, dO, O, lse, x0, rq, rk, qn, kn, cos, sin, L, scale)


print(f"dbg={os.environ.get('TA355_ATTN_BWD_GQA_DEBUG', '0')}  tiled + Delta pass: warm {timed(old):6.1f} us  cold {timed(old, cold=True):6.1f} us   "
      f"one workgroup per (clip, kv head): warm {timed(new):6.1f} us  cold {timed(new, cold=True):6.1f} us")
