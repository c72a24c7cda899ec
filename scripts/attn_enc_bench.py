#!/usr/bin/env python3
"""Encoder attention alone at the step's shape (B = 32, 20 heads x 64, S = 500): ta_attention_enc_fwd (round 3) against the
round-2 kernel (ta_attention_fwd over head-major Q / K and a V^T image); also the target of the PMC passes (scripts/gpu_pmc.sh
with PMC_CMD).   usage: attn_enc_bench.py [--reps N] [--only new|old]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops

reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else "both"
B, nh, S, hd = 32, 20, 500, 64
H = nh * hd
dev = "cuda"
qkv = torch.randn(B * S, 3 * H, device=dev).to(torch.bfloat16)


def timeit(fn):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


flops = 4.0 * S * S * hd * B * nh
if only in ("both", "new"):
    us = timeit(lambda: ops.attention_enc_fwd(qkv, B, nh, S))
    print(f"ta_attention_enc_fwd  {us:7.1f} us  {flops / us / 1e6:6.0f} TF/s")
if only in ("both", "old"):
    q, k, v = (qkv[:, i * H:(i + 1) * H].reshape(B, S, nh, hd).transpose(1, 2).contiguous() for i in range(3))
    Sp = ops.pad64(S)
    vt = torch.zeros(B, nh, hd, Sp, device=dev, dtype=torch.bfloat16); vt[..., :S] = v.transpose(-1, -2)
    us = timeit(lambda: ops.attention_fwd(q, k, vt, S, False, 0.125, None, want_lse=False))
    print(f"ta_attention_fwd (r2) {us:7.1f} us  {flops / us / 1e6:6.0f} TF/s")
