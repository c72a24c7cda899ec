#!/usr/bin/env python3
"""Where the cycles of the ping-pong GEMM go: the TIMING build (TA355_GEMM_VARIANT=8) stamps s_memtime around the load
interval, the two barriers and the compute interval of every K half-step; per-wave sums are written into the output buffer."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tiny_audio_amd import ops

DEV, BF16 = "cuda", torch.bfloat16
for name, M, N, K in [("enc_o", 16000, 1280, 1280), ("enc_fc2", 16000, 1280, 5120), ("enc_fc1", 16000, 5120, 1280), ("sq4096", 4096, 4096, 4096)]:
    A = torch.randn(M, K, device=DEV).to(BF16)
    W = (torch.randn(N, K, device=DEV) / K ** 0.5).to(BF16)
    out = torch.zeros(M, N, device=DEV, dtype=BF16)
    os.environ["TA355_GEMM_VARIANT"] = "8"
    for _ in range(3):
        ops.gemm_nt(A, W, M, N, K, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.gemm_nt(A, W, M, N, K, out=out); b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3
    tiles = ((M + 255) // 256) * ((N + 319) // 320)
    raw = out.view(torch.int16).cpu().numpy().reshape(-1).view(np.uint64)[: tiles * 8 * 8].reshape(tiles, 8, 8).astype(np.float64)
    nk = raw[0, 0, 5]
    per = raw[:, :, :5] / (2 * nk)                       # ticks per half-step
    g0, g1 = per[:, :4].mean((0, 1)), per[:, 4:].mean((0, 1))
    print(f"{name}: kernel {us:.1f} us, {int(nk)} K tiles; per half-step [L, wait-b1, C, wait-b2 | loop/2nk] ticks: "
          f"group0 {np.round(g0, 1)}  group1 {np.round(g1, 1)};  loop ticks total {raw[:, :, 4].mean():.0f} = {raw[:, :, 4].mean() / us:.1f} ticks/us")
os.environ["TA355_GEMM_VARIANT"] = ""
