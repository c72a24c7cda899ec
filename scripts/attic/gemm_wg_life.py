#!/usr/bin/env python3
"""Experiment: where a workgroup of the 256x320 ping-pong GEMM spends its life, and how long a CU sits between two workgroups.
TA355_GEMM_VARIANT=8 + TA355_GEMM_DEBUG=4: every workgroup stamps s_memtime at entry, first K tile landed, loop end, stores
issued, stores acknowledged, plus HW_ID / XCC_ID, behind the C matrix.  usage: gemm_wg_life.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tiny_audio_amd import ops

DEV, BF16 = "cuda", torch.bfloat16
shapes = [("enc_o", 16000, 1280, 1280), ("enc_qkv", 16000, 3840, 1280), ("enc_fc1", 16000, 5120, 1280), ("lm_gu", 6144, 6144, 1024), ("big_fc1", 64000, 5120, 1280)]
if "--few" in sys.argv:     # how the epilogue scales with the number of CUs storing at once: 1, 8 (one per XCD), 64, 128, 252 tiles
    shapes = [("t1", 256, 320, 1280), ("t8", 256, 2560, 1280), ("t32", 1024, 2560, 1280), ("t64", 2048, 2560, 1280), ("t128", 4096, 2560, 1280), ("t252", 16000, 1280, 1280)]
PERSIST = "--v4" in sys.argv            # variant 9: the persistent kernel (stamps: loop top, first K tile landed, loop end, next tile's DMA issued, stores issued)
os.environ["TA355_GEMM_VARIANT"] = "9" if PERSIST else "8"
os.environ["TA355_GEMM_DEBUG"] = "0" if PERSIST else "4"
for name, M, N, K in shapes:
    A = torch.randn(M, K, device=DEV).to(BF16)
    W = (torch.randn(N, K, device=DEV) / K ** 0.5).to(BF16)
    ntile = ((M + 255) // 256) * ((N + 319) // 320)
    out = torch.zeros(M * N + ntile * 2 * 8 * 4, device=DEV, dtype=BF16)
    for _ in range(3):
        ops.gemm_nt(A, W, M, N, K, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.gemm_nt(A, W, M, N, K, out=out); b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3
    raw = out[M * N:].view(torch.int64).cpu().numpy().reshape(ntile, 2, 8)
    g0 = raw[:, 0, :]
    st = g0[:, :5].astype(np.int64)
    cu = (g0[:, 6] & 15) * 65536 + ((g0[:, 5] >> 8) & 0xFFFF)          # (xcc, se | sh | cu)
    life = st[:, 4] - st[:, 0]
    clk = None
    print(f"{name}: {M}x{N}x{K}, {ntile} tiles, launch {us:.1f} us; distinct CUs seen {len(set(cu.tolist()))}")
    ph = np.stack([st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 4] - st[:, 3]], 1)
    fmt = tuple(f"{np.median(ph[:, i]):.0f}/{np.percentile(ph[:, i], 10):.0f}/{np.percentile(ph[:, i], 90):.0f}" for i in range(4))
    if PERSIST:
        print("   cycles (median / p10 / p90): wait for the first K tile %s | main loop %s | next tile's set-up + DMA issue %s | epilogue %s" % fmt)
        if ntile > 256:
            first = st[:256]; later = st[256:]
            print(f"   first tiles: wait {np.median(first[:, 1] - first[:, 0]):.0f}; later tiles: wait {np.median(later[:, 1] - later[:, 0]):.0f}; "
                  f"tile period of a workgroup (top -> next top) {np.median(st[256:512, 0] - st[:256, 0][:len(st[256:512])]):.0f} cycles")
        continue
    print("   cycles (median / p10 / p90): prologue [entry -> first K tile landed] %s | main loop %s | epilogue issue %s | store drain %s" % fmt)
    gaps, firsts = [], []
    for c in set(cu.tolist()):
        rows = st[cu == c]
        rows = rows[np.argsort(rows[:, 0])]
        firsts.append(rows[0, 0])
        for i in range(1, len(rows)):
            gaps.append(rows[i, 0] - rows[i - 1, 4])
    if gaps:
        gaps = np.array(gaps)
        print(f"   CU turn-around [stores acknowledged -> next workgroup's entry]: median {np.median(gaps):.0f}, p10 {np.percentile(gaps, 10):.0f}, p90 {np.percentile(gaps, 90):.0f} cycles ({len(gaps)} hand-overs)")
    # per-XCC span (memtime is per XCC): first entry -> last ack
    for x in sorted(set((g0[:, 6] & 15).tolist()))[:2]:
        m = (g0[:, 6] & 15) == x
        span = st[m, 4].max() - st[m, 0].min()
        print(f"   xcc {x}: {m.sum()} workgroups, span {span} cycles = {span / us:.0f} cycles/us of the launch; round ends (ack) spread p10-p90 of last round: "
              f"{np.percentile(st[m, 4], 90) - np.percentile(st[m, 4], 10):.0f}")
os.environ["TA355_GEMM_VARIANT"] = ""; os.environ["TA355_GEMM_DEBUG"] = "0"
