#!/usr/bin/env python3
"""Race hunt for the ping-pong GEMM's DMA schedule: the step's shapes, 30 launches each on fresh random operands, every result
compared bit-for-bit with the FIRST variant-independent reference (the 128x128 kernel, TA355_GEMM_VARIANT=0 semantics via ops)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops

SHAPES = [(16000, 3840, 1280), (16000, 1280, 1280), (16000, 5120, 1280), (16000, 1280, 5120), (6016, 4096, 1024), (6016, 6144, 1024),
          (6016, 3072, 1024), (6016, 2048, 1024), (1000, 640, 64), (300, 320, 128), (8192, 8192, 1024)]
bad = 0
for M, N, K in SHAPES:
    for rep in range(6):
        g = torch.Generator(device="cuda").manual_seed(rep * 7919 + M)
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
        ref = (A.float() @ W.float().t())
        outs = [ops.gemm_nt(A, W) for _ in range(5)]
        torch.cuda.synchronize()
        for o in outs[1:]:
            if not torch.equal(o, outs[0]):
                bad += 1; print("NON-DETERMINISTIC", M, N, K, rep, float((o.float() - outs[0].float()).abs().max()))
        err = float((outs[0].float() - ref).abs().max() / ref.abs().max())
        if err > 8e-3:
            bad += 1; print("WRONG", M, N, K, rep, err)
print("gemm_repeat_check: %d problems" % bad)
sys.exit(1 if bad else 0)
