#!/usr/bin/env python3
"""Run the encoder attention forward repeatedly on identical inputs (head-major and strided layouts) and count bitwise
differences between runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd import ops
DEV, BF16 = "cuda", torch.bfloat16
torch.manual_seed(0)
for S in (104, 100, 99, 500):
    B, nh, hd = 3, 5, 64
    M, H = B * S, nh * hd
    qk = torch.randn(M, 2 * H, device=DEV).to(BF16)
    v = torch.randn(M, H, device=DEV).to(BF16)
    vt = torch.zeros(H * M + 64, device=DEV, dtype=BF16); vt[:H * M] = v.T.contiguous().reshape(-1)
    lay = (S * 2 * H, 64, 2 * H, S * 2 * H, 64, 2 * H, S, 64 * M, M)
    q = qk[:, :H].reshape(B, S, nh, hd).transpose(1, 2).contiguous()
    k = qk[:, H:].reshape(B, S, nh, hd).transpose(1, 2).contiguous()
    Sp = ops.pad64(S)
    vT = torch.zeros(B, nh, hd, Sp, device=DEV, dtype=BF16); vT[..., :S] = v.reshape(B, S, nh, hd).permute(0, 2, 3, 1)
    a0 = ops.attention_fwd_strided(qk, qk[:, H:], vt, B, nh, nh, S, hd, False, 0.125, lay).clone()
    b0, _ = ops.attention_fwd(q, k, vT, S, False, 0.125, None, want_lse=False); b0 = b0.clone()
    da = db = dab = 0
    for i in range(50):
        a = ops.attention_fwd_strided(qk, qk[:, H:], vt, B, nh, nh, S, hd, False, 0.125, lay)
        b, _ = ops.attention_fwd(q, k, vT, S, False, 0.125, None, want_lse=False)
        da += int((a != a0).sum()); db += int((b != b0).sum()); dab += int((a != b).sum())
    print(f"S={S}: strided run-to-run diffs {da}, head-major run-to-run diffs {db}, strided-vs-head-major diffs {dab}", flush=True)
