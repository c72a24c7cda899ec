#!/usr/bin/env python3
"""Which hipBLASLt kernels (macro tile, schedule) does torch.matmul pick on the step's GEMM shapes?  Run under
`rocprofv3 --kernel-trace --stats`: shape i is launched 10 + i times, so the call counts of the stats table identify it.

    rocprofv3 --kernel-trace --stats -d out -- python scripts/blaslt_kernel_names.py
"""
import torch

SHAPES = [("enc q|k|v", 16000, 3840, 1280), ("enc o_proj", 16000, 1280, 1280), ("enc fc1", 16000, 5120, 1280),
          ("enc fc2", 16000, 1280, 5120), ("lm q|k|v", 6016, 4096, 1024), ("lm o", 6016, 1024, 2048),
          ("lm gate|up", 6016, 6144, 1024), ("lm down", 6016, 1024, 3072), ("lm d(act)", 6016, 3072, 1024),
          ("lm d(xn) gu", 6016, 1024, 6144), ("lm d(ao)", 6016, 2048, 1024), ("lm d(xn) qkv", 6016, 1024, 4096),
          ("sq 8192", 8192, 8192, 8192)]

for i, (name, M, N, K) in enumerate(SHAPES):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(10 + i):
        torch.matmul(A, W.t(), out=out)
    torch.cuda.synchronize()
    print(f"{10 + i} launches: {name} M={M} N={N} K={K}")
