#!/usr/bin/env python3
"""Does the Infinity Cache (256 MB, memory side) serve a read of what the PREVIOUS kernel wrote?  For buffer sizes 16 MB ... 512 MB:
read rate of `dst.copy_(src)`-style streaming reads (a) right after another kernel WROTE the buffer, (b) right after another
kernel READ it, (c) cold (1 GB written in between).  Uses torch kernels only (float4 streaming copies / fills)."""
import torch
dev = "cuda"
flush = torch.empty(1 << 30, device=dev, dtype=torch.uint8)


def timed(fn, prep, reps=8):
    ts = []
    for _ in range(reps):
        prep()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


print(f"{'MB':>6s} {'read after WRITE':>18s} {'read after READ':>18s} {'read COLD':>14s}   (GB/s of the read of the buffer: reduction kernel)")
for mb in (16, 32, 64, 96, 128, 164, 200, 256, 384, 512):
    n = mb * (1 << 20) // 4
    buf = torch.empty(n, device=dev, dtype=torch.float32)
    src = torch.randn(n, device=dev, dtype=torch.float32)
    rd = lambda: buf.sum()                                   # streaming read of buf
    w = timed(rd, lambda: buf.copy_(src))                    # the previous kernel wrote buf (and read src)
    r = timed(rd, lambda: (flush.fill_(1), buf.sum()))       # the previous kernel read buf
    c = timed(rd, lambda: (buf.sum(), flush.fill_(1)))       # 1 GB written in between
    f = lambda ms: f"{mb * 1.048576 / ms:10.0f}"
    print(f"{mb:6d} {f(w):>18s} {f(r):>18s} {f(c):>14s}")
    del buf, src
