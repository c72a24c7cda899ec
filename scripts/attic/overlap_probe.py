#!/usr/bin/env python3
"""Probe: encoder forward on one stream (B=32) vs two half-batches on two streams (do HBM-bound kernels of one half
hide under the MFMA-bound GEMMs of the other?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tiny_audio_amd.asr_config import ASRConfig
from tiny_audio_amd.encoder import GlmAsrEncoderMI355X

cfg = ASRConfig()
enc = GlmAsrEncoderMI355X(cfg.audio_config, device="cuda").random_init(0)
B = 32
x = torch.randn(B, 128, 1000, device="cuda") * 0.5
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def full():
    return enc(x).last_hidden_state

def split(n=2):
    outs = []
    cur = torch.cuda.current_stream()
    streams = [s1, s2][:n]
    for s in streams: s.wait_stream(cur)
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            outs.append(enc(x[i * B // n:(i + 1) * B // n]).last_hidden_state)
    for s in streams: cur.wait_stream(s)
    return outs

def seq_halves():
    return [enc(x[:16]).last_hidden_state, enc(x[16:]).last_hidden_state]

for name, fn in (("full B=32, 1 stream", full), ("2 x B=16, 2 streams", split), ("2 x B=16, 1 stream", seq_halves), ("full B=32, 1 stream", full), ("2 x B=16, 2 streams", split)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
    print(f"{name:24s}: {t * 1e3:7.2f} ms", flush=True)
