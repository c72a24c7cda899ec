#!/usr/bin/env python3
"""Phase stamps of the LM attention kernels inside the real step (experiment build, -DTA355_ATTN_STAMPS).

    python scripts/attn_stamps.py --build      # here (no GPU): tiny_audio_amd/libta355_stamps.so from the product objects + a stamped attention.o
    python scripts/attn_stamps.py [--streams f32] [--out gpurun_out/x.txt]      # on the GPU box

Runs two full-depth B = 32 training steps with the stamped library and reads the stamps of the LAST forward / backward attention
launch (layer 27 forward, layer 0 backward): s_memtime at the phase boundaries of wave 0 of every workgroup.  Prints mean / p90
durations of each phase in thousands of s_memtime ticks (= shader cycles on gfx950, MI355X_MICROARCH.md: ~2 k per microsecond at the
step's clocks; the launch span against the kernel's rocprof duration calibrates it).
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tiny_audio_amd", "libta355_stamps.so")


def build():
    from tiny_audio_amd import _lib
    _lib.build()
    objdir = os.path.join(_lib.CSRC, "build")
    obj = os.path.join(objdir, "attention_stamps.o")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fPIC", "-DTA355_ATTN_STAMPS"]
    subprocess.run([_lib.hipcc_path(), *flags, "-c", os.path.join(_lib.CSRC, "attention.hip"), "-o", obj], check=True)
    objs = [os.path.join(objdir, s[:-4] + ".o") for s in _lib.SOURCES if s != "attention.hip"] + [obj]
    subprocess.run([_lib.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB], check=True)
    print("built", LIB)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--streams", default="f32")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.build:
        return build()
    os.environ["TA355_LIB"] = LIB
    import numpy as np
    import torch
    from tiny_audio_amd import _lib, ops
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    from tiny_audio_amd.asr_processing import LogMelFeatureExtractor
    from tiny_audio_amd.synthetic import token_batch
    dev = torch.device("cuda:0")
    cfg = ASRConfig(model_dtype={"f32": "float32", "bf16": "bfloat16"}[a.streams], projector_hidden_dim=1024, audio_token_dropout=0.1)
    torch.manual_seed(0)
    model = ASRModel(cfg, device=dev, init="random", seed=0)
    model.train()
    fe = LogMelFeatureExtractor(128, dev)
    B, L, V = a.batch, 192, cfg.text_config.vocab_size
    wav = 0.1 * torch.randn(B, 160000, device=dev)
    lens = torch.full((B,), 160000, device=dev, dtype=torch.int64)
    ids, att, lab, counts, n_lab = token_batch(B, 125, V, cfg.audio_token_id, cfg.pad_token_id, cfg.eos_token_id, L=L)
    T = lambda x: torch.from_numpy(x).to(dev)
    for _ in range(2):
        feats, _m = fe.extract(wav, lens)
        out = model(input_ids=T(ids), input_features=feats, attention_mask=T(att), labels=T(lab), audio_token_counts=T(counts),
                    return_logits=False)
        out.loss.backward()
        torch.cuda.synchronize()
    handle = C.CDLL(LIB)                                         # same image: the symbol that is not in the header
    handle.ta_debug_attn_stamps.argtypes = [C.c_void_p, C.c_int]
    buf = np.zeros((8192, 16), dtype=np.uint64)
    assert handle.ta_debug_attn_stamps(buf.ctypes.data, 8192) == 0
    tick_us = 0.001      # report k-ticks
    lines = [f"attention phase stamps, B = {B}, L = {L}, streams {a.streams}, loss {float(out.loss):.4f} (wave 0 of every workgroup; thousands of shader cycles)"]

    def report(name, rows, phases):
        rows = rows[rows[:, 0] > 0]
        if not len(rows):
            lines.append(f"{name}: no stamps"); return
        lines.append(f"{name}: {len(rows)} workgroups (differences inside a workgroup only: s_memtime is not synchronised across XCDs)")
        for label, i0, i1 in phases:
            d = (rows[:, i1].astype(np.int64) - rows[:, i0].astype(np.int64)) * tick_us
            d = d[(rows[:, i0] > 0) & (rows[:, i1] > 0)]
            if not len(d):
                lines.append(f"    {label:<58s} (no stamps)"); continue
            lines.append(f"    {label:<58s} mean {d.mean():7.2f}  p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f}")
    fwd = buf[6144:6144 + B * 8]
    lines.append("raw fwd stamps of workgroup 0 (relative to its first): " + " ".join(str(int(x) - int(fwd[0, 0])) if x else "-" for x in fwd[0, :12]))
    report("attn_fwd_gqa_qkv_kernel (one workgroup per (clip, kv head))", fwd, [
        ("issue K/V DMA + rope rows + mask", 0, 1), ("wait for them (vmcnt 0 + barrier)", 1, 2), ("K norm + rope in LDS, K/V/rk written out", 2, 3),
        ("barrier", 3, 4), ("pass 0: stage + normalise 32 queries (wave 0)", 4, 5), ("pass 0: tile loop (1 tile)", 5, 6), ("pass 0: output stores issued", 6, 7),
        ("pass 1: stage + normalise 32 queries", 7, 8), ("pass 1: tile loop (3 tiles)", 8, 9), ("pass 1: output stores issued", 9, 10),
        ("stores acknowledged", 10, 11), ("whole workgroup (wave 0)", 0, 11)])
    dkv = buf[:3 * B * 8]
    for nit in sorted(set(int(x) for x in dkv[dkv[:, 0] > 0][:, 5])):
        sel = dkv[(dkv[:, 0] > 0) & (dkv[:, 5] == nit)]
        report(f"attn_bwd dK/dV body, {nit} (head, query tile) iterations", sel, [
            ("first Q / dO tile landed", 0, 1), ("iterations", 1, 2), ("epilogue: 2 x (stage + q|k|v post backward)", 2, 3),
            ("stores acknowledged", 3, 4), ("whole workgroup", 0, 4)])
    dq = buf[4096:4096 + 6 * B * 8]
    for nt in sorted(set(int(x) for x in dq[dq[:, 0] > 0][:, 5])):
        sel = dq[(dq[:, 0] > 0) & (dq[:, 5] == nt)]
        report(f"attn_bwd dQ body, {nt} key tiles", sel, [
            ("first K / V tile landed (+ q, dO fragments)", 0, 1), ("tiles", 1, 2), ("epilogue: stage + q post backward", 2, 3),
            ("stores acknowledged", 3, 4), ("whole workgroup", 0, 4)])
    text = "\n".join(lines)
    print(text)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
