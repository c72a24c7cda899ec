#!/usr/bin/env python3
"""What one GPU can tell about the N > 1 default: overlapped (deferred update) vs synchronous all-reduce.

RCCL itself cannot run on a one-GPU box, but what its kernels DO to the step can be emulated: a ring all-reduce is a handful of
channel workgroups (no LDS to speak of, copy traffic) resident for as long as the collective takes.  `ta_debug_occupy` is such a kernel.
The step's GEMMs are persistent one-workgroup-per-CU kernels that leave no room for a second resident workgroup (147 KB of LDS, the
whole register file), so a side-stream kernel is admitted only in the gap between two launches and then holds its CUs -- the next
GEMM's round runs on fewer CUs (round 4 measured +1.0 ms per LoRA step for adapter-gradient kernels on a side stream).  This script
measures the same trade for the collective:

  none         no collective at all (the one-GPU step)
  sync         the footprint on the COMPUTE stream between backward and update (exposed in full, disturbs nothing)
  overlapped   the footprint on a SIDE stream, launched after the backward, waited for after the next step's frozen-encoder forward
               (trainer.py's deferred update: hidden, but it shares the chip with log-mel + 32 encoder layers)

for a collective of `--ms` milliseconds on `--wgs` workgroups.  usage: allreduce_footprint.py [--ms 0.3] [--wgs 32] [--full-ft]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tiny_audio_amd import _lib, ops, trainer as TR
from tiny_audio_amd.asr_config import ASRConfig
from tiny_audio_amd.asr_modeling import ASRModel
from tiny_audio_amd.asr_processing import LogMelFeatureExtractor
from tiny_audio_amd.synthetic import token_batch

ap = argparse.ArgumentParser()
ap.add_argument("--ms", type=float, nargs="+", default=[0.3])
ap.add_argument("--wgs", type=int, nargs="+", default=[32])
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--full-ft", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = ASRConfig(audio_token_dropout=0.10, freeze_language_model=not a.full_ft, projector_hidden_dim=2048 if a.full_ft else 1024)
torch.manual_seed(0)
model = ASRModel(cfg, device=dev, init="random", seed=0)
model.train()
fe = LogMelFeatureExtractor(128, dev)
B, L, V = 32, 192, cfg.text_config.vocab_size
g = torch.Generator(device=dev); g.manual_seed(1234)
wav = 0.1 * torch.randn(B, 160000, device=dev, generator=g)
lens = torch.full((B,), 160000, device=dev, dtype=torch.int64)
ids, att, lab, counts, n_lab = token_batch(B, 125, V, cfg.audio_token_id, cfg.pad_token_id, cfg.eos_token_id, L=L)
ids_d, att_d, lab_d, counts_d = (torch.from_numpy(x).to(dev) for x in (ids, att, lab, counts))
side = torch.cuda.Stream()
buf = torch.zeros(64 * 1024 * 1024, device=dev, dtype=torch.uint8)          # what the collective streams
state = {"mode": "none", "ms": 0.3, "wgs": 32}


class _Work:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def fake_allreduce(flat, group=None, async_op=False):
    """stands in for trainer.allreduce_flat: the collective's FOOTPRINT (no data is reduced: one rank)"""
    if state["mode"] == "none":
        return None if async_op else flat
    L_ = _lib.lib()
    if not async_op:
        _lib.check(L_.ta_debug_occupy(state["wgs"], state["ms"] * 1e3, buf.data_ptr(), buf.numel(), ops.stream()), "ta_debug_occupy")
        return flat
    cur = torch.cuda.current_stream()
    ready = torch.cuda.Event(); ready.record(cur)
    side.wait_event(ready)
    with torch.cuda.stream(side):
        _lib.check(L_.ta_debug_occupy(state["wgs"], state["ms"] * 1e3, buf.data_ptr(), buf.numel(), side.cuda_stream), "ta_debug_occupy")
        done = torch.cuda.Event(); done.record(side)
    return _Work(done)


TR.allreduce_flat = fake_allreduce


def run(mode, ms, wgs, steps):
    state.update(mode=mode, ms=ms, wgs=wgs)
    tr = TR.ASRTrainer(model, TR.TrainingArguments(learning_rate=1e-4, max_grad_norm=1.0), overlap_allreduce=(mode == "overlapped"),
                       decoder_learning_rate=1e-5 if a.full_ft else None)

    def step():
        feats, _ = fe.extract(wav, lens)
        rows, tg, _n = ops.label_rows(lab_d)
        tr.training_step(dict(input_ids=ids_d, input_features=feats, attention_mask=att_d, labels=lab_d, audio_token_counts=counts_d,
                              label_meta=(rows, tg, n_lab)), return_logits=False)
    for _ in range(3):
        step()
    tr.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    tr.flush(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


import gc
res = {}
run("none", 0.3, 32, 4); gc.collect(); gc.freeze()
for rep in range(2):
    for ms in a.ms:
        for wgs in a.wgs:
            for mode in ("none", "sync", "overlapped"):
                t = run(mode, ms, wgs, a.steps)
                res.setdefault(f"{ms} ms x {wgs} wgs", {}).setdefault(mode, []).append(round(t, 3))
print(("full-FT (2.4 GB of gradients)" if a.full_ft else "MLP projector (25 MB of gradients)") + ", B = 32, ms per step, two passes:")
for k, v in res.items():
    base = sum(v["none"]) / len(v["none"])
    print(f"  collective {k}: " + "  ".join(f"{m} {v[m]} (+{sum(v[m]) / len(v[m]) - base:.2f})" for m in ("none", "sync", "overlapped")))
print(json.dumps(res))
