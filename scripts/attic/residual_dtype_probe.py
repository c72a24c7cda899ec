#!/usr/bin/env python3
"""Full-depth effect of the bf16 residual streams: run once per mode (the env toggles are read at first use) and compare.
usage: residual_dtype_probe.py out.pt     (run with / without TA355_ENC_RES_F32=1 TA355_LM_RES_F32=1), then --compare a.pt b.pt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == "--compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    cos = lambda x, y: float((x.flatten().double() @ y.flatten().double()) / (x.double().norm() * y.double().norm()))
    print("encoder output cosine %.6f  rel-to-max %.4f" % (cos(a["enc"], b["enc"]), float((a["enc"] - b["enc"]).abs().max() / b["enc"].abs().max())))
    print("loss %.5f vs %.5f" % (a["loss"], b["loss"]))
    for k in a["grads"]:
        print("grad cosine %-22s %.6f" % (k, cos(a["grads"][k], b["grads"][k])))
    sys.exit(0)
import numpy as np
from oracle import weights as OW
from tiny_audio_amd.asr_config import ASRConfig
from tiny_audio_amd.asr_modeling import ASRModel
torch.manual_seed(0)                      # the projector init uses the global RNG
cfg = ASRConfig(audio_token_dropout=0.0)
m = ASRModel(cfg, device="cuda", init="random", seed=0)
B, L = 4, 192
g = torch.Generator(device="cuda"); g.manual_seed(5)
feats = torch.randn(B, 128, 1000, device="cuda", generator=g) * 0.5
ids, att, lab, counts = OW.synthetic_tokens(B, 125, cfg.text_config.vocab_size, cfg.audio_token_id, cfg.pad_token_id, cfg.eos_token_id, L=L)
enc = m.audio_tower(feats, return_f32=True).last_hidden_state
m.train()
out = m(input_features=feats, input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(att), labels=torch.from_numpy(lab),
        audio_token_counts=torch.from_numpy(counts), return_logits=False)
out.loss.backward()
torch.save({"enc": enc.float().cpu(), "loss": float(out.loss), "grads": {k: p.grad.float().cpu() for k, p in m.projector.named_parameters()}}, sys.argv[1])
print("saved", sys.argv[1], float(out.loss))
