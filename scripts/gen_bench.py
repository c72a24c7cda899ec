#!/usr/bin/env python3
"""Greedy-generation timing at full model size (secondary metric: the WER loop's inference side).
B clips of 10 s, prompt = 3 + 125 <audio> + 24 tokens, N new tokens; reports prompt-pass and per-token latency."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tiny_audio_amd.asr_config import ASRConfig
from tiny_audio_amd.asr_modeling import ASRModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = ASRConfig()
m = ASRModel(cfg, device="cuda", init="random", seed=0)
feats = torch.randn(B, 128, 1000, device="cuda") * 0.5
amask = torch.ones(B, 1000, dtype=torch.int64)
ids = torch.tensor([[5, 6, 7] + [cfg.audio_token_id] * 125 + list(range(100, 124))] * B)
kw = dict(input_ids=ids, input_features=feats, audio_attention_mask=amask, attention_mask=torch.ones_like(ids), eos_token_id=[])
res = {}
for n in (1, N):
    m.generate(**kw, max_new_tokens=n); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = m.generate(**kw, max_new_tokens=n)
    torch.cuda.synchronize()
    res[n] = (time.perf_counter() - t0) / 3
per_tok = (res[N] - res[1]) / (N - 1)
# decode roofline: a step streams every LM weight once (bf16 linears of all layers + the tied lm_head) and the KV cache of the
# positions so far (prompt 152 + on average N / 2 generated); HBM peak 8 TB/s (MI355X_MICROARCH.md)
t = cfg.text_config
D, F, nl = t.hidden_size, t.intermediate_size, t.num_hidden_layers
nq, nkv, hd = t.num_attention_heads, t.num_key_value_heads, t.head_dim
vocab_pad = (t.vocab_size + 127) // 128 * 128
w_bytes = 2 * (nl * (D * (nq + 2 * nkv) * hd + nq * hd * D + 3 * D * F) + vocab_pad * D)
kv_bytes = 2 * 2 * nl * B * nkv * hd * (ids.shape[1] + N // 2)
floor_ms = (w_bytes + kv_bytes) / 8e12 * 1e3
print(json.dumps({"B": B, "new_tokens": N, "prompt_pass_ms": round(res[1] * 1e3, 2), "per_token_ms": round(per_tok * 1e3, 3),
                  "tokens_per_s": round(B / per_tok, 1), "total_ms": round(res[N] * 1e3, 1),
                  "rtf_audio_s_per_s": round(B * 10.0 / res[N], 1),
                  "roofline": {"bound": "hbm", "weight_bytes": w_bytes, "kv_cache_bytes": kv_bytes, "peak": 8000.0, "unit": "GB/s",
                               "achieved": round((w_bytes + kv_bytes) / per_tok / 1e9, 1),
                               "frac": round(floor_ms / (per_tok * 1e3), 4), "floor_ms_per_token": round(floor_ms, 4),
                               "frac_weights_only": round(w_bytes / 8e12 / per_tok, 4)},
                  "fused": os.environ.get("TA355_DECODE_FUSED", "1") != "0"}))
