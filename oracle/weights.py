"""Seeded synthetic weights in the reference's parameter naming.

Pretrained checkpoints are not reachable from the build container or the GPU
box, so parity is pinned on random weights at the true (or reduced) shapes.
The recipe is pure numpy so that ``tests/golden/make_golden.py`` (which loads
these arrays into the reference's torch modules), the oracle and the HIP path
all see bit-identical float32 parameters.

Names follow ``GlmAsrEncoder`` (TF:models/glmasr/modeling_glmasr.py:286-311),
``Qwen3ForCausalLM`` (TF:models/qwen3/modeling_qwen3.py:211-324,432-446) and the
projector modules (tiny_audio/projectors.py:23-50, 185-251).
"""
from __future__ import annotations

import numpy as np


def enc_config(hidden=1280, ffn=5120, layers=32, heads=20, n_mels=128,
               rope_theta=10000.0, partial_rotary=0.5, ln_eps=1e-5):
    """GlmAsrEncoderConfig defaults, TF:models/glmasr/configuration_glmasr.py:44-61."""
    assert hidden % heads == 0
    return dict(hidden=hidden, ffn=ffn, layers=layers, heads=heads, n_mels=n_mels,
                head_dim=hidden // heads, rope_theta=rope_theta,
                partial_rotary=partial_rotary, ln_eps=ln_eps)


def lm_config(vocab=151670, hidden=1024, ffn=3072, layers=28, heads=16, kv_heads=8,
              head_dim=128, rms_eps=1e-6, rope_theta=1e6):
    """Qwen3-0.6B shape (public hub config; SURVEY.md section 8 preamble)."""
    return dict(vocab=vocab, hidden=hidden, ffn=ffn, layers=layers, heads=heads,
                kv_heads=kv_heads, head_dim=head_dim, rms_eps=rms_eps, rope_theta=rope_theta)


def _lin(rng, out_f, in_f, gain=1.0):
    return (rng.standard_normal((out_f, in_f)) * (gain / np.sqrt(in_f))).astype(np.float32)


def _vec(rng, n, mean=0.0, std=0.02):
    return (mean + std * rng.standard_normal(n)).astype(np.float32)


def init_encoder(cfg, seed=0):
    rng = np.random.RandomState(seed)
    H, F, M = cfg["hidden"], cfg["ffn"], cfg["n_mels"]
    w = {}
    w["conv1.weight"] = (rng.standard_normal((H, M, 3)) / np.sqrt(3 * M)).astype(np.float32)
    w["conv1.bias"] = _vec(rng, H)
    w["conv2.weight"] = (rng.standard_normal((H, H, 3)) / np.sqrt(3 * H)).astype(np.float32)
    w["conv2.bias"] = _vec(rng, H)
    for i in range(cfg["layers"]):
        p = f"layers.{i}."
        w[p + "input_layernorm.weight"] = _vec(rng, H, 1.0, 0.1)
        w[p + "input_layernorm.bias"] = _vec(rng, H)
        w[p + "self_attn.q_proj.weight"] = _lin(rng, H, H)
        w[p + "self_attn.q_proj.bias"] = _vec(rng, H)
        w[p + "self_attn.k_proj.weight"] = _lin(rng, H, H)      # no bias (modeling_glmasr.py:184)
        w[p + "self_attn.v_proj.weight"] = _lin(rng, H, H)
        w[p + "self_attn.v_proj.bias"] = _vec(rng, H)
        w[p + "self_attn.o_proj.weight"] = _lin(rng, H, H, 0.5)
        w[p + "self_attn.o_proj.bias"] = _vec(rng, H)
        w[p + "post_attention_layernorm.weight"] = _vec(rng, H, 1.0, 0.1)
        w[p + "post_attention_layernorm.bias"] = _vec(rng, H)
        w[p + "mlp.fc1.weight"] = _lin(rng, F, H)
        w[p + "mlp.fc1.bias"] = _vec(rng, F)
        w[p + "mlp.fc2.weight"] = _lin(rng, H, F, 0.5)
        w[p + "mlp.fc2.bias"] = _vec(rng, H)
    w["norm.weight"] = _vec(rng, H, 1.0, 0.1)
    w["norm.bias"] = _vec(rng, H)
    return w


def init_lm(cfg, seed=1):
    rng = np.random.RandomState(seed)
    D, F, V = cfg["hidden"], cfg["ffn"], cfg["vocab"]
    hq, hkv, hd = cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
    w = {}
    # embed rows ~ N(0, 1/sqrt(D)): tied lm_head then yields O(1) logits.
    w["model.embed_tokens.weight"] = (rng.standard_normal((V, D)) / np.sqrt(D)).astype(np.float32)
    for i in range(cfg["layers"]):
        p = f"model.layers.{i}."
        w[p + "input_layernorm.weight"] = _vec(rng, D, 1.0, 0.1)
        w[p + "self_attn.q_proj.weight"] = _lin(rng, hq * hd, D)
        w[p + "self_attn.k_proj.weight"] = _lin(rng, hkv * hd, D)
        w[p + "self_attn.v_proj.weight"] = _lin(rng, hkv * hd, D)
        w[p + "self_attn.o_proj.weight"] = _lin(rng, D, hq * hd, 0.5)
        w[p + "self_attn.q_norm.weight"] = _vec(rng, hd, 1.0, 0.1)
        w[p + "self_attn.k_norm.weight"] = _vec(rng, hd, 1.0, 0.1)
        w[p + "post_attention_layernorm.weight"] = _vec(rng, D, 1.0, 0.1)
        w[p + "mlp.gate_proj.weight"] = _lin(rng, F, D)
        w[p + "mlp.up_proj.weight"] = _lin(rng, F, D)
        w[p + "mlp.down_proj.weight"] = _lin(rng, D, F, 0.5)
    w["model.norm.weight"] = _vec(rng, D, 1.0, 0.1)
    return w


def init_mlp_projector(enc_dim, llm_dim, hidden=None, k=4, seed=2):
    """tiny_audio/projectors.py:26-50. nn.Linear default init is
    kaiming-uniform(a=sqrt(5)) == U(-1/sqrt(in), 1/sqrt(in)); norms start at 1."""
    rng = np.random.RandomState(seed)
    H = hidden or llm_dim
    in_dim = enc_dim * k

    def ku(o, i):
        b = 1.0 / np.sqrt(i)
        return rng.uniform(-b, b, size=(o, i)).astype(np.float32)
    return {
        "linear_1.weight": ku(H, in_dim),
        "norm.weight": _vec(rng, H, 1.0, 0.05),
        "linear_2.weight": ku(llm_dim, H),
        "norm_2.weight": _vec(rng, llm_dim, 1.0, 0.05),
    }


def init_moe_projector(enc_dim, llm_dim, hidden=None, k=4, num_experts=4, seed=3):
    """tiny_audio/projectors.py:195-251 (_init_weights: router N(0,.02),
    fc1 xavier-uniform, fc2 N(0,.01), default-uniform biases)."""
    rng = np.random.RandomState(seed)
    H = hidden or llm_dim
    in_dim = enc_dim * k
    w = {"norm.weight": _vec(rng, in_dim, 1.0, 0.05),
         "router.weight": (0.02 * rng.standard_normal((num_experts, in_dim))).astype(np.float32)}

    def adapter(prefix):
        a = np.sqrt(6.0 / (in_dim + H))
        w[prefix + "fc1.weight"] = rng.uniform(-a, a, size=(H, in_dim)).astype(np.float32)
        w[prefix + "fc1.bias"] = rng.uniform(-1, 1, size=H).astype(np.float32) / np.float32(np.sqrt(in_dim))
        w[prefix + "fc2.weight"] = (0.01 * rng.standard_normal((llm_dim, H))).astype(np.float32)
        w[prefix + "fc2.bias"] = rng.uniform(-1, 1, size=llm_dim).astype(np.float32) / np.float32(np.sqrt(H))
    for e in range(num_experts):
        adapter(f"experts.{e}.")
    adapter("shared_expert.")
    return w


def init_mosa_projector(enc_dim, llm_dim, num_experts=4, adapter_hidden=4096, router_hidden=512, seed=6):
    """tiny_audio/projectors.py:100-150 (torch default inits: uniform(+-1/sqrt(fan_in)) weights and biases)."""
    rng = np.random.RandomState(seed)
    u = lambda fan_in, *s: (rng.uniform(-1, 1, size=s) / np.sqrt(fan_in)).astype(np.float32)
    w = {"downsampler.0.weight": u(3 * enc_dim, enc_dim, enc_dim, 3), "downsampler.0.bias": u(3 * enc_dim, enc_dim),
         "downsampler.2.weight": u(3 * enc_dim, llm_dim, enc_dim, 3), "downsampler.2.bias": u(3 * enc_dim, llm_dim),
         "router.0.weight": u(llm_dim, router_hidden, llm_dim), "router.0.bias": u(llm_dim, router_hidden),
         "router.2.weight": u(router_hidden, num_experts, router_hidden), "router.2.bias": u(router_hidden, num_experts)}
    for e in range(num_experts):
        w[f"experts.{e}.fc1.weight"] = u(llm_dim, adapter_hidden, llm_dim); w[f"experts.{e}.fc1.bias"] = u(llm_dim, adapter_hidden)
        w[f"experts.{e}.fc2.weight"] = u(adapter_hidden, llm_dim, adapter_hidden); w[f"experts.{e}.fc2.bias"] = u(adapter_hidden, llm_dim)
    return w


def init_qformer_projector(enc_dim, llm_dim, hidden=None, layers=2, ffn=None, nq=3, seed=5):
    """tiny_audio/projectors.py:359-420: query ~ N(0, 1); Blip2QFormerModel weights N(0, 0.02) (initializer_range),
    LayerNorm 1 / 0 (perturbed here so that the affine gradients are exercised), final Linear default-uniform."""
    rng = np.random.RandomState(seed)
    H = hidden or enc_dim
    F = ffn or 4 * H
    n = lambda *s: (0.02 * rng.standard_normal(s)).astype(np.float32)
    w = {"query": rng.standard_normal((1, nq, H)).astype(np.float32),
         "qformer.layernorm.weight": _vec(rng, H, 1.0, 0.05), "qformer.layernorm.bias": _vec(rng, H, 0.0, 0.05)}
    if enc_dim != H:
        w["encoder_proj.weight"] = rng.uniform(-1, 1, size=(H, enc_dim)).astype(np.float32) / np.float32(np.sqrt(enc_dim))
    for i in range(layers):
        p = f"qformer.encoder.layer.{i}."
        for att in ("attention.", "crossattention."):
            for m in ("query", "key", "value"):
                w[p + att + f"attention.{m}.weight"] = n(H, H); w[p + att + f"attention.{m}.bias"] = n(H)
            w[p + att + "output.dense.weight"] = n(H, H); w[p + att + "output.dense.bias"] = n(H)
            w[p + att + "output.LayerNorm.weight"] = _vec(rng, H, 1.0, 0.05); w[p + att + "output.LayerNorm.bias"] = _vec(rng, H, 0.0, 0.05)
        w[p + "intermediate_query.dense.weight"] = n(F, H); w[p + "intermediate_query.dense.bias"] = n(F)
        w[p + "output_query.dense.weight"] = n(H, F); w[p + "output_query.dense.bias"] = n(H)
        w[p + "output_query.LayerNorm.weight"] = _vec(rng, H, 1.0, 0.05); w[p + "output_query.LayerNorm.bias"] = _vec(rng, H, 0.0, 0.05)
    w["linear.weight"] = rng.uniform(-1, 1, size=(llm_dim, H)).astype(np.float32) / np.float32(np.sqrt(H))
    w["linear.bias"] = rng.uniform(-1, 1, size=llm_dim).astype(np.float32) / np.float32(np.sqrt(H))
    return w


def init_lora(cfg, rank=8, seed=4, b_std=0.02, targets=None):
    """LoRA adapters on q,k,v,o,gate,up,down of every layer (tiny_audio/asr_config.py:142-150, r=8), or on the ``targets``
    subset of them (peft suffix names, e.g. ("q_proj", "v_proj")).  peft initialises lora_A kaiming-uniform and lora_B = 0;
    B is given small random values here so that parity is non-trivial."""
    rng = np.random.RandomState(seed)
    D, F = cfg["hidden"], cfg["ffn"]
    hq, hkv, hd = cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
    dims = {"self_attn.q_proj": (hq * hd, D), "self_attn.k_proj": (hkv * hd, D), "self_attn.v_proj": (hkv * hd, D),
            "self_attn.o_proj": (D, hq * hd), "mlp.gate_proj": (F, D), "mlp.up_proj": (F, D), "mlp.down_proj": (D, F)}
    lo = {}
    for i in range(cfg["layers"]):
        for n, (o, k) in dims.items():
            if targets is not None and n.split(".")[-1] not in targets:
                continue
            b = 1.0 / np.sqrt(k)
            lo[f"model.layers.{i}.{n}.lora_A"] = rng.uniform(-b, b, size=(rank, k)).astype(np.float32)
            lo[f"model.layers.{i}.{n}.lora_B"] = (b_std * rng.standard_normal((o, rank))).astype(np.float32)
    return lo


def synthetic_wave(b: int, n: int = 160000) -> np.ndarray:
    """wav[b] = 0.1*N(0,1), RandomState(1234+b)."""
    return (0.1 * np.random.RandomState(1234 + b).standard_normal(n)).astype(np.float32)


def synthetic_tokens(B, n_audio, vocab, audio_id, pad_id, eos_id, L=None,
                     n_prefix=3, n_suffix=24, n_text=35, ragged=False):
    """Token stream of SURVEY.md section 8(d) (no tokenizer offline):
    [prefix | <audio>*n | suffix | transcript | <|im_end|> | pad...], labels
    -100 except transcript + <|im_end|>.  ``n_audio`` may be an int or a
    per-sample list (ragged batches); ``ragged`` shortens transcripts too."""
    counts = [n_audio] * B if np.isscalar(n_audio) else list(n_audio)
    texts = [n_text - (3 * b if ragged else 0) for b in range(B)]
    need = max(n_prefix + c + n_suffix + t + 1 for c, t in zip(counts, texts))
    L = L or need
    assert L >= need
    ids = np.full((B, L), pad_id, dtype=np.int64)
    att = np.zeros((B, L), dtype=np.int64)
    lab = np.full((B, L), -100, dtype=np.int64)
    hi = min(vocab, audio_id) - 30
    for b in range(B):
        rng = np.random.RandomState(99 + b)
        c, t = counts[b], texts[b]
        pos = 0
        ids[b, pos:pos + n_prefix] = rng.randint(0, hi, n_prefix); pos += n_prefix
        ids[b, pos:pos + c] = audio_id; pos += c
        ids[b, pos:pos + n_suffix] = rng.randint(0, hi, n_suffix); pos += n_suffix
        ids[b, pos:pos + t] = rng.randint(0, hi, t)
        lab[b, pos:pos + t] = ids[b, pos:pos + t]; pos += t
        ids[b, pos] = eos_id; lab[b, pos] = eos_id; pos += 1
        att[b, :pos] = 1
    return ids, att, lab, np.asarray(counts, dtype=np.int64)
