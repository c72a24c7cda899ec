"""Frozen GLM-ASR encoder forward (row a3 of SURVEY.md section 8), numpy.

Restates ``GlmAsrEncoder.forward`` TF:models/glmasr/modeling_glmasr.py:313-327
and the layers it calls.  Forward only: the reference runs it under
``torch.no_grad()`` (tiny_audio/asr_modeling.py:448-450).
"""
from __future__ import annotations

import numpy as np
from scipy.special import erf


def gelu(x):
    """nn.functional.gelu, exact erf form (modeling_glmasr.py:314-315)."""
    return (0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))).astype(x.dtype)


def layer_norm(x, w, b, eps=1e-5):
    """nn.LayerNorm (modeling_glmasr.py:246-247,305): biased variance."""
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return ((x - mu) / np.sqrt(var + eps) * w + b).astype(x.dtype)


def conv1d(x, w, b, stride):
    """nn.Conv1d(k=3, padding=1) on [B, Cin, T] (modeling_glmasr.py:299-300)."""
    B, Cin, T = x.shape
    Cout, _, K = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1)))
    Tout = (T + 2 - (K - 1) - 1) // stride + 1
    out = np.zeros((B, Cout, Tout), dtype=x.dtype)
    for k in range(K):
        seg = xp[:, :, k:k + stride * (Tout - 1) + 1:stride]        # [B, Cin, Tout]
        out += np.einsum("oc,bct->bot", w[:, :, k], seg)
    return out + b[None, :, None]


def rope_tables(n_pos, rot_dim, theta, dtype=np.float32):
    """cos/sin [n_pos, rot_dim] as GlmAsrRotaryEmbedding / Qwen3RotaryEmbedding
    build them (modeling_glmasr.py:68-106; modeling_qwen3.py:101-133):
    inv_freq = theta^(-2i/rot_dim), emb = cat(freqs, freqs)."""
    inv = 1.0 / (theta ** (np.arange(0, rot_dim, 2, dtype=np.float32) / np.float32(rot_dim)))
    freqs = np.arange(n_pos, dtype=np.float32)[:, None] * inv[None, :].astype(np.float32)
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb).astype(dtype), np.sin(emb).astype(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def apply_rope(x, cos, sin):
    """x [B, heads, S, hd]; cos/sin [S, rot] (or [B, S, rot] for per-clip position ids); first ``rot`` dims
    rotated (modeling_glmasr.py:153-168: partial rotary; Qwen3 uses rot == hd)."""
    rot = cos.shape[-1]
    xr, xp = x[..., :rot], x[..., rot:]
    c, s_ = (cos[None, None], sin[None, None]) if cos.ndim == 2 else (cos[:, None], sin[:, None])
    xr = xr * c + rotate_half(xr) * s_
    return np.concatenate([xr, xp], axis=-1).astype(x.dtype)


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def encoder_layer(x, w, p, cfg, cos, sin):
    """GlmAsrEncoderLayer.forward modeling_glmasr.py:249-270, attention
    :187-221 (non-causal, attention_mask=None, scale head_dim**-0.5)."""
    B, S, H = x.shape
    nh, hd = cfg["heads"], cfg["head_dim"]
    h = layer_norm(x, w[p + "input_layernorm.weight"], w[p + "input_layernorm.bias"], cfg["ln_eps"])
    q = h @ w[p + "self_attn.q_proj.weight"].T + w[p + "self_attn.q_proj.bias"]
    k = h @ w[p + "self_attn.k_proj.weight"].T
    v = h @ w[p + "self_attn.v_proj.weight"].T + w[p + "self_attn.v_proj.bias"]
    q = q.reshape(B, S, nh, hd).transpose(0, 2, 1, 3)
    k = k.reshape(B, S, nh, hd).transpose(0, 2, 1, 3)
    v = v.reshape(B, S, nh, hd).transpose(0, 2, 1, 3)
    q = apply_rope(q, cos, sin)
    k = apply_rope(k, cos, sin)
    att = softmax((q @ k.transpose(0, 1, 3, 2)) * np.float32(hd ** -0.5))
    o = (att @ v).transpose(0, 2, 1, 3).reshape(B, S, H)
    o = o @ w[p + "self_attn.o_proj.weight"].T + w[p + "self_attn.o_proj.bias"]
    x = x + o
    h = layer_norm(x, w[p + "post_attention_layernorm.weight"],
                   w[p + "post_attention_layernorm.bias"], cfg["ln_eps"])
    h = gelu(h @ w[p + "mlp.fc1.weight"].T + w[p + "mlp.fc1.bias"])
    h = h @ w[p + "mlp.fc2.weight"].T + w[p + "mlp.fc2.bias"]
    return (x + h).astype(np.float32)


def conv_frontend(feats, w):
    """modeling_glmasr.py:314-316: gelu(conv1) -> gelu(conv2, stride 2) -> [B, S, H]."""
    x = gelu(conv1d(feats.astype(np.float32), w["conv1.weight"], w["conv1.bias"], 1))
    x = gelu(conv1d(x, w["conv2.weight"], w["conv2.bias"], 2))
    return np.ascontiguousarray(x.transpose(0, 2, 1))


def encoder_forward(feats, w, cfg, return_all=False):
    """[B, n_mels, T] -> last_hidden_state [B, S, H] (modeling_glmasr.py:313-327).
    NB: no attention mask -- padded frames are attended to, as in the reference."""
    x = conv_frontend(feats, w)
    S = x.shape[1]
    rot = int(cfg["head_dim"] * cfg["partial_rotary"])
    cos, sin = rope_tables(S, rot, cfg["rope_theta"])
    hs = [x]
    for i in range(cfg["layers"]):
        x = encoder_layer(x, w, f"layers.{i}.", cfg, cos, sin)
        if return_all:
            hs.append(x)
    out = layer_norm(x, w["norm.weight"], w["norm.bias"], cfg["ln_eps"])
    return (out, hs) if return_all else out
