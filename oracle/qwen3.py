"""Frozen Qwen3 causal LM: forward, shifted cross-entropy, and the
activation-gradient (dX-only) backward (rows a9/a10 of SURVEY.md section 8).

Restates TF:models/qwen3/modeling_qwen3.py (Qwen3RMSNorm :50-64, Qwen3MLP
:70-83, rotary :86-170, Qwen3Attention :211-280, Qwen3DecoderLayer :283-324,
Qwen3Model.forward :367-428, Qwen3ForCausalLM.forward :448-508) and
TF:loss/loss_utils.py:33-71 (ForCausalLMLoss / fixed_cross_entropy).
"""
from __future__ import annotations

import numpy as np

from .encoder import apply_rope, rope_tables, rotate_half


def rms(x, w, eps):
    r = 1.0 / np.sqrt((x ** 2).mean(-1, keepdims=True) + eps)
    return (w * (x * r)).astype(np.float32), r.astype(np.float32)


def rms_bwd_dx(dy, x, r, w):
    xh = x * r
    dn = dy * w
    return (r * (dn - xh * (dn * xh).mean(-1, keepdims=True))).astype(np.float32)


def silu(x):
    return x / (1.0 + np.exp(-x))


def attn_allowed(att_mask, L):
    """create_causal_mask (modeling_qwen3.py:392-408): key j visible to query i
    iff j <= i and attention_mask[b, j] == 1."""
    causal = np.tril(np.ones((L, L), dtype=bool))
    if att_mask is None:
        return causal[None]
    return causal[None] & (np.asarray(att_mask) != 0)[:, None, :]


def layer_forward(x, w, p, cfg, cos, sin, allowed):
    B, L, D = x.shape
    hq, hkv, hd, eps = cfg["heads"], cfg["kv_heads"], cfg["head_dim"], cfg["rms_eps"]
    g = hq // hkv
    xn, r_in = rms(x, w[p + "input_layernorm.weight"], eps)
    q0 = (xn @ w[p + "self_attn.q_proj.weight"].T).reshape(B, L, hq, hd)
    k0 = (xn @ w[p + "self_attn.k_proj.weight"].T).reshape(B, L, hkv, hd)
    v = (xn @ w[p + "self_attn.v_proj.weight"].T).reshape(B, L, hkv, hd).transpose(0, 2, 1, 3)
    qn, rq = rms(q0, w[p + "self_attn.q_norm.weight"], eps)      # per-head norm :251-252
    kn, rk = rms(k0, w[p + "self_attn.k_norm.weight"], eps)
    q = apply_rope(qn.transpose(0, 2, 1, 3), cos, sin)
    k = apply_rope(kn.transpose(0, 2, 1, 3), cos, sin)
    kr = np.repeat(k, g, axis=1)                                  # repeat_kv :173-182
    vr = np.repeat(v, g, axis=1)
    s = (q @ kr.transpose(0, 1, 3, 2)) * np.float32(hd ** -0.5)
    s = np.where(allowed[:, None], s, -np.inf)
    m = s.max(-1, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    e = np.exp(s - m)
    den = e.sum(-1, keepdims=True)
    P = np.where(den > 0, e / np.maximum(den, 1e-30), 0.0).astype(np.float32)
    ao = (P @ vr).transpose(0, 2, 1, 3).reshape(B, L, hq * hd)
    x1 = x + ao @ w[p + "self_attn.o_proj.weight"].T
    xn2, r_post = rms(x1, w[p + "post_attention_layernorm.weight"], eps)
    gt = xn2 @ w[p + "mlp.gate_proj.weight"].T
    up = xn2 @ w[p + "mlp.up_proj.weight"].T
    act = silu(gt) * up
    x2 = x1 + act @ w[p + "mlp.down_proj.weight"].T
    cache = dict(x=x, r_in=r_in, xn=xn, q0=q0, k0=k0, rq=rq, rk=rk, q=q, k=k, v=v, P=P, ao=ao,
                 x1=x1, r_post=r_post, xn2=xn2, gt=gt, up=up)
    return x2.astype(np.float32), cache


def layer_backward_dx(dx2, w, p, cfg, cos, sin, c):
    """dL/dx for a frozen decoder layer (no weight gradients)."""
    B, L, D = dx2.shape
    hq, hkv, hd = cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
    g = hq // hkv
    # MLP
    dact = dx2 @ w[p + "mlp.down_proj.weight"]
    sg = 1.0 / (1.0 + np.exp(-c["gt"]))
    dgt = dact * c["up"] * (sg * (1.0 + c["gt"] * (1.0 - sg)))
    dup = dact * (c["gt"] * sg)
    dxn2 = dgt @ w[p + "mlp.gate_proj.weight"] + dup @ w[p + "mlp.up_proj.weight"]
    dx1 = dx2 + rms_bwd_dx(dxn2, c["x1"], c["r_post"], w[p + "post_attention_layernorm.weight"])
    # attention
    dao = (dx1 @ w[p + "self_attn.o_proj.weight"]).reshape(B, L, hq, hd).transpose(0, 2, 1, 3)
    kr = np.repeat(c["k"], g, axis=1)
    vr = np.repeat(c["v"], g, axis=1)
    P = c["P"]
    dvr = P.transpose(0, 1, 3, 2) @ dao
    dP = dao @ vr.transpose(0, 1, 3, 2)
    dS = P * (dP - (dP * P).sum(-1, keepdims=True)) * np.float32(hd ** -0.5)
    dq = dS @ kr
    dkr = dS.transpose(0, 1, 3, 2) @ c["q"]
    dk = dkr.reshape(B, hkv, g, L, hd).sum(2)
    dv = dvr.reshape(B, hkv, g, L, hd).sum(2)
    # rope backward: y = x*cos + rot(x)*sin  =>  dx = dy*cos - rot(dy*sin)
    def rope_bwd(dy):
        return dy * cos[None, None] - rotate_half(dy * sin[None, None])
    dqn = rope_bwd(dq).transpose(0, 2, 1, 3)
    dkn = rope_bwd(dk).transpose(0, 2, 1, 3)
    dq0 = rms_bwd_dx(dqn, c["q0"], c["rq"], w[p + "self_attn.q_norm.weight"]).reshape(B, L, hq * hd)
    dk0 = rms_bwd_dx(dkn, c["k0"], c["rk"], w[p + "self_attn.k_norm.weight"]).reshape(B, L, hkv * hd)
    dv0 = dv.transpose(0, 2, 1, 3).reshape(B, L, hkv * hd)
    dxn = (dq0 @ w[p + "self_attn.q_proj.weight"] + dk0 @ w[p + "self_attn.k_proj.weight"]
           + dv0 @ w[p + "self_attn.v_proj.weight"])
    dx = dx1 + rms_bwd_dx(dxn, c["x"], c["r_in"], w[p + "input_layernorm.weight"])
    return dx.astype(np.float32)


def lm_forward(inputs_embeds, att_mask, w, cfg, position_ids=None, keep_cache=True):
    """inputs_embeds [B, L, D] -> logits [B, L, V] (tied lm_head, :485-487)."""
    B, L, D = inputs_embeds.shape
    pos = np.arange(L) if position_ids is None else np.asarray(position_ids)
    cos_t, sin_t = rope_tables(int(pos.max()) + 1, cfg["head_dim"], cfg["rope_theta"])
    cos, sin = cos_t[pos], sin_t[pos]
    allowed = attn_allowed(att_mask, L)
    x = inputs_embeds.astype(np.float32)
    caches = []
    for i in range(cfg["layers"]):
        x, c = layer_forward(x, w, f"model.layers.{i}.", cfg, cos, sin, allowed)
        caches.append(c if keep_cache else None)
    hn, r_f = rms(x, w["model.norm.weight"], cfg["rms_eps"])
    logits = hn @ w["model.embed_tokens.weight"].T
    return logits.astype(np.float32), dict(layers=caches, x_final=x, r_f=r_f, hn=hn, cos=cos, sin=sin)


def causal_lm_loss(logits, labels, num_items_in_batch=None):
    """ForCausalLMLoss TF:loss/loss_utils.py:48-71: labels padded with -100 and
    shifted by one, CE over non-ignored targets; mean, or sum/num_items."""
    B, L, V = logits.shape
    shift = np.concatenate([labels[:, 1:], np.full((B, 1), -100, dtype=labels.dtype)], axis=1)
    flat = logits.reshape(-1, V).astype(np.float64)
    tgt = shift.reshape(-1)
    valid = tgt != -100
    m = flat.max(-1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(flat - m).sum(-1))
    nll = lse - flat[np.arange(flat.shape[0]), np.where(valid, tgt, 0)]
    n = valid.sum() if num_items_in_batch is None else num_items_in_batch
    loss = (nll * valid).sum() / max(n, 1)
    probs = np.exp(flat - lse[:, None])
    dlogits = probs
    dlogits[np.arange(flat.shape[0]), np.where(valid, tgt, 0)] -= 1.0
    dlogits = dlogits * (valid[:, None] / max(n, 1))
    return np.float32(loss), dlogits.reshape(B, L, V).astype(np.float32), int(valid.sum())


def lm_backward_dx(dlogits, w, cfg, cache):
    """d loss / d inputs_embeds through the frozen LM."""
    dhn = dlogits @ w["model.embed_tokens.weight"]
    dx = rms_bwd_dx(dhn, cache["x_final"], cache["r_f"], w["model.norm.weight"])
    for i in reversed(range(cfg["layers"])):
        dx = layer_backward_dx(dx, w, f"model.layers.{i}.", cfg, cache["cos"], cache["sin"],
                               cache["layers"][i])
    return dx
