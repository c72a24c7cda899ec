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


def rms_bwd_dw(dy, x, r):
    """d loss / d weight of Qwen3RMSNorm (y = w * x * r): sum over every leading axis."""
    g = dy * (x * r)
    return g.reshape(-1, g.shape[-1]).sum(0).astype(np.float32)


def _acc(wgrads, name, g):
    """Full decoder fine-tuning (freeze_language_model=False, tiny_audio/asr_modeling.py:251-253): collect d loss / d W."""
    if wgrads is not None:
        wgrads[name] = (wgrads[name] + g if name in wgrads else g).astype(np.float32)


def silu(x):
    return x / (1.0 + np.exp(-x))


def attn_allowed(att_mask, L):
    """create_causal_mask (modeling_qwen3.py:392-408): key j visible to query i
    iff j <= i and attention_mask[b, j] == 1."""
    causal = np.tril(np.ones((L, L), dtype=bool))
    if att_mask is None:
        return causal[None]
    return causal[None] & (np.asarray(att_mask) != 0)[:, None, :]


LORA_TARGETS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")        # tiny_audio/asr_config.py:142-150


def _eff(w, lora, name, scale):
    """Effective weight of a LoRA-adapted linear: W + (alpha/r) B A  (peft LoraLayer: result = base(x) +
    lora_B(lora_A(dropout(x))) * scaling, dropout 0; wired at tiny_audio/asr_modeling.py:289-301).  peft itself is not
    installed offline, so this is a restatement of its documented formula."""
    W = w[name + ".weight"]
    if lora is None or (name + ".lora_A") not in lora:
        return W
    return (W + np.float32(scale) * (lora[name + ".lora_B"] @ lora[name + ".lora_A"])).astype(np.float32)


def _lora_grads(grads, lora, name, scale, dW):
    """dA = s B^T dW, dB = s dW A^T for W_eff = W + s B A."""
    if grads is None or lora is None or (name + ".lora_A") not in lora:
        return
    A, Bm = lora[name + ".lora_A"], lora[name + ".lora_B"]
    grads[name + ".lora_A"] = (np.float32(scale) * (Bm.T @ dW)).astype(np.float32)
    grads[name + ".lora_B"] = (np.float32(scale) * (dW @ A.T)).astype(np.float32)


def layer_forward(x, w, p, cfg, cos, sin, allowed, lora=None, lora_scale=0.0):
    B, L, D = x.shape
    w = {**w, **{p + n + ".weight": _eff(w, lora, p + n, lora_scale) for n in LORA_TARGETS}} if lora is not None else w
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
                 x1=x1, r_post=r_post, xn2=xn2, gt=gt, up=up, act=act)
    return x2.astype(np.float32), cache


def layer_backward_dx(dx2, w, p, cfg, cos, sin, c, lora=None, lora_scale=0.0, grads=None, wgrads=None):
    """dL/dx for a decoder layer; with ``lora`` also the adapter gradients, with ``wgrads`` (a dict) also the gradient
    of every base weight of the layer (full decoder fine-tuning)."""
    B, L, D = dx2.shape
    w = {**w, **{p + n + ".weight": _eff(w, lora, p + n, lora_scale) for n in LORA_TARGETS}} if lora is not None else w
    flat = lambda a: a.reshape(-1, a.shape[-1])
    _lora_grads(grads, lora, p + "mlp.down_proj", lora_scale, flat(dx2).T @ flat(c["act"]))
    if wgrads is not None:
        _acc(wgrads, p + "mlp.down_proj.weight", flat(dx2).T @ flat(c["act"]))
    hq, hkv, hd = cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
    g = hq // hkv
    # MLP
    dact = dx2 @ w[p + "mlp.down_proj.weight"]
    sg = 1.0 / (1.0 + np.exp(-c["gt"]))
    dgt = dact * c["up"] * (sg * (1.0 + c["gt"] * (1.0 - sg)))
    dup = dact * (c["gt"] * sg)
    _lora_grads(grads, lora, p + "mlp.gate_proj", lora_scale, flat(dgt).T @ flat(c["xn2"]))
    _lora_grads(grads, lora, p + "mlp.up_proj", lora_scale, flat(dup).T @ flat(c["xn2"]))
    dxn2 = dgt @ w[p + "mlp.gate_proj.weight"] + dup @ w[p + "mlp.up_proj.weight"]
    if wgrads is not None:
        _acc(wgrads, p + "mlp.gate_proj.weight", flat(dgt).T @ flat(c["xn2"]))
        _acc(wgrads, p + "mlp.up_proj.weight", flat(dup).T @ flat(c["xn2"]))
        _acc(wgrads, p + "post_attention_layernorm.weight", rms_bwd_dw(dxn2, c["x1"], c["r_post"]))
    dx1 = dx2 + rms_bwd_dx(dxn2, c["x1"], c["r_post"], w[p + "post_attention_layernorm.weight"])
    # attention
    _lora_grads(grads, lora, p + "self_attn.o_proj", lora_scale, flat(dx1).T @ flat(c["ao"]))
    if wgrads is not None:
        _acc(wgrads, p + "self_attn.o_proj.weight", flat(dx1).T @ flat(c["ao"]))
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
    cb, sb = (cos[None, None], sin[None, None]) if cos.ndim == 2 else (cos[:, None], sin[:, None])   # per-clip position ids

    def rope_bwd(dy):
        return dy * cb - rotate_half(dy * sb)
    dqn = rope_bwd(dq).transpose(0, 2, 1, 3)
    dkn = rope_bwd(dk).transpose(0, 2, 1, 3)
    dq0 = rms_bwd_dx(dqn, c["q0"], c["rq"], w[p + "self_attn.q_norm.weight"]).reshape(B, L, hq * hd)
    dk0 = rms_bwd_dx(dkn, c["k0"], c["rk"], w[p + "self_attn.k_norm.weight"]).reshape(B, L, hkv * hd)
    dv0 = dv.transpose(0, 2, 1, 3).reshape(B, L, hkv * hd)
    _lora_grads(grads, lora, p + "self_attn.q_proj", lora_scale, flat(dq0).T @ flat(c["xn"]))
    _lora_grads(grads, lora, p + "self_attn.k_proj", lora_scale, flat(dk0).T @ flat(c["xn"]))
    _lora_grads(grads, lora, p + "self_attn.v_proj", lora_scale, flat(dv0).T @ flat(c["xn"]))
    dxn = (dq0 @ w[p + "self_attn.q_proj.weight"] + dk0 @ w[p + "self_attn.k_proj.weight"]
           + dv0 @ w[p + "self_attn.v_proj.weight"])
    if wgrads is not None:
        _acc(wgrads, p + "self_attn.q_norm.weight", rms_bwd_dw(dqn, c["q0"], c["rq"]))
        _acc(wgrads, p + "self_attn.k_norm.weight", rms_bwd_dw(dkn, c["k0"], c["rk"]))
        _acc(wgrads, p + "self_attn.q_proj.weight", flat(dq0).T @ flat(c["xn"]))
        _acc(wgrads, p + "self_attn.k_proj.weight", flat(dk0).T @ flat(c["xn"]))
        _acc(wgrads, p + "self_attn.v_proj.weight", flat(dv0).T @ flat(c["xn"]))
        _acc(wgrads, p + "input_layernorm.weight", rms_bwd_dw(dxn, c["x"], c["r_in"]))
    dx = dx1 + rms_bwd_dx(dxn, c["x"], c["r_in"], w[p + "input_layernorm.weight"])
    return dx.astype(np.float32)


def lm_forward(inputs_embeds, att_mask, w, cfg, position_ids=None, keep_cache=True, lora=None, lora_scale=0.0):
    """inputs_embeds [B, L, D] -> logits [B, L, V] (tied lm_head, :485-487)."""
    B, L, D = inputs_embeds.shape
    pos = np.arange(L) if position_ids is None else np.asarray(position_ids)
    cos_t, sin_t = rope_tables(int(pos.max()) + 1, cfg["head_dim"], cfg["rope_theta"])
    cos, sin = cos_t[pos], sin_t[pos]
    allowed = attn_allowed(att_mask, L)
    x = inputs_embeds.astype(np.float32)
    caches = []
    for i in range(cfg["layers"]):
        x, c = layer_forward(x, w, f"model.layers.{i}.", cfg, cos, sin, allowed, lora, lora_scale)
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


def lm_backward_dx(dlogits, w, cfg, cache, lora=None, lora_scale=0.0, grads=None, wgrads=None):
    """d loss / d inputs_embeds through the LM (and, with ``lora``/``grads``, the adapter gradients; with ``wgrads`` the
    gradient of every LM weight -- the tied lm_head's share of ``model.embed_tokens.weight`` included, the input-lookup
    share is added by the caller that owns the token ids)."""
    dhn = dlogits @ w["model.embed_tokens.weight"]
    if wgrads is not None:
        V = dlogits.shape[-1]
        _acc(wgrads, "model.embed_tokens.weight", dlogits.reshape(-1, V).T @ cache["hn"].reshape(-1, cache["hn"].shape[-1]))
        _acc(wgrads, "model.norm.weight", rms_bwd_dw(dhn, cache["x_final"], cache["r_f"]))
    dx = rms_bwd_dx(dhn, cache["x_final"], cache["r_f"], w["model.norm.weight"])
    for i in reversed(range(cfg["layers"])):
        dx = layer_backward_dx(dx, w, f"model.layers.{i}.", cfg, cache["cos"], cache["sin"],
                               cache["layers"][i], lora, lora_scale, grads, wgrads)
    return dx
