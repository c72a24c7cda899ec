"""QFormer audio projector (north_star "MLP/MoE/QFormer projectors"), numpy.
TEST INFRASTRUCTURE ONLY.

tiny_audio/projectors.py:359-475 (QFormerAudioProjector: learnable queries, windows of 15 frames, 3 queries per
window, final Linear) on top of ``Blip2QFormerModel`` (TF:models/blip_2/modeling_blip_2.py: Blip2QFormerModel.forward
-> layernorm(query) -> Blip2QFormerLayer x N: self-attention + SelfOutput(dense, +res, LayerNorm), cross-attention
to the window, intermediate_query (dense + GELU) / output_query (dense, +res, LayerNorm); cross_attention_frequency=1,
layer_norm_eps 1e-12).  Eval-mode arithmetic: the 0.1 hidden / attention-probs dropouts of the Granite config cannot be
pinned (RNG); ``keep`` masks can be injected to test the dropout algebra.
Backward is hand-written (every parameter is trainable; no gradient w.r.t. the frozen encoder output is needed).
"""
from __future__ import annotations

import math

import numpy as np

from .projectors import gelu, gelu_grad


def layer_norm(x, g, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xh = (x - mu) * rstd
    return (xh * g + b).astype(np.float32), (xh.astype(np.float32), rstd.astype(np.float32))


def layer_norm_bwd(dy, cache, g):
    xh, rstd = cache
    gg = dy * g
    dx = rstd * (gg - gg.mean(-1, keepdims=True) - xh * (gg * xh).mean(-1, keepdims=True))
    red = tuple(range(dy.ndim - 1))
    return dx.astype(np.float32), (dy * xh).sum(red).astype(np.float32), dy.sum(red).astype(np.float32)


def _heads(x, nh):
    EB, L, H = x.shape
    return x.reshape(EB, L, nh, H // nh).transpose(0, 2, 1, 3)


def _attn_fwd(xq, xkv, w, p, nh, keep_p=None):
    q = xq @ w[p + "query.weight"].T + w[p + "query.bias"]
    k = xkv @ w[p + "key.weight"].T + w[p + "key.bias"]
    v = xkv @ w[p + "value.weight"].T + w[p + "value.bias"]
    qh, kh, vh = _heads(q, nh), _heads(k, nh), _heads(v, nh)
    scale = np.float32((q.shape[-1] // nh) ** -0.5)
    s = (qh @ kh.transpose(0, 1, 3, 2)) * scale
    s = s - s.max(-1, keepdims=True)
    pr = np.exp(s); pr /= pr.sum(-1, keepdims=True)
    pd = pr if keep_p is None else pr * keep_p
    o = (pd @ vh).transpose(0, 2, 1, 3).reshape(q.shape)
    return o.astype(np.float32), dict(xq=xq, xkv=xkv, qh=qh, kh=kh, vh=vh, pr=pr, keep_p=keep_p, scale=scale)


def _attn_bwd(do, w, p, nh, c, grads):
    EB, Lq, H = do.shape
    doh = _heads(do, nh)
    pd = c["pr"] if c["keep_p"] is None else c["pr"] * c["keep_p"]
    dv = pd.transpose(0, 1, 3, 2) @ doh
    dpd = doh @ c["vh"].transpose(0, 1, 3, 2)
    dpr = dpd if c["keep_p"] is None else dpd * c["keep_p"]
    ds = c["pr"] * (dpr - (dpr * c["pr"]).sum(-1, keepdims=True))
    dq = (ds @ c["kh"]) * c["scale"]
    dk = (ds.transpose(0, 1, 3, 2) @ c["qh"]) * c["scale"]
    unh = lambda t: t.transpose(0, 2, 1, 3).reshape(t.shape[0], t.shape[2], H)
    dq, dk, dv = unh(dq), unh(dk), unh(dv)
    f = lambda t: t.reshape(-1, t.shape[-1])
    grads[p + "query.weight"] = f(dq).T @ f(c["xq"]); grads[p + "query.bias"] = f(dq).sum(0)
    grads[p + "key.weight"] = f(dk).T @ f(c["xkv"]); grads[p + "key.bias"] = f(dk).sum(0)
    grads[p + "value.weight"] = f(dv).T @ f(c["xkv"]); grads[p + "value.bias"] = f(dv).sum(0)
    dxq = dq @ w[p + "query.weight"]
    dxkv = dk @ w[p + "key.weight"] + dv @ w[p + "value.weight"]
    return dxq, dxkv


def _out_fwd(h, res, w, p, eps, keep=None):
    z = h @ w[p + "dense.weight"].T + w[p + "dense.bias"]
    if keep is not None:
        z = z * keep
    y, lc = layer_norm(z + res, w[p + "LayerNorm.weight"], w[p + "LayerNorm.bias"], eps)
    return y, dict(h=h, lc=lc, keep=keep)


def _out_bwd(dy, w, p, c, grads):
    du, dg, db = layer_norm_bwd(dy, c["lc"], w[p + "LayerNorm.weight"])
    grads[p + "LayerNorm.weight"], grads[p + "LayerNorm.bias"] = dg, db
    dz = du if c["keep"] is None else du * c["keep"]
    f = lambda t: t.reshape(-1, t.shape[-1])
    grads[p + "dense.weight"] = f(dz).T @ f(c["h"]); grads[p + "dense.bias"] = f(dz).sum(0)
    return dz @ w[p + "dense.weight"], du                       # (d dense input, d residual)


def output_length(S, window=15, downsample=5):
    return (S + window - 1) // window * (window // downsample)


def qformer_forward(hs, w, cfg, keeps=None):
    """hs [B, S, E] -> [B, nblocks * nq, D].  cfg: heads, layers, window (15), downsample (5), eps (1e-12).
    ``keeps``: optional dict of dropout keep masks ALREADY divided by (1 - p) (keys 'emb', 'l{i}.sa', 'l{i}.sa_p',
    'l{i}.ca', 'l{i}.ca_p', 'l{i}.ffn')."""
    keeps = keeps or {}
    B, S, E = hs.shape
    win, nq, nh, eps = cfg.get("window", 15), cfg.get("window", 15) // cfg.get("downsample", 5), cfg["heads"], cfg.get("eps", 1e-12)
    x = hs.astype(np.float32)
    c = {}
    if "encoder_proj.weight" in w:
        c["x_in"] = x
        x = x @ w["encoder_proj.weight"].T
    nb = math.ceil(S / win)
    if nb * win > S:
        x = np.concatenate([x, np.zeros((B, nb * win - S, x.shape[-1]), np.float32)], axis=1)
    enc = x.reshape(B * nb, win, -1)
    q0 = np.broadcast_to(w["query"], (B * nb, nq, w["query"].shape[-1])).astype(np.float32)
    h, c["emb"] = layer_norm(q0, w["qformer.layernorm.weight"], w["qformer.layernorm.bias"], eps)
    if "emb" in keeps:
        h = h * keeps["emb"]
    layers = []
    for i in range(cfg["layers"]):
        p = f"qformer.encoder.layer.{i}."
        lc = {}
        a, lc["sa"] = _attn_fwd(h, h, w, p + "attention.attention.", nh, keeps.get(f"l{i}.sa_p"))
        a, lc["sa_o"] = _out_fwd(a, h, w, p + "attention.output.", eps, keeps.get(f"l{i}.sa"))
        cx, lc["ca"] = _attn_fwd(a, enc, w, p + "crossattention.attention.", nh, keeps.get(f"l{i}.ca_p"))
        cx, lc["ca_o"] = _out_fwd(cx, a, w, p + "crossattention.output.", eps, keeps.get(f"l{i}.ca"))
        hi = cx @ w[p + "intermediate_query.dense.weight"].T + w[p + "intermediate_query.dense.bias"]
        ai = gelu(hi)
        h, lc["ffn_o"] = _out_fwd(ai, cx, w, p + "output_query.", eps, keeps.get(f"l{i}.ffn"))
        lc["cx"], lc["hi"] = cx, hi
        layers.append(lc)
    c.update(layers=layers, h_last=h, enc=enc, B=B, nb=nb, nq=nq, keeps=keeps)
    y = h.reshape(B, nb * nq, -1) @ w["linear.weight"].T + w["linear.bias"]
    return y.astype(np.float32), c


def qformer_backward(dy, w, cfg, c):
    """-> gradients of every parameter (dict with the reference's state_dict names)."""
    nh = cfg["heads"]
    g = {}
    B, nb, nq = c["B"], c["nb"], c["nq"]
    f = lambda t: t.reshape(-1, t.shape[-1])
    dy = dy.astype(np.float32)
    g["linear.weight"] = f(dy).T @ f(c["h_last"]); g["linear.bias"] = f(dy).sum(0)
    dh = (dy @ w["linear.weight"]).reshape(B * nb, nq, -1)
    denc = np.zeros_like(c["enc"])
    for i in reversed(range(cfg["layers"])):
        p = f"qformer.encoder.layer.{i}."
        lc = c["layers"][i]
        dai, dcx = _out_bwd(dh, w, p + "output_query.", lc["ffn_o"], g)
        dhi = dai * gelu_grad(lc["hi"])
        g[p + "intermediate_query.dense.weight"] = f(dhi).T @ f(lc["cx"]); g[p + "intermediate_query.dense.bias"] = f(dhi).sum(0)
        dcx = dcx + dhi @ w[p + "intermediate_query.dense.weight"]
        dco, da = _out_bwd(dcx, w, p + "crossattention.output.", lc["ca_o"], g)
        dxq, dxkv = _attn_bwd(dco, w, p + "crossattention.attention.", nh, lc["ca"], g)
        da = da + dxq
        denc += dxkv
        dso, dh_res = _out_bwd(da, w, p + "attention.output.", lc["sa_o"], g)
        dxq, dxkv = _attn_bwd(dso, w, p + "attention.attention.", nh, lc["sa"], g)
        dh = dh_res + dxq + dxkv
    if "emb" in c["keeps"]:
        dh = dh * c["keeps"]["emb"]
    dq0, dg, db = layer_norm_bwd(dh, c["emb"], w["qformer.layernorm.weight"])
    g["qformer.layernorm.weight"], g["qformer.layernorm.bias"] = dg, db
    g["query"] = dq0.sum(0, keepdims=True)
    if "encoder_proj.weight" in w:
        S = c["x_in"].shape[1]
        de = denc.reshape(B, -1, denc.shape[-1])[:, :S]
        g["encoder_proj.weight"] = f(de).T @ f(c["x_in"])
    return {k: np.asarray(v, np.float32) for k, v in g.items()}
