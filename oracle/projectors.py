"""Audio projectors (rows a5/a6 of SURVEY.md section 8), numpy forward + backward.

MLP:  tiny_audio/projectors.py:23-71   (MLPAudioProjector)
MoE:  tiny_audio/projectors.py:185-351 (MoEAudioProjector, _forward_sparse)
RMSNorm is ``LlamaRMSNorm`` (TF:models/llama/modeling_llama.py: x * rsqrt(mean(x^2)+eps) * w, eps 1e-6).
"""
from __future__ import annotations

import numpy as np
from scipy.special import erf

from .encoder import gelu


def gelu_grad(x):
    return (0.5 * (1.0 + erf(x / np.sqrt(2.0))) + x * np.exp(-0.5 * x * x) / np.sqrt(2.0 * np.pi)).astype(x.dtype)


def frame_stack(x, k):
    """tiny_audio/projectors.py:79-87: drop the tail, view [B, S//k, k*dim]."""
    B, S, D = x.shape
    n = (S - k) // k + 1
    return np.ascontiguousarray(x[:, : n * k, :]).reshape(B, n, D * k)


def rms_norm(x, w, eps=1e-6):
    r = 1.0 / np.sqrt((x.astype(np.float32) ** 2).mean(-1, keepdims=True) + eps)
    return (x * r * w).astype(np.float32), r.astype(np.float32)


def rms_norm_bwd(dy, x, r, w):
    """Returns (dx, dw) for y = w * (x * r)."""
    xh = x * r
    dw = (dy * xh).reshape(-1, x.shape[-1]).sum(0)
    dn = dy * w
    dx = r * (dn - xh * (dn * xh).mean(-1, keepdims=True))
    return dx.astype(np.float32), dw.astype(np.float32)


# ----------------------------------------------------------------------------- MLP
def mlp_forward(x, w, k=4, eps=1e-6):
    """MLPAudioProjector.forward tiny_audio/projectors.py:57-71.
    x [B, S, enc_dim] -> y [B, N, llm_dim]; cache for backward."""
    xs = frame_stack(x.astype(np.float32), k)
    h1 = xs @ w["linear_1.weight"].T
    n1, r1 = rms_norm(h1, w["norm.weight"], eps)
    a1 = gelu(n1)
    h2 = a1 @ w["linear_2.weight"].T
    y, r2 = rms_norm(h2, w["norm_2.weight"], eps)
    return y, dict(xs=xs, h1=h1, r1=r1, n1=n1, a1=a1, h2=h2, r2=r2)


def mlp_backward(dy, w, c, need_dx=False):
    dh2, dg2 = rms_norm_bwd(dy.astype(np.float32), c["h2"], c["r2"], w["norm_2.weight"])
    H = c["a1"].shape[-1]
    D = dh2.shape[-1]
    dW2 = dh2.reshape(-1, D).T @ c["a1"].reshape(-1, H)
    da1 = dh2 @ w["linear_2.weight"]
    dn1 = da1 * gelu_grad(c["n1"])
    dh1, dg1 = rms_norm_bwd(dn1, c["h1"], c["r1"], w["norm.weight"])
    dW1 = dh1.reshape(-1, H).T @ c["xs"].reshape(-1, c["xs"].shape[-1])
    grads = {"linear_1.weight": dW1.astype(np.float32), "norm.weight": dg1,
             "linear_2.weight": dW2.astype(np.float32), "norm_2.weight": dg2}
    if need_dx:
        grads["_dx_stacked"] = dh1 @ w["linear_1.weight"]
    return grads


# ----------------------------------------------------------------------------- MoE
def _adapter_fwd(x, w, p):
    h = x @ w[p + "fc1.weight"].T + w[p + "fc1.bias"]
    a = gelu(h)
    return a @ w[p + "fc2.weight"].T + w[p + "fc2.bias"], (h, a)


def _adapter_bwd(dy, x, w, p, cache, grads):
    h, a = cache
    grads[p + "fc2.weight"] = grads.get(p + "fc2.weight", 0) + dy.T @ a
    grads[p + "fc2.bias"] = grads.get(p + "fc2.bias", 0) + dy.sum(0)
    dh = (dy @ w[p + "fc2.weight"]) * gelu_grad(h)
    grads[p + "fc1.weight"] = grads.get(p + "fc1.weight", 0) + dh.T @ x
    grads[p + "fc1.bias"] = grads.get(p + "fc1.bias", 0) + dh.sum(0)
    return dh @ w[p + "fc1.weight"]


def moe_forward(x, w, k=4, num_experts=4, top_k=2, training=False, jitter_noise=None,
                aux_coef=0.01, z_coef=1e-4, eps=1e-6):
    """MoEAudioProjector.forward / _forward_sparse tiny_audio/projectors.py:257-347.
    ``jitter_noise`` [T, E] is the injected multiplicative noise (the reference
    draws U(1-0.01, 1+0.01) when training, :294-300); None means no jitter."""
    xs = frame_stack(x.astype(np.float32), k)
    B, N, In = xs.shape
    xn, r = rms_norm(xs, w["norm.weight"], eps)
    flat = xn.reshape(-1, In)
    out, sh_cache = _adapter_fwd(flat, w, "shared_expert.")
    logits = flat @ w["router.weight"].T
    if training and jitter_noise is not None:
        logits = logits * jitter_noise.astype(np.float32)
    m = logits.max(-1, keepdims=True)
    e = np.exp(logits - m)
    probs = e / e.sum(-1, keepdims=True)
    order = np.argsort(-probs, axis=-1, kind="stable")[:, :top_k]
    topw_raw = np.take_along_axis(probs, order, axis=-1)
    denom = topw_raw.sum(-1, keepdims=True) + 1e-6
    topw = topw_raw / denom
    aux = np.float32(0.0)
    lse = (m[:, 0] + np.log(e.sum(-1)))
    if training:
        pbar = probs.mean(0)
        balance = aux_coef * ((pbar - 1.0 / num_experts) ** 2).mean() * num_experts
        z = z_coef * (lse ** 2).mean()
        aux = np.float32(balance + z)
    out = out.copy()
    ex_cache = {}
    for ei in range(num_experts):
        tok, kk = np.where(order == ei)
        if tok.size == 0:
            continue
        y, c = _adapter_fwd(flat[tok], w, f"experts.{ei}.")
        out[tok] += y * topw[tok, kk][:, None]
        ex_cache[ei] = (tok, kk, y, c)
    cache = dict(xs=xs, r=r, flat=flat, sh=sh_cache, logits=logits, probs=probs, order=order,
                 topw_raw=topw_raw, denom=denom, topw=topw, ex=ex_cache, lse=lse,
                 noise=jitter_noise if (training and jitter_noise is not None) else None,
                 training=training, aux_coef=aux_coef, z_coef=z_coef, E=num_experts)
    return out.reshape(B, N, -1).astype(np.float32), aux, cache


def moe_backward(dy, w, c, d_aux=1.0):
    """Gradients of (sum(dy*out) + d_aux*aux) w.r.t. every projector parameter."""
    B, N, D = dy.shape
    T = B * N
    dflat_out = dy.reshape(T, D).astype(np.float32)
    flat = c["flat"]
    grads = {}
    dflat = _adapter_bwd(dflat_out, flat, w, "shared_expert.", c["sh"], grads)
    E = c["E"]
    dtopw = np.zeros_like(c["topw"])
    for ei in range(E):
        p = f"experts.{ei}."
        if ei not in c["ex"]:
            for s in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
                grads[p + s] = np.zeros_like(w[p + s])
            continue
        tok, kk, y, cache = c["ex"][ei]
        wt = c["topw"][tok, kk][:, None]
        dtopw[tok, kk] = (dflat_out[tok] * y).sum(-1)
        dx_e = _adapter_bwd(dflat_out[tok] * wt, flat[tok], w, p, cache, grads)
        np.add.at(dflat, tok, dx_e)
    # top-k renormalisation: w_k = p_k / (sum_j p_j + 1e-6)
    denom = c["denom"]
    draw = dtopw / denom - (dtopw * c["topw_raw"]).sum(-1, keepdims=True) / (denom ** 2)
    dprobs = np.zeros_like(c["probs"])
    np.put_along_axis(dprobs, c["order"], draw, axis=-1)
    dlogits = np.zeros_like(c["logits"])
    if c["training"]:
        pbar = c["probs"].mean(0)
        dpbar = c["aux_coef"] * E * 2.0 * (pbar - 1.0 / E) / E
        dprobs = dprobs + d_aux * dpbar[None, :] / T
        dlogits = dlogits + d_aux * c["z_coef"] * 2.0 * c["lse"][:, None] * c["probs"] / T
    dlogits = dlogits + c["probs"] * (dprobs - (dprobs * c["probs"]).sum(-1, keepdims=True))
    if c["noise"] is not None:
        dlogits = dlogits * c["noise"]
    grads["router.weight"] = (dlogits.T @ flat).astype(np.float32)
    dflat = dflat + dlogits @ w["router.weight"]
    _, dg = rms_norm_bwd(dflat.reshape(c["xs"].shape), c["xs"], c["r"], w["norm.weight"])
    grads["norm.weight"] = dg
    return {k_: np.asarray(v, dtype=np.float32) for k_, v in grads.items()}
