"""MOSA projector, numpy.  TEST INFRASTRUCTURE ONLY.

tiny_audio/projectors.py:88-182 (MOSAProjector, arXiv:2508.18998): Conv1d(E->E, k3, s2, p1) + GELU, Conv1d(E->D, k3,
s2, p1) + GELU; router Linear(D,512) + ReLU + Linear(512, n_experts) -> softmax (dense mixture, no aux loss); experts
SimpleAdapter = Linear(D,4096) + GELU + Linear(4096, D); output = sum_e softmax_e * expert_e(x).
Hand-written backward for every parameter (no gradient w.r.t. the frozen encoder output).
"""
from __future__ import annotations

import numpy as np

from .projectors import gelu, gelu_grad


def conv_out_len(n):
    return (n + 2 - 3) // 2 + 1


def output_length(S):
    return conv_out_len(conv_out_len(S))


def _im2col(x):
    """x [B, T, C] -> [B, T_out, 3*C] for kernel 3, stride 2, padding 1 (column = tap*C + c)."""
    B, T, C = x.shape
    xp = np.zeros((B, T + 2, C), x.dtype); xp[:, 1:T + 1] = x
    To = conv_out_len(T)
    idx = 2 * np.arange(To)[:, None] + np.arange(3)[None, :]
    return xp[:, idx].reshape(B, To, 3 * C)


def _col2im(dcol, T):
    B, To, K = dcol.shape
    C = K // 3
    dxp = np.zeros((B, T + 2, C), dcol.dtype)
    d = dcol.reshape(B, To, 3, C)
    for tap in range(3):
        dxp[:, 2 * np.arange(To) + tap] += d[:, :, tap]
    return dxp[:, 1:T + 1]


def _wmat(w):                       # Conv1d weight [out, in, 3] -> [out, 3*in] matching _im2col's column order
    return w.transpose(0, 2, 1).reshape(w.shape[0], -1)


def mosa_forward(hs, w):
    x = hs.astype(np.float32)
    c = {}
    c["col1"] = _im2col(x)
    c["h1"] = c["col1"] @ _wmat(w["downsampler.0.weight"]).T + w["downsampler.0.bias"]
    a1 = gelu(c["h1"])
    c["col2"] = _im2col(a1)
    c["h2"] = c["col2"] @ _wmat(w["downsampler.2.weight"]).T + w["downsampler.2.bias"]
    x2 = gelu(c["h2"])
    c["x2"] = x2
    c["r1"] = x2 @ w["router.0.weight"].T + w["router.0.bias"]
    c["r1a"] = np.maximum(c["r1"], 0)
    lg = c["r1a"] @ w["router.2.weight"].T + w["router.2.bias"]
    lg = lg - lg.max(-1, keepdims=True)
    rw = np.exp(lg); rw /= rw.sum(-1, keepdims=True)
    c["rw"] = rw
    E = rw.shape[-1]
    out = 0
    c["eh"], c["eo"] = [], []
    for e in range(E):
        h = x2 @ w[f"experts.{e}.fc1.weight"].T + w[f"experts.{e}.fc1.bias"]
        o = gelu(h) @ w[f"experts.{e}.fc2.weight"].T + w[f"experts.{e}.fc2.bias"]
        c["eh"].append(h); c["eo"].append(o)
        out = out + o * rw[..., e:e + 1]
    c["T"] = hs.shape[1]
    return out.astype(np.float32), c


def mosa_backward(dy, w, c):
    g = {}
    f = lambda t: t.reshape(-1, t.shape[-1])
    dy = dy.astype(np.float32)
    rw, x2 = c["rw"], c["x2"]
    E = rw.shape[-1]
    dx2 = np.zeros_like(x2)
    drw = np.zeros_like(rw)
    for e in range(E):
        do = dy * rw[..., e:e + 1]
        drw[..., e] = (dy * c["eo"][e]).sum(-1)
        a = gelu(c["eh"][e])
        g[f"experts.{e}.fc2.weight"] = f(do).T @ f(a); g[f"experts.{e}.fc2.bias"] = f(do).sum(0)
        dh = (do @ w[f"experts.{e}.fc2.weight"]) * gelu_grad(c["eh"][e])
        g[f"experts.{e}.fc1.weight"] = f(dh).T @ f(x2); g[f"experts.{e}.fc1.bias"] = f(dh).sum(0)
        dx2 += dh @ w[f"experts.{e}.fc1.weight"]
    dlg = rw * (drw - (drw * rw).sum(-1, keepdims=True))
    g["router.2.weight"] = f(dlg).T @ f(c["r1a"]); g["router.2.bias"] = f(dlg).sum(0)
    dr1 = (dlg @ w["router.2.weight"]) * (c["r1"] > 0)
    g["router.0.weight"] = f(dr1).T @ f(x2); g["router.0.bias"] = f(dr1).sum(0)
    dx2 += dr1 @ w["router.0.weight"]
    dh2 = dx2 * gelu_grad(c["h2"])
    dW = f(dh2).T @ f(c["col2"])
    D, K = dW.shape
    g["downsampler.2.weight"] = dW.reshape(D, 3, K // 3).transpose(0, 2, 1); g["downsampler.2.bias"] = f(dh2).sum(0)
    da1 = _col2im(dh2 @ _wmat(w["downsampler.2.weight"]), c["h1"].shape[1])
    dh1 = da1 * gelu_grad(c["h1"])
    dW = f(dh1).T @ f(c["col1"])
    Eo, K = dW.shape
    g["downsampler.0.weight"] = dW.reshape(Eo, 3, K // 3).transpose(0, 2, 1); g["downsampler.0.bias"] = f(dh1).sum(0)
    return {k: np.ascontiguousarray(v, np.float32) for k, v in g.items()}
