"""QFormer audio projector on MI355X (north_star "MLP/MoE/QFormer projectors").

Drop-in for ``QFormerAudioProjector`` (tiny_audio/projectors.py:359-475): the same constructor arguments, the same
parameter names (``query``, ``qformer.layernorm.*``, ``qformer.encoder.layer.N.{attention,crossattention}.*``,
``...intermediate_query.*``, ``...output_query.*``, ``linear.*``), ``forward([B,S,E]) -> [B, nblocks*3, llm_dim]``,
``get_output_length``.  The arithmetic is Blip2QFormerModel's (TF:models/blip_2/modeling_blip_2.py), restated as a
hand-written forward/backward over the ta355 C ABI: every linear is ``ta_gemm_bf16_nt`` (bf16 operands, fp32
accumulate, fp32 masters), LayerNorm / softmax / residuals / dropout masks are fp32 kernels of csrc/nn_prims.hip.
All parameters are trainable, so backward produces dW (transpose + split-K GEMM), db (column sums), LayerNorm affine
gradients and d(query); no gradient flows to the frozen encoder output.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import ops
from .ops import BF16, F32


def _pad64(n):
    return (n + 63) // 64 * 64


class _Lin:
    """y = x W^T + b with saved operands; backward returns dx (bf16) and fills grads[name.weight / name.bias]."""

    def __init__(self, w, b, name):
        self.w, self.b, self.name = w, b, name
        self.wb = ops.cast_bf16(w.detach().contiguous())

    def fwd(self, xb, out_dtype):
        self.xb = xb
        return ops.gemm_nt(xb, self.wb, bias=None if self.b is None else self.b.detach(), out_dtype=out_dtype)

    def bwd(self, dyb, grads, need_dx=True):
        M = dyb.shape[0]
        Mp = _pad64(M)
        dyT, xT = ops.transpose_to_bf16(dyb, ld_out=Mp), ops.transpose_to_bf16(self.xb, ld_out=Mp)
        N, K = self.w.shape
        splits = 1
        while (N // 128 + 1) * (K // 128 + 1) * splits < 512 and Mp // 64 // (2 * splits) >= 4 and splits < 32:
            splits *= 2
        grads[self.name + ".weight"] = ops.gemm_nt(dyT, xT, out_dtype=F32, splits=splits)
        if self.b is not None:
            grads[self.name + ".bias"] = ops.colsum(dyb)
        if not need_dx:
            return None
        wT = ops.transpose_to_bf16(self.wb, ld_out=_pad64(N))                # [K, Np]: dx = dy W  as an NT GEMM
        if _pad64(N) != N:
            dyb = torch.nn.functional.pad(dyb, (0, _pad64(N) - N))
        return ops.gemm_nt(dyb.contiguous(), wT, out_dtype=BF16)


class _QFormerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, keeps, *params):
        P = dict(zip(mod._names, params))
        B, S, E = x.shape
        dev = x.device
        win, nq, nh, H, eps = mod.window_size, mod.num_queries, mod.num_heads, mod.hidden, mod.eps
        nb = math.ceil(S / win)
        EB, M, Me = B * nb, B * nb * nq, B * nb * win
        keeps = keeps or {}
        def kp(k):                      # hidden-state masks arrive as [EB, nq, H] or [M, H]; probability masks [EB, nh, Lq, Lk]
            if keeps.get(k) is None:
                return None
            t = keeps[k].to(device=dev, dtype=F32).contiguous()
            return t if k.endswith("_p") else t.reshape(M, H)
        xb = x.detach()
        xb = (xb if xb.dtype == BF16 else xb.to(BF16)).contiguous()
        T = {"lins": {}}

        def lin(name, bias=True):
            l = _Lin(P[name + ".weight"], P[name + ".bias"] if bias else None, name)
            T["lins"][name] = l
            return l

        if mod.encoder_proj is not None:
            xb = lin("encoder_proj", bias=False).fwd(xb.reshape(B * S, E), BF16).reshape(B, S, H)
        enc = torch.zeros((B, nb * win, H), device=dev, dtype=BF16)              # zero padded last window (:447-451)
        enc[:, :S] = xb
        enc = enc.reshape(Me, H)
        q0 = P["query"].detach().to(F32).reshape(nq, H).contiguous()
        # embeddings: LayerNorm(query) (+ dropout); identical for every window -> res_rows broadcast is not needed, tile it
        z0 = q0.repeat(EB, 1)
        h, hb, T["emb_xhat"], T["emb_rstd"] = ops.layernorm_res_fwd(z0, P["qformer.layernorm.weight"].detach(),
                                                                    P["qformer.layernorm.bias"].detach(), eps)
        if kp("emb") is not None:
            h = h * kp("emb"); hb = h.to(BF16)
        scale = float((H // nh) ** -0.5)
        T["layers"] = []
        for i in range(mod.num_layers):
            p = f"qformer.encoder.layer.{i}."
            Lc = {}

            def attn(pre, xq_b, xkv_b, Lk, keep_p, Lc=Lc):
                q = lin(pre + "attention.query").fwd(xq_b, BF16)
                k = lin(pre + "attention.key").fwd(xkv_b, BF16)
                v = lin(pre + "attention.value").fwd(xkv_b, BF16)
                o, pr = ops.attn_small_fwd(q, k, v, EB, nh, nq, Lk, scale, keep_p)
                Lc[pre] = (q, k, v, pr, keep_p, Lk)
                return o

            def out(pre, hin_b, res, keep, Lc=Lc):
                z = lin(pre + "dense").fwd(hin_b, F32)
                y, yb, xh, rs = ops.layernorm_res_fwd(z, P[pre + "LayerNorm.weight"].detach(), P[pre + "LayerNorm.bias"].detach(),
                                                      eps, res=res, keep=keep)
                Lc[pre + "ln"] = (xh, rs, keep)
                return y, yb

            o = attn(p + "attention.", hb, hb, nq, kp(f"l{i}.sa_p"))
            a, ab = out(p + "attention.output.", o, h, kp(f"l{i}.sa"))
            o = attn(p + "crossattention.", ab, enc, win, kp(f"l{i}.ca_p"))
            c, cb = out(p + "crossattention.output.", o, a, kp(f"l{i}.ca"))
            hi = lin(p + "intermediate_query.dense").fwd(cb, BF16)
            Lc["hi"] = hi
            ai = ops.gelu_fwd(hi)
            h, hb = out(p + "output_query.", ai, c, kp(f"l{i}.ffn"))
            T["layers"].append(Lc)
        y = lin("linear").fwd(hb, F32)
        ctx.mod, ctx.T, ctx.dims, ctx.keep_emb = mod, T, (B, S, nb, EB, M, Me), kp("emb")
        ctx.P = {k: v.detach() for k, v in P.items()}
        return y.reshape(B, nb * nq, -1)

    @staticmethod
    def backward(ctx, dy):
        mod, T, P = ctx.mod, ctx.T, ctx.P
        B, S, nb, EB, M, Me = ctx.dims
        win, nq, nh, H = mod.window_size, mod.num_queries, mod.num_heads, mod.hidden
        scale = float((H // nh) ** -0.5)
        dev = dy.device
        g = {}
        L = T["lins"]
        dyb = ops.cast_bf16(dy.to(F32).reshape(M, -1).contiguous())
        dh_b = L["linear"].bwd(dyb, g)                                    # bf16 [M, H]
        dh = dh_b.to(F32)
        denc_needed = mod.encoder_proj is not None
        denc = torch.zeros((Me, H), device=dev, dtype=F32) if denc_needed else None

        def out_bwd(pre, dy_f32, Lc):
            xh, rs, keep = Lc[pre + "ln"]
            g[pre + "LayerNorm.weight"] = torch.zeros(H, device=dev, dtype=F32)
            g[pre + "LayerNorm.bias"] = torch.zeros(H, device=dev, dtype=F32)
            du, dz = ops.layernorm_bwd(dy_f32.contiguous(), xh, rs, P[pre + "LayerNorm.weight"], g[pre + "LayerNorm.weight"],
                                       g[pre + "LayerNorm.bias"], keep=keep)
            return L[pre + "dense"].bwd(dz, g), du                        # (d dense input bf16, d residual f32)

        def attn_bwd(pre, do_b, Lc, kv_is_enc):
            q, k, v, pr, keep_p, Lk = Lc[pre]
            dq, dk, dv = ops.attn_small_bwd(do_b, q, k, v, pr, EB, nh, nq, Lk, scale, keep_p)
            dxq = L[pre + "attention.query"].bwd(dq, g)
            need = (not kv_is_enc) or denc_needed
            dxk = L[pre + "attention.key"].bwd(dk, g, need_dx=need)
            dxv = L[pre + "attention.value"].bwd(dv, g, need_dx=need)
            dxkv = None if not need else dxk.to(F32) + dxv.to(F32)
            return dxq.to(F32), dxkv

        for i in reversed(range(mod.num_layers)):
            p = f"qformer.encoder.layer.{i}."
            Lc = T["layers"][i]
            dai, dc = out_bwd(p + "output_query.", dh, Lc)
            dhi = ops.gelu_bwd(dai, Lc["hi"])
            dc = dc + L[p + "intermediate_query.dense"].bwd(dhi, g).to(F32)
            dco, da = out_bwd(p + "crossattention.output.", dc, Lc)
            dxq, dxkv = attn_bwd(p + "crossattention.", dco, Lc, True)
            da = da + dxq
            if denc_needed:
                denc += dxkv
            dso, dh_res = out_bwd(p + "attention.output.", da, Lc)
            dxq, dxkv = attn_bwd(p + "attention.", dso, Lc, False)
            dh = dh_res + dxq + dxkv
        if ctx.keep_emb is not None:
            dh = dh * ctx.keep_emb
        g["qformer.layernorm.weight"] = torch.zeros(H, device=dev, dtype=F32)
        g["qformer.layernorm.bias"] = torch.zeros(H, device=dev, dtype=F32)
        dq0, _ = ops.layernorm_bwd(dh.contiguous(), T["emb_xhat"], T["emb_rstd"], P["qformer.layernorm.weight"],
                                   g["qformer.layernorm.weight"], g["qformer.layernorm.bias"], want_dz=False)
        g["query"] = ops.colsum(dq0.reshape(EB, nq * H)).reshape(1, nq, H)
        if denc_needed:
            de = ops.cast_bf16(denc.reshape(B, nb * win, H)[:, :S].reshape(B * S, H).contiguous())
            L["encoder_proj"].bwd(de, g, need_dx=False)
        ctx.T = None
        return (None, None, None) + tuple(g[n].reshape(P[n].shape) for n in mod._names)


def _mk(shape, std, gen, dev):
    return nn.Parameter(torch.randn(shape, generator=gen, dtype=F32).mul_(std).to(dev))


class _Holder(nn.Module):
    pass


class QFormerAudioProjector(nn.Module):
    """BLIP-2 QFormer projector with learnable queries (Granite-style windows of 15 frames -> 3 queries)."""

    def __init__(self, config):
        super().__init__()
        E, D = config.encoder_dim, config.llm_dim
        self.window_size = getattr(config, "qformer_window_size", 15)
        self.downsample_rate = getattr(config, "downsample_rate", 5)
        self.num_queries = self.window_size // self.downsample_rate
        self.hidden = H = getattr(config, "qformer_hidden_size", None) or E
        self.num_layers = getattr(config, "qformer_num_layers", 2)
        self.num_heads = getattr(config, "qformer_num_heads", 16)
        F_ = getattr(config, "qformer_intermediate_size", None) or 4 * H
        self.eps = 1e-12
        self.hidden_dropout = self.attn_dropout = 0.1                         # Granite config (projectors.py:409-411)
        self.encoder_dim, self.llm_dim = E, D
        if H % self.num_heads or (H // self.num_heads) % 8:
            raise ValueError("qformer hidden size must split into heads of a multiple of 8")
        gen = torch.Generator(device="cpu"); gen.manual_seed(1234)
        dev = "cpu"
        self.query = _mk((1, self.num_queries, H), 1.0, gen, dev)                 # Granite: std 1.0
        self.encoder_proj = None
        if E != H:
            self.encoder_proj = nn.Linear(E, H, bias=False)

        def linear(i, o):
            m = _Holder(); m.weight = _mk((o, i), 0.02, gen, dev); m.bias = nn.Parameter(torch.zeros(o)); return m

        def lnorm():
            m = _Holder(); m.weight = nn.Parameter(torch.ones(H)); m.bias = nn.Parameter(torch.zeros(H)); return m

        def attention():
            a = _Holder(); a.attention = _Holder()
            a.attention.query, a.attention.key, a.attention.value = linear(H, H), linear(H, H), linear(H, H)
            a.output = _Holder(); a.output.dense, a.output.LayerNorm = linear(H, H), lnorm()
            return a

        self.qformer = _Holder()
        self.qformer.layernorm = lnorm()
        self.qformer.encoder = _Holder()
        layers = []
        for _ in range(self.num_layers):
            l = _Holder()
            l.attention, l.crossattention = attention(), attention()
            l.intermediate_query = _Holder(); l.intermediate_query.dense = linear(H, F_)
            l.output_query = _Holder(); l.output_query.dense, l.output_query.LayerNorm = linear(F_, H), lnorm()
            layers.append(l)
        self.qformer.encoder.layer = nn.ModuleList(layers)
        self.linear = nn.Linear(H, D)
        self._names = [n for n, _ in self.named_parameters()]

    def get_output_length(self, input_length):
        nblocks = (input_length + self.window_size - 1) // self.window_size
        return nblocks * self.num_queries

    def _dropout_masks(self, x):
        """Train-mode keep masks (already divided by 1 - p), one per dropout site of Blip2QFormerModel."""
        if not self.training or (self.hidden_dropout <= 0 and self.attn_dropout <= 0):
            return None
        B, S, _ = x.shape
        nb = math.ceil(S / self.window_size)
        EB, M, H, nh, nq, win = B * nb, B * nb * self.num_queries, self.hidden, self.num_heads, self.num_queries, self.window_size
        self._seed = getattr(self, "_seed", 0x51F0) + 1
        ks = {}

        def mk(key, shape, p, salt):
            n = 1
            for s in shape:
                n *= s
            ks[key] = (ops.bernoulli_keep(n, 1.0 - p, self._seed * 64 + salt, x.device) / (1.0 - p)).reshape(shape)
        mk("emb", (M, H), self.hidden_dropout, 0)
        for i in range(self.num_layers):
            mk(f"l{i}.sa", (M, H), self.hidden_dropout, 1 + 8 * i); mk(f"l{i}.ca", (M, H), self.hidden_dropout, 2 + 8 * i)
            mk(f"l{i}.ffn", (M, H), self.hidden_dropout, 3 + 8 * i)
            mk(f"l{i}.sa_p", (EB, nh, nq, nq), self.attn_dropout, 4 + 8 * i); mk(f"l{i}.ca_p", (EB, nh, nq, win), self.attn_dropout, 5 + 8 * i)
        return ks

    def forward(self, hidden_states, keeps=None):
        """hidden_states [B, S, encoder_dim] -> [B, nblocks * num_queries, llm_dim] (fp32).  ``keeps`` injects dropout
        keep masks (tests); otherwise they are drawn in training mode and absent in eval mode."""
        if keeps is None:
            keeps = self._dropout_masks(hidden_states)
        params = [p for _, p in self.named_parameters()]
        return _QFormerFn.apply(hidden_states, self, keeps, *params)
