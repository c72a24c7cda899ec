"""MOSA projector on MI355X.

Drop-in for ``MOSAProjector`` (tiny_audio/projectors.py:101-182): Conv1d x2 (k3, s2, p1) + GELU downsampler, ReLU router
-> softmax over 4 dense experts (SimpleAdapter: Linear -> GELU -> Linear), same parameter names
(``downsampler.{0,2}.*``, ``router.{0,2}.*``, ``experts.N.fc{1,2}.*``).  The convolutions are ``ta_gemm_bf16_nt`` with
affine row maps over zero-padded time-major buffers (no im2col, as the encoder's conv front-end), the mixture is
``ta_mix_fwd/bwd``; forward and backward are hand-written over the C ABI, every parameter is trainable, no gradient
flows to the frozen encoder output.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .ops import BF16, F32
from .qformer_projector import _Lin, _pad64


def _conv_len(n):
    return (n + 2 - 3) // 2 + 1


def _wmat(w):
    """Conv1d weight [out, in, 3] -> GEMM weight [out, 3*in] (column = tap*in + c), bf16."""
    return ops.cast_bf16(w.detach().permute(0, 2, 1).reshape(w.shape[0], -1).contiguous())


class _MOSAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, *params):
        P = dict(zip(mod._names, params))
        B, S, E = x.shape
        D, dev = mod.llm_dim, x.device
        T1, T2 = _conv_len(S), _conv_len(_conv_len(S))
        M1, M = B * T1, B * T2
        xb = x.detach()
        xb = (xb if xb.dtype == BF16 else xb.to(BF16))
        xp = torch.zeros((B, S + 2, E), device=dev, dtype=BF16)              # conv padding = 1
        xp[:, 1:S + 1] = xb
        # conv1 (+bias) -> pre-activation written INTO the padded buffer of conv2's input; GELU(0) = 0 keeps the pads
        h1p = torch.zeros((B, T1 + 2, E), device=dev, dtype=BF16)
        ops.gemm_nt(xp, _wmat(P["downsampler.0.weight"]), M=M1, N=E, K=3 * E, out=h1p, bias=P["downsampler.0.bias"].detach(),
                    a_map=(2 * E, T1, (S + 2) * E), c_map=(E, T1, (T1 + 2) * E, E))
        a1p = ops.gelu_fwd(h1p)
        w2m = _wmat(P["downsampler.2.weight"])
        h2 = ops.gemm_nt(a1p, w2m, M=M, N=D, K=3 * E, bias=P["downsampler.2.bias"].detach(), a_map=(2 * E, T2, (T1 + 2) * E),
                         out_dtype=BF16)
        x2 = ops.gelu_fwd(h2)
        T = {"lins": {}}

        def lin(name):
            l = _Lin(P[name + ".weight"], P[name + ".bias"], name)
            T["lins"][name] = l
            return l

        r1 = lin("router.0").fwd(x2, BF16)
        r1a = ops.relu_fwd(r1)
        logits = lin("router.2").fwd(r1a, F32)
        nE = mod.num_experts
        o = torch.empty((nE, M, D), device=dev, dtype=F32)
        T["eh"] = []
        for e in range(nE):
            h = lin(f"experts.{e}.fc1").fwd(x2, BF16)
            T["eh"].append(h)
            l2 = lin(f"experts.{e}.fc2")
            l2.xb = ops.gelu_fwd(h)
            ops.gemm_nt(l2.xb, l2.wb, bias=l2.b.detach(), out=o[e])
        rw, out = ops.mix_fwd(logits, o)
        T.update(xp=xp, h1p=h1p, a1p=a1p, h2=h2, x2=x2, r1=r1, o=o, rw=rw, w2=P["downsampler.2.weight"].detach())
        ctx.mod, ctx.T, ctx.dims = mod, T, (B, S, E, T1, T2)
        return out.reshape(B, T2, D)

    @staticmethod
    def backward(ctx, dy):
        mod, T = ctx.mod, ctx.T
        B, S, E, T1, T2 = ctx.dims
        D, dev, L = mod.llm_dim, dy.device, T["lins"]
        M1, M = B * T1, B * T2
        g = {}
        dob, dlg = ops.mix_bwd(dy.to(F32).reshape(M, D).contiguous(), T["o"], T["rw"])
        dx2 = torch.zeros((M, D), device=dev, dtype=F32)
        for e in range(mod.num_experts):
            da = L[f"experts.{e}.fc2"].bwd(dob[e], g)
            dh = ops.gelu_bwd(da, T["eh"][e])
            dx2 += L[f"experts.{e}.fc1"].bwd(dh, g).to(F32)
        dr1a = L["router.2"].bwd(ops.cast_bf16(dlg), g)[:, : T["r1"].shape[1]].contiguous()
        dx2 += L["router.0"].bwd(ops.relu_bwd(dr1a, T["r1"]), g).to(F32)
        # conv2: dW from the row-mapped (im2col) view of a1p, dX scattered tap by tap into the padded gradient buffer
        dh2 = ops.gelu_bwd(ops.cast_bf16(dx2), T["h2"])
        Mp = _pad64(M)
        colT = ops.transpose_to_bf16(T["a1p"].reshape(-1, E), ld_out=Mp, in_map=(2 * E, (T1 + 2) * E, T2), rows=M, cols=3 * E)
        dW = ops.gemm_nt(ops.transpose_to_bf16(dh2, ld_out=Mp), colT, out_dtype=F32, splits=4)
        g["downsampler.2.weight"] = dW.reshape(D, 3, E).permute(0, 2, 1).contiguous()
        g["downsampler.2.bias"] = ops.colsum(dh2)
        da1p = torch.zeros((B, T1 + 2, E), device=dev, dtype=F32)
        for tap in range(3):
            wt = ops.cast_bf16(T["w2"][:, :, tap].t().contiguous())           # [E_in, D_out]: d a1p[b, 2t+tap] += dh2[b,t] W[:, :, tap]
            ops.gemm_nt(dh2, wt, M=M, N=E, K=D, out=da1p, residual=da1p, c_map=(2 * E, T2, (T1 + 2) * E, tap * E))
        dh1p = ops.gelu_bwd(ops.cast_bf16(da1p), T["h1p"])
        dh1 = dh1p[:, 1:T1 + 1].reshape(M1, E).contiguous()                    # the pad rows are not conv-1 outputs
        Mp1 = _pad64(M1)
        colT = ops.transpose_to_bf16(T["xp"].reshape(-1, E), ld_out=Mp1, in_map=(2 * E, (S + 2) * E, T1), rows=M1, cols=3 * E)
        dW = ops.gemm_nt(ops.transpose_to_bf16(dh1, ld_out=Mp1), colT, out_dtype=F32, splits=4)
        g["downsampler.0.weight"] = dW.reshape(E, 3, E).permute(0, 2, 1).contiguous()
        g["downsampler.0.bias"] = ops.colsum(dh1)
        ctx.T = None
        return (None, None) + tuple(g[n].reshape(p.shape) for n, p in zip(mod._names, mod.parameters()))


class SimpleAdapter(nn.Module):
    """Simple 2-layer GELU adapter (MOSA paper): parameters only, the arithmetic runs in _MOSAFn."""

    def __init__(self, input_dim, hidden_dim, output_dim):
        super().__init__()
        self.fc1 = nn.Linear(input_dim, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, output_dim)


class MOSAProjector(nn.Module):
    ADAPTER_HIDDEN_DIM = 4096
    ROUTER_HIDDEN_DIM = 512

    def __init__(self, config):
        super().__init__()
        self.encoder_dim = getattr(config, "encoder_dim", None) or 1280
        self.llm_dim = getattr(config, "llm_dim", None) or 2048
        self.num_experts = getattr(config, "num_experts", None) or 4
        E, D = self.encoder_dim, self.llm_dim
        if E % 64 or D % 64:
            raise ValueError("encoder_dim and llm_dim must be multiples of 64")
        self.downsampler = nn.Sequential(nn.Conv1d(E, E, 3, 2, 1), nn.GELU(), nn.Conv1d(E, D, 3, 2, 1), nn.GELU())
        self.router = nn.Sequential(nn.Linear(D, self.ROUTER_HIDDEN_DIM), nn.ReLU(), nn.Linear(self.ROUTER_HIDDEN_DIM, self.num_experts))
        self.experts = nn.ModuleList([SimpleAdapter(D, self.ADAPTER_HIDDEN_DIM, D) for _ in range(self.num_experts)])
        self._names = [n for n, _ in self.named_parameters()]

    def get_output_length(self, input_length):
        return _conv_len(_conv_len(input_length))

    def forward(self, x):
        """x [B, S, encoder_dim] -> [B, out_len, llm_dim] (fp32)"""
        return _MOSAFn.apply(x, self, *[p for _, p in self.named_parameters()])
