"""Trainable audio projectors on MI355X -- same module tree / parameter names / methods as
``tiny_audio/projectors.py`` so checkpoints (``projector.linear_1.weight`` ...) interchange.

MLPAudioProjector (tiny_audio/projectors.py:23-71): frame-stack k -> Linear(kE->H, no bias) -> RMSNorm ->
erf-GELU -> Linear(H->D, no bias) -> RMSNorm.  Forward and backward are single calls into libta355
(``ta_mlp_projector_forward`` / ``_backward``); fp32 master weights live in ordinary ``nn.Parameter``s and
are re-cast to bf16 (plus a transposed copy of W2) only when the optimizer has changed them.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn

from . import _lib
from .ops import BF16, F32, cast_bf16, ptr, stream, transpose_to_bf16


class _RMSNormWeight(nn.Module):
    """Parameter holder with the reference's name (``LlamaRMSNorm.weight``); arithmetic is in the fused kernels."""

    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.variance_epsilon = eps


class _MLPProjectorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, g1, w2, g2, mod):
        B, S, _ = x.shape
        xb = x.detach()
        if xb.dtype != BF16:
            xb = xb.to(BF16)
        xb = xb.contiguous()
        wts = mod._packed_weights()
        L_ = _lib.lib()
        N = mod.get_output_length(S)
        tape = torch.empty(L_.ta_mlp_tape_bytes(C.byref(wts), B, S), device=x.device, dtype=torch.uint8)
        y = torch.empty((B, N, mod.llm_dim), device=x.device, dtype=F32)
        _lib.check(L_.ta_mlp_projector_forward(C.byref(wts), ptr(xb), B, S, ptr(y), ptr(tape), stream()),
                   "ta_mlp_projector_forward")
        ctx.mod, ctx.xb, ctx.tape, ctx.dims = mod, xb, tape, (B, S)
        return y

    @staticmethod
    def backward(ctx, dy):
        mod, (B, S) = ctx.mod, ctx.dims
        wts = mod._packed_weights()
        L_ = _lib.lib()
        dev = dy.device
        dy = dy.to(F32).contiguous()
        dW1 = torch.empty_like(mod.linear_1.weight, dtype=F32)
        dW2 = torch.empty_like(mod.linear_2.weight, dtype=F32)
        dg1 = torch.empty_like(mod.norm.weight, dtype=F32)
        dg2 = torch.empty_like(mod.norm_2.weight, dtype=F32)
        ws = torch.empty(L_.ta_mlp_bwd_workspace_bytes(C.byref(wts), B, S), device=dev, dtype=torch.uint8)
        _lib.check(L_.ta_mlp_projector_backward(C.byref(wts), ptr(ctx.xb), B, S, ptr(dy), ptr(ctx.tape), ptr(dW1), ptr(dg1),
                                                ptr(dW2), ptr(dg2), ptr(ws), ws.numel(), stream()),
                   "ta_mlp_projector_backward")
        ctx.tape = None
        # the encoder is frozen: no gradient w.r.t. x is ever needed on the training path
        return None, dW1, dg1, dW2, dg2, None


class MLPAudioProjector(nn.Module):
    """2-layer MLP projector with frame-stacking downsampling (tiny_audio/projectors.py:23-71)."""

    def __init__(self, config):
        super().__init__()
        encoder_dim = getattr(config, "encoder_dim", 768)
        llm_dim = getattr(config, "llm_dim", 2048)
        self.k = getattr(config, "projector_pool_stride", 4)
        in_dim = encoder_dim * self.k
        hidden_dim = getattr(config, "projector_hidden_dim", None) or llm_dim
        self.encoder_dim, self.llm_dim, self.hidden_dim = encoder_dim, llm_dim, hidden_dim
        self.linear_1 = nn.Linear(in_dim, hidden_dim, bias=False)      # parameter holders (same names / init
        self.norm = _RMSNormWeight(hidden_dim, eps=1e-6)                # as the reference: kaiming-uniform, ones)
        self.linear_2 = nn.Linear(hidden_dim, llm_dim, bias=False)
        self.norm_2 = _RMSNormWeight(llm_dim, eps=1e-6)
        self._pack = None
        self._pack_versions = None

    def get_output_length(self, input_length):
        """(L - k) // k + 1   (tiny_audio/projectors.py:52-55)."""
        return (input_length - self.k) // self.k + 1

    def _packed_weights(self):
        """bf16 images of the fp32 masters for the kernels, refreshed when a parameter's version changes."""
        ps = (self.linear_1.weight, self.norm.weight, self.linear_2.weight, self.norm_2.weight)
        versions = tuple((p.data_ptr(), p._version) for p in ps)
        if self._pack is None or versions != self._pack_versions:
            w1, g1, w2, g2 = (p.detach().to(F32).contiguous() for p in ps)
            w1b = cast_bf16(w1)
            w2b = cast_bf16(w2)
            w2t = transpose_to_bf16(w2)              # [H, D]
            wts = _lib.MlpWeights(enc_dim=self.encoder_dim, k=self.k, hidden=self.hidden_dim, llm_dim=self.llm_dim,
                                  eps=1e-6, w1=w1b.data_ptr(), w2=w2b.data_ptr(), w2_t=w2t.data_ptr(),
                                  g1=g1.data_ptr(), g2=g2.data_ptr())
            self._pack = (wts, (w1b, w2b, w2t, g1, g2))
            self._pack_versions = versions
        return self._pack[0]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, S, encoder_dim] (bf16 encoder output) -> [B, (S-k)//k+1, llm_dim] fp32."""
        if not x.is_cuda and not _lib.DRY_RUN:
            raise _lib.Ta355Error("MLPAudioProjector runs on the MI355X HIP path only (no CPU fallback)")
        return _MLPProjectorFn.apply(x, self.linear_1.weight, self.norm.weight, self.linear_2.weight, self.norm_2.weight,
                                     self)


PROJECTOR_CLASSES = {"mlp": MLPAudioProjector}
