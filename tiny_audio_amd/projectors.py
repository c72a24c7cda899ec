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

from . import _lib, torch_ops
from .ops import BF16, F32, cast_bf16, ptr, stream, transpose_to_bf16


class _RMSNormWeight(nn.Module):
    """Parameter holder with the reference's name (``LlamaRMSNorm.weight``); arithmetic is in the fused kernels."""

    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.variance_epsilon = eps


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

    def _packed_weights_meta(self):
        """Dimensions only (null pointers): enough for the host-side ``*_bytes`` size queries of the C ABI."""
        return _lib.MlpWeights(enc_dim=self.encoder_dim, k=self.k, hidden=self.hidden_dim, llm_dim=self.llm_dim, eps=1e-6)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, S, encoder_dim] (bf16 encoder output) -> [B, (S-k)//k+1, llm_dim] fp32."""
        if not x.is_cuda and not _lib.DRY_RUN:
            raise _lib.Ta355Error("MLPAudioProjector runs on the MI355X HIP path only (no CPU fallback)")
        # torch.ops.ta355.mlp_projector (torch_ops.py): forward + registered autograd; outputs 1, 2 are its saved state
        return torch.ops.ta355.mlp_projector(x, self.linear_1.weight, self.norm.weight, self.linear_2.weight, self.norm_2.weight,
                                             torch_ops.register_module(self))[0]


# =============================================================================
# Shared + sparse MoE projector (tiny_audio/projectors.py:185-351)
# =============================================================================
class SimpleAdapter(nn.Module):
    """Parameter holder for a 2-layer GELU adapter (tiny_audio/projectors.py:90-100): fc1 / fc2 with bias."""

    def __init__(self, input_dim, hidden_dim, output_dim):
        super().__init__()
        self.fc1 = nn.Linear(input_dim, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, output_dim)


class MoEAudioProjector(nn.Module):
    """MoE projector with shared expert (DeepSeek-style): norm -> shared adapter + top-2-of-E routed adapters."""

    def __init__(self, config):
        super().__init__()
        self.k = getattr(config, "projector_pool_stride", 4)
        self.aux_coef = getattr(config, "router_aux_loss_coef", 0.01)
        self.router_z_loss_coef = getattr(config, "router_z_loss_coef", 1e-4)          # projectors.py:206-211
        self.router_jitter_noise = getattr(config, "router_jitter_noise", 0.01)
        self.encoder_dim, self.llm_dim = config.encoder_dim, config.llm_dim
        in_dim = config.encoder_dim * self.k
        self.hidden_dim = getattr(config, "projector_hidden_dim", None) or config.llm_dim
        self.num_experts = getattr(config, "num_experts", 4)
        self.top_k = getattr(config, "num_experts_per_tok", 2)
        if self.top_k != 2 or not (2 <= self.num_experts <= 8):
            raise ValueError("ta355 MoE kernels implement top-2 routing over 2..8 experts (the reference's configuration)")
        self.norm = _RMSNormWeight(in_dim, eps=1e-6)
        self.router = nn.Linear(in_dim, self.num_experts, bias=False)
        self.experts = nn.ModuleList([SimpleAdapter(in_dim, self.hidden_dim, config.llm_dim) for _ in range(self.num_experts)])
        self.shared_expert = SimpleAdapter(in_dim, self.hidden_dim, config.llm_dim)
        self._init_weights()
        self.last_aux_loss = torch.tensor(0.0)
        self._aux_shadow = None          # ASRTrainer (more than one rank / gradient accumulation): shadow gradient segments of the aux share
        self._grad_direct = None         # ASRTrainer (one micro-batch per step): the flat-buffer segments the backward writes straight into
        self._pack = None
        self._pack_versions = None

    def _init_weights(self):
        """projectors.py:242-251: router N(0, .02); fc1 xavier-uniform; fc2 N(0, .01)."""
        with torch.no_grad():
            nn.init.normal_(self.router.weight, mean=0.0, std=0.02)
            for expert in [self.shared_expert, *self.experts]:
                nn.init.xavier_uniform_(expert.fc1.weight)
                nn.init.normal_(expert.fc2.weight, mean=0.0, std=0.01)

    def get_output_length(self, input_length):
        return (input_length - self.k) // self.k + 1

    # the parameters the auxiliary losses reach (through the router logits): ASRTrainer gives them a shadow gradient segment
    aux_shadow_params = ("norm.weight", "router.weight")

    def get_aux_loss(self):
        return self.last_aux_loss

    def _param_list(self):
        ps = [self.norm.weight, self.router.weight]
        for a in list(self.experts) + [self.shared_expert]:
            ps += [a.fc1.weight, a.fc1.bias, a.fc2.weight, a.fc2.bias]
        return ps

    def _packed_weights(self):
        ps = self._param_list()
        versions = tuple((p.data_ptr(), p._version) for p in ps)
        if self._pack is None or versions != self._pack_versions:
            E = self.num_experts
            keep = []
            f32 = lambda p: p.detach().to(F32).contiguous()
            norm_w, router_w = f32(self.norm.weight), f32(self.router.weight)
            # every kind of image is ONE tensor over the E + 1 adapters: the routed experts then sit at a constant stride,
            # which lets libta355 run all of them in one grouped GEMM launch (ta_gemm_bf16_nt_grouped)
            dev, H, In, D = norm_w.device, self.hidden_dim, self.encoder_dim * self.k, self.llm_dim
            W1a, W1ta = torch.empty((E + 1, H, In), device=dev, dtype=BF16), torch.empty((E + 1, In, H), device=dev, dtype=BF16)
            W2a, W2ta = torch.empty((E + 1, D, H), device=dev, dtype=BF16), torch.empty((E + 1, H, D), device=dev, dtype=BF16)
            B1a, B2a = torch.empty((E + 1, H), device=dev, dtype=F32), torch.empty((E + 1, D), device=dev, dtype=F32)
            adapters = list(self.experts) + [self.shared_expert]
            m_w1, m_b1 = [f32(a.fc1.weight) for a in adapters], [f32(a.fc1.bias) for a in adapters]
            m_w2, m_b2 = [f32(a.fc2.weight) for a in adapters], [f32(a.fc2.bias) for a in adapters]
            parr = lambda ts: (C.c_void_p * (E + 1))(*[t.data_ptr() for t in ts])
            # every image of every adapter in two launches (ta_moe_pack_images; rounds 1-3: 20 cast / transpose launches + 10 bias copies)
            _lib.check(_lib.lib().ta_moe_pack_images(parr(m_w1), parr(m_b1), parr(m_w2), parr(m_b2), E + 1, H, In, D, ptr(W1a), ptr(W1ta),
                                                     ptr(W2a), ptr(W2ta), ptr(B1a), ptr(B2a), stream()), "ta_moe_pack_images")
            w1, w1t, b1 = list(W1a.unbind(0)), list(W1ta.unbind(0)), list(B1a.unbind(0))
            w2, w2t, b2 = list(W2a.unbind(0)), list(W2ta.unbind(0)), list(B2a.unbind(0))
            keep_m = [m_w1, m_b1, m_w2, m_b2]
            arr = lambda ts: (C.c_void_p * (E + 1))(*[t.data_ptr() for t in ts])
            arrays = [arr(w1), arr(w1t), arr(b1), arr(w2), arr(w2t), arr(b2)]
            wts = _lib.MoeWeights(enc_dim=self.encoder_dim, k=self.k, hidden=self.hidden_dim, llm_dim=self.llm_dim,
                                  num_experts=E, eps=1e-6, aux_coef=float(self.aux_coef), z_coef=float(self.router_z_loss_coef),
                                  norm_w=norm_w.data_ptr(), router_w=router_w.data_ptr())
            for name, a in zip(("w1", "w1_t", "b1", "w2", "w2_t", "b2"), arrays):
                setattr(wts, name, C.cast(a, C.POINTER(C.c_void_p)))
            keep = [norm_w, router_w, w1, w1t, b1, w2, w2t, b2, arrays, W1a, W1ta, W2a, W2ta, B1a, B2a, keep_m]
            self._pack = (wts, keep)
            self._pack_versions = versions
        return self._pack[0]

    def _packed_weights_meta(self):
        """Dimensions only (null pointers): enough for the host-side ``*_bytes`` size queries of the C ABI."""
        return _lib.MoeWeights(enc_dim=self.encoder_dim, k=self.k, hidden=self.hidden_dim, llm_dim=self.llm_dim,
                               num_experts=self.num_experts, eps=1e-6, aux_coef=float(self.aux_coef),
                               z_coef=float(self.router_z_loss_coef))

    def forward(self, x: torch.Tensor, jitter_noise: torch.Tensor = None) -> torch.Tensor:
        """x [B, S, encoder_dim] -> [B, N, llm_dim] fp32.  ``jitter_noise`` [B*N, E] injects the multiplicative router
        noise (tests); in training it is otherwise drawn U(1-eps, 1+eps) as the reference does (projectors.py:294-300)."""
        if not x.is_cuda and not _lib.DRY_RUN:
            raise _lib.Ta355Error("MoEAudioProjector runs on the MI355X HIP path only (no CPU fallback)")
        B, S, _ = x.shape
        T = B * self.get_output_length(S)
        noise = jitter_noise
        if noise is None and self.training and self.router_jitter_noise > 0:
            noise = torch.empty((T, self.num_experts), device=x.device, dtype=F32).uniform_(
                1.0 - self.router_jitter_noise, 1.0 + self.router_jitter_noise)
        if noise is not None:
            noise = noise.to(device=x.device, dtype=F32).contiguous()
        y, aux, _xb, _tape = torch.ops.ta355.moe_projector(x, noise, self._param_list(), torch_ops.register_module(self),
                                                           bool(self.training))
        self.last_aux_loss = aux
        return y


from .qformer_projector import QFormerAudioProjector  # noqa: E402
from .mosa_projector import MOSAProjector  # noqa: E402

PROJECTOR_CLASSES = {"mlp": MLPAudioProjector, "mosa": MOSAProjector, "moe": MoEAudioProjector, "qformer": QFormerAudioProjector}
