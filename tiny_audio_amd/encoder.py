"""Frozen GLM-ASR audio encoder on MI355X (drop-in for ``ASRModel.audio_tower``).

Reference: ``GlmAsrEncoder`` TF:models/glmasr/modeling_glmasr.py:286-327, invoked under ``no_grad`` at
tiny_audio/asr_modeling.py:448-450.  Weights are immutable bf16 device buffers packed for the HIP kernels
(fused q|k|v, conv taps flattened for the row-mapped GEMM); biases / LayerNorm parameters stay fp32.
``load_state_dict_hf`` accepts the reference's parameter names so real checkpoints drop in.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch

from . import _lib
from .asr_config import EncoderConfig
from .ops import BF16, F32, ptr, stream


def _t(x, device, dtype=None):
    t = torch.as_tensor(x)
    return t.to(device=device, dtype=dtype or t.dtype).contiguous()


class BaseModelOutput:
    def __init__(self, last_hidden_state):
        self.last_hidden_state = last_hidden_state


class GlmAsrEncoderMI355X(torch.nn.Module):
    """``encoder(input_features=[B, n_mels, T] f32).last_hidden_state -> [B, S, H]`` (bf16 by default)."""

    def __init__(self, config: EncoderConfig, device="cuda"):
        super().__init__()
        self.config = config
        self.device_ = torch.device(device)
        self._bufs = {}          # name -> tensor (keeps device memory alive)
        self._layers_arr = None
        self._w = None
        self._ws = None
        self._ws_key = None
        self.out_dtype = BF16
        self._res_f32 = False    # storage of the residual stream (ta_encoder_weights.res_f32): bf16 unless the owner asks for fp32

    @property
    def res_f32(self) -> bool:
        """True = the residual stream is stored in fp32 (the training recipe's fp32 modules under bf16 autocast), False = bf16
        (bf16 modules).  A field of THIS encoder's weights handle (include/ta355.h, ABI 4): no process-wide state."""
        return self._res_f32

    @res_f32.setter
    def res_f32(self, v):
        self._res_f32 = bool(v)
        if self._w is not None:
            self._w.res_f32 = int(self._res_f32)

    # ------------------------------------------------------------------ weights
    def _rope_tables(self):
        c = self.config
        rot = int((c.hidden_size // c.num_attention_heads) * c.partial_rotary_factor)       # 32
        inv = 1.0 / (c.rope_theta ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))  # modeling_glmasr.py:87
        freqs = torch.arange(c.max_position_embeddings, dtype=torch.float32)[:, None] * inv[None, :]
        return freqs.cos().contiguous(), freqs.sin().contiguous()

    def load_state_dict_hf(self, sd):
        """sd: {reference parameter name: array-like fp32} (numpy or torch)."""
        c, dev = self.config, self.device_
        H, L = c.hidden_size, c.num_hidden_layers
        b = self._bufs = {}
        g = lambda k: torch.as_tensor(sd[k]).float()
        b["conv1_w"] = _t(g("conv1.weight").permute(0, 2, 1).reshape(H, -1), dev, BF16)   # col = tap*Cin + cin
        b["conv1_b"] = _t(g("conv1.bias"), dev, F32)
        b["conv2_w"] = _t(g("conv2.weight").permute(0, 2, 1).reshape(H, -1), dev, BF16)
        b["conv2_b"] = _t(g("conv2.bias"), dev, F32)
        b["norm_w"] = _t(g("norm.weight"), dev, F32)
        b["norm_b"] = _t(g("norm.bias"), dev, F32)
        cos, sin = self._rope_tables()
        b["rope_cos"], b["rope_sin"] = _t(cos, dev, F32), _t(sin, dev, F32)
        self._q_masters = {}
        for i in range(L):
            p = f"layers.{i}."
            a = p + "self_attn."
            self._q_masters[i] = g(a + "q_proj.weight").detach().to(torch.float32).cpu()      # consumed by _derive_fused_qkv
            b[p + "wqkv"] = _t(torch.cat([g(a + "q_proj.weight"), g(a + "k_proj.weight"), g(a + "v_proj.weight")], 0), dev, BF16)
            b[p + "bqkv"] = _t(torch.cat([g(a + "q_proj.bias"), torch.zeros(H), g(a + "v_proj.bias")], 0), dev, F32)
            b[p + "wo"] = _t(g(a + "o_proj.weight"), dev, BF16)
            b[p + "bo"] = _t(g(a + "o_proj.bias"), dev, F32)
            b[p + "w1"] = _t(g(p + "mlp.fc1.weight"), dev, BF16)
            b[p + "b1"] = _t(g(p + "mlp.fc1.bias"), dev, F32)
            b[p + "w2"] = _t(g(p + "mlp.fc2.weight"), dev, BF16)
            b[p + "b2"] = _t(g(p + "mlp.fc2.bias"), dev, F32)
            for n, k in (("ln1_w", "input_layernorm.weight"), ("ln1_b", "input_layernorm.bias"),
                         ("ln2_w", "post_attention_layernorm.weight"), ("ln2_b", "post_attention_layernorm.bias")):
                b[p + n] = _t(g(p + k), dev, F32)
        self._finalize()
        return self

    @torch.no_grad()
    def random_init(self, seed=0):
        """Seeded random weights generated directly on the device at the configured shapes (bench / smoke)."""
        c, dev = self.config, self.device_
        H, F, M, L = c.hidden_size, c.intermediate_size, c.num_mel_bins, c.num_hidden_layers
        gen = torch.Generator(device=dev); gen.manual_seed(seed)
        rn = lambda *s, std=1.0: torch.randn(*s, device=dev, generator=gen, dtype=F32) * std
        b = self._bufs = {}
        b["conv1_w"] = rn(H, 3 * M, std=1 / math.sqrt(3 * M)).to(BF16)
        b["conv1_b"] = rn(H, std=0.02)
        b["conv2_w"] = rn(H, 3 * H, std=1 / math.sqrt(3 * H)).to(BF16)
        b["conv2_b"] = rn(H, std=0.02)
        b["norm_w"] = 1 + rn(H, std=0.1); b["norm_b"] = rn(H, std=0.02)
        cos, sin = self._rope_tables()
        b["rope_cos"], b["rope_sin"] = _t(cos, dev, F32), _t(sin, dev, F32)
        for i in range(L):
            p = f"layers.{i}."
            b[p + "wqkv"] = rn(3 * H, H, std=1 / math.sqrt(H)).to(BF16)
            bq = rn(3 * H, std=0.02); bq[H:2 * H] = 0
            b[p + "bqkv"] = bq
            b[p + "wo"] = rn(H, H, std=0.5 / math.sqrt(H)).to(BF16); b[p + "bo"] = rn(H, std=0.02)
            b[p + "w1"] = rn(F, H, std=1 / math.sqrt(H)).to(BF16); b[p + "b1"] = rn(F, std=0.02)
            b[p + "w2"] = rn(H, F, std=0.5 / math.sqrt(F)).to(BF16); b[p + "b2"] = rn(H, std=0.02)
            b[p + "ln1_w"] = 1 + rn(H, std=0.1); b[p + "ln1_b"] = rn(H, std=0.02)
            b[p + "ln2_w"] = 1 + rn(H, std=0.1); b[p + "ln2_b"] = rn(H, std=0.02)
        self._finalize()
        return self

    @torch.no_grad()
    def _derive_fused_qkv(self):
        """Weight images of the fused q|k|v path (ta355.h, ta_enc_layer.wqkv_fa): q|k rows with every head's rotary pairs
        interleaved, the v_proj bias folded through o_proj, and the (cos, sin) table in pair order."""
        c, b = self.config, self._bufs
        H, nh = c.hidden_size, c.num_attention_heads
        hd = H // nh
        dev = b["rope_cos"].device
        p = torch.arange(hd, device=dev)
        d_of_p = torch.where(p < 32, (p >> 1) + 16 * (p & 1), p)                       # column p of a head <- head dim
        rows = (torch.arange(nh, device=dev)[:, None] * hd + d_of_p[None, :]).reshape(-1)
        rows = torch.cat([rows, H + rows])
        b["rope_il"] = torch.stack([b["rope_cos"], b["rope_sin"]], dim=-1).contiguous()   # [max_pos, 16, 2]
        # ta_enc_layer.wqkv_fa: softmax scale and log2(e) folded into the q rows, so that ta_attention_enc_fwd exponentiates
        # q.k directly in base 2.  From the fp32 masters when load_state_dict_hf kept them (one rounding to bf16, like the
        # reference's own cast), else from the bf16 image.
        qs = (hd ** -0.5) * math.log2(math.e)
        masters = getattr(self, "_q_masters", None) or {}
        for i in range(c.num_hidden_layers):
            q = f"layers.{i}."
            wq32 = masters[i].to(dev) if i in masters else b[q + "wqkv"][:H].float()
            b[q + "wqkv_fa"] = torch.cat([(wq32[rows[:H]] * qs).to(BF16), b[q + "wqkv"][rows[H:]], b[q + "wqkv"][2 * H:]], 0).contiguous()
            b[q + "bqkv_fa"] = torch.cat([b[q + "bqkv"][rows[:H]] * qs, torch.zeros(H, device=dev), b[q + "bqkv"][2 * H:]], 0).contiguous()

    def _finalize(self):
        c, b = self.config, self._bufs
        L = c.num_hidden_layers
        if c.hidden_size // c.num_attention_heads == 64:
            self._derive_fused_qkv()
        self._q_masters = None
        arr = (_lib.EncLayer * L)()
        for i in range(L):
            p = f"layers.{i}."
            for f, _ in _lib.EncLayer._fields_:
                setattr(arr[i], f, b[p + f].data_ptr() if (p + f) in b else None)
        w = _lib.EncoderWeights(hidden=c.hidden_size, ffn=c.intermediate_size, n_layers=L, heads=c.num_attention_heads,
                                n_mels=c.num_mel_bins, max_pos=c.max_position_embeddings, ln_eps=c.layer_norm_eps,
                                res_f32=int(self._res_f32))
        for f in ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "norm_w", "norm_b", "rope_cos", "rope_sin"):
            setattr(w, f, b[f].data_ptr())
        w.rope_il = b["rope_il"].data_ptr() if "rope_il" in b else None
        w.layers = C.cast(arr, C.POINTER(_lib.EncLayer))
        self._layers_arr, self._w = arr, w

    def export_state_dict_hf(self):
        """Back to the reference's parameter names as fp32 numpy (used by bench.py's CPU-baseline leg)."""
        c, b = self.config, self._bufs
        H, M = c.hidden_size, c.num_mel_bins
        f = lambda t: t.detach().float().cpu().numpy()
        sd = {"conv1.weight": f(b["conv1_w"].float().reshape(H, 3, M).permute(0, 2, 1).contiguous()),
              "conv1.bias": f(b["conv1_b"]),
              "conv2.weight": f(b["conv2_w"].float().reshape(H, 3, H).permute(0, 2, 1).contiguous()),
              "conv2.bias": f(b["conv2_b"]), "norm.weight": f(b["norm_w"]), "norm.bias": f(b["norm_b"])}
        for i in range(c.num_hidden_layers):
            p = f"layers.{i}."
            a = p + "self_attn."
            w, bq = f(b[p + "wqkv"]), f(b[p + "bqkv"])
            sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"] = w[:H], w[H:2 * H], w[2 * H:]
            sd[a + "q_proj.bias"], sd[a + "v_proj.bias"] = bq[:H], bq[2 * H:]
            sd[a + "o_proj.weight"], sd[a + "o_proj.bias"] = f(b[p + "wo"]), f(b[p + "bo"])
            sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = f(b[p + "w1"]), f(b[p + "b1"])
            sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = f(b[p + "w2"]), f(b[p + "b2"])
            sd[p + "input_layernorm.weight"], sd[p + "input_layernorm.bias"] = f(b[p + "ln1_w"]), f(b[p + "ln1_b"])
            sd[p + "post_attention_layernorm.weight"] = f(b[p + "ln2_w"])
            sd[p + "post_attention_layernorm.bias"] = f(b[p + "ln2_b"])
        return sd

    # ------------------------------------------------------------------ forward
    def output_length(self, T):
        return (T - 1) // 2 + 1

    @torch.no_grad()
    def forward(self, input_features, frame_keep=None, return_f32=False, **_):
        """model.audio_tower(input_features=...).last_hidden_state through torch.ops.ta355.encoder_forward."""
        from . import torch_ops
        x = input_features.to(device=self.device_, dtype=F32)
        if frame_keep is not None:
            frame_keep = frame_keep.to(device=self.device_, dtype=F32).contiguous()
        return BaseModelOutput(torch.ops.ta355.encoder_forward(x, frame_keep, torch_ops.register_module(self), bool(return_f32)))

    def _forward_impl(self, input_features, frame_keep=None, return_f32=False):
        if self._w is None:
            raise _lib.Ta355Error("encoder weights not loaded (load_state_dict_hf / random_init)")
        x = input_features.to(device=self.device_, dtype=F32).contiguous()
        B, _, T = x.shape
        S, H = self.output_length(T), self.config.hidden_size
        key = (B, T)
        if self._ws_key != key:
            n = _lib.lib().ta_encoder_workspace_bytes(C.byref(self._w), B, T)
            self._ws = torch.empty(n, device=self.device_, dtype=torch.uint8)
            self._ws_key = key
        out_b = torch.empty((B, S, H), device=self.device_, dtype=BF16)
        out_f = torch.empty((B, S, H), device=self.device_, dtype=F32) if return_f32 else None
        if frame_keep is not None:
            frame_keep = frame_keep.to(device=self.device_, dtype=F32).contiguous()
        _lib.check(_lib.lib().ta_encoder_forward(C.byref(self._w), ptr(x), B, T, ptr(frame_keep), ptr(out_b), ptr(out_f),
                                                 ptr(self._ws), self._ws.numel(), stream()), "ta_encoder_forward")
        return out_f if return_f32 else out_b
