"""Frozen Qwen3 causal LM on MI355X (drop-in for ``ASRModel.language_model`` on the training path).

Reference: ``Qwen3ForCausalLM.forward`` TF:models/qwen3/modeling_qwen3.py:448-508 + ``ForCausalLMLoss``
TF:loss/loss_utils.py:33-71, called at tiny_audio/asr_modeling.py:517-526 with frozen weights
(``requires_grad_(False)``, :251-253), so backward is activation-gradient only.

Every frozen matrix is kept twice in bf16 -- [out,in] for the forward GEMM and [in,out] for dX = dY W --
so both directions run the same NT MFMA kernel (288 GB of HBM: the second copy costs 1.2 GB).
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from .asr_config import LMConfig
from .ops import BF16, F32, ptr, stream


def _pad128(n):
    return (n + 127) // 128 * 128


class Qwen3MI355X(torch.nn.Module):
    def __init__(self, config: LMConfig, device="cuda"):
        super().__init__()
        self.config = config
        self.device_ = torch.device(device)
        self.vocab_pad = _pad128(config.vocab_size)
        self._bufs = {}
        self._w = None
        self._layers_arr = None

    # ------------------------------------------------------------------ weights
    def _rope_tables(self):
        c = self.config
        inv = 1.0 / (c.rope_theta ** (torch.arange(0, c.head_dim, 2, dtype=torch.float32) / c.head_dim))  # modeling_qwen3.py:117
        freqs = torch.arange(c.max_position_embeddings, dtype=torch.float32)[:, None] * inv[None, :]
        return freqs.cos().contiguous(), freqs.sin().contiguous()

    def _pack_matrix(self, name, w):
        """w fp32 [out, in] on device -> bf16 copy + transposed bf16 copy."""
        wb = w.to(BF16).contiguous()
        self._bufs[name] = wb
        self._bufs[name + "_t"] = wb.t().contiguous()

    def _set_embedding(self, emb):
        c, dev = self.config, self.device_
        V, D = emb.shape
        assert V == c.vocab_size and D == c.hidden_size
        self._bufs["embed_f32"] = emb.to(device=dev, dtype=F32).contiguous()
        eb = torch.zeros((self.vocab_pad, D), device=dev, dtype=BF16)
        eb[:V] = self._bufs["embed_f32"].to(BF16)
        self._bufs["embed_bf16"] = eb
        self._bufs["embed_t_bf16"] = eb.t().contiguous()

    def load_state_dict_hf(self, sd):
        """sd: {``model.layers.N.self_attn.q_proj.weight`` ...: array-like fp32} in the reference's naming."""
        c, dev = self.config, self.device_
        g = lambda k: torch.as_tensor(sd[k]).to(device=dev, dtype=F32)
        self._bufs = {}
        self._set_embedding(g("model.embed_tokens.weight"))
        self._bufs["norm_w"] = g("model.norm.weight").contiguous()
        for i in range(c.num_hidden_layers):
            p = f"model.layers.{i}."
            q = f"layers.{i}."
            a = p + "self_attn."
            self._pack_matrix(q + "wqkv", torch.cat([g(a + "q_proj.weight"), g(a + "k_proj.weight"), g(a + "v_proj.weight")], 0))
            self._pack_matrix(q + "wo", g(a + "o_proj.weight"))
            self._pack_matrix(q + "wgu", torch.cat([g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight")], 0))
            self._pack_matrix(q + "wd", g(p + "mlp.down_proj.weight"))
            self._bufs[q + "ln_in_w"] = g(p + "input_layernorm.weight").contiguous()
            self._bufs[q + "ln_post_w"] = g(p + "post_attention_layernorm.weight").contiguous()
            self._bufs[q + "qn_w"] = g(a + "q_norm.weight").contiguous()
            self._bufs[q + "kn_w"] = g(a + "k_norm.weight").contiguous()
        self._finalize()
        return self

    @torch.no_grad()
    def random_init(self, seed=1):
        c, dev = self.config, self.device_
        D, F, V = c.hidden_size, c.intermediate_size, c.vocab_size
        nq, nkv, hd = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        gen = torch.Generator(device=dev); gen.manual_seed(seed)
        rn = lambda *s, std=1.0: torch.randn(*s, device=dev, generator=gen, dtype=F32) * std
        self._bufs = {}
        self._set_embedding(rn(V, D, std=1 / math.sqrt(D)))
        self._bufs["norm_w"] = 1 + rn(D, std=0.1)
        for i in range(c.num_hidden_layers):
            q = f"layers.{i}."
            self._pack_matrix(q + "wqkv", rn((nq + 2 * nkv) * hd, D, std=1 / math.sqrt(D)))
            self._pack_matrix(q + "wo", rn(D, nq * hd, std=0.5 / math.sqrt(nq * hd)))
            self._pack_matrix(q + "wgu", rn(2 * F, D, std=1 / math.sqrt(D)))
            self._pack_matrix(q + "wd", rn(D, F, std=0.5 / math.sqrt(F)))
            self._bufs[q + "ln_in_w"] = 1 + rn(D, std=0.1)
            self._bufs[q + "ln_post_w"] = 1 + rn(D, std=0.1)
            self._bufs[q + "qn_w"] = 1 + rn(hd, std=0.1)
            self._bufs[q + "kn_w"] = 1 + rn(hd, std=0.1)
        self._finalize()
        return self

    def _finalize(self):
        c, b, dev = self.config, self._bufs, self.device_
        cos, sin = self._rope_tables()
        b["rope_cos"], b["rope_sin"] = cos.to(dev), sin.to(dev)
        L = c.num_hidden_layers
        arr = (_lib.LmLayer * L)()
        for i in range(L):
            for f, _ in _lib.LmLayer._fields_:
                setattr(arr[i], f, b[f"layers.{i}.{f}"].data_ptr())
        w = _lib.LmWeights(vocab=c.vocab_size, vocab_pad=self.vocab_pad, hidden=c.hidden_size, ffn=c.intermediate_size,
                           n_layers=L, heads=c.num_attention_heads, kv_heads=c.num_key_value_heads, head_dim=c.head_dim,
                           max_pos=c.max_position_embeddings, eps=c.rms_norm_eps)
        for f in ("embed_f32", "embed_bf16", "embed_t_bf16", "norm_w", "rope_cos", "rope_sin"):
            setattr(w, f, b[f].data_ptr())
        w.layers = C.cast(arr, C.POINTER(_lib.LmLayer))
        self._layers_arr, self._w = arr, w

    def export_state_dict_hf(self):
        """Back to the reference's parameter names as fp32 numpy (used by bench.py's CPU-baseline leg)."""
        c, b = self.config, self._bufs
        nq, nkv, hd, F = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.intermediate_size
        f = lambda t: t.detach().float().cpu().numpy()
        sd = {"model.embed_tokens.weight": f(b["embed_f32"]), "model.norm.weight": f(b["norm_w"])}
        for i in range(c.num_hidden_layers):
            p, q = f"model.layers.{i}.", f"layers.{i}."
            a = p + "self_attn."
            w = f(b[q + "wqkv"])
            sd[a + "q_proj.weight"], sd[a + "k_proj.weight"] = w[: nq * hd], w[nq * hd:(nq + nkv) * hd]
            sd[a + "v_proj.weight"] = w[(nq + nkv) * hd:]
            sd[a + "o_proj.weight"] = f(b[q + "wo"])
            gu = f(b[q + "wgu"])
            sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = gu[:F], gu[F:]
            sd[p + "mlp.down_proj.weight"] = f(b[q + "wd"])
            sd[p + "input_layernorm.weight"], sd[p + "post_attention_layernorm.weight"] = f(b[q + "ln_in_w"]), f(b[q + "ln_post_w"])
            sd[a + "q_norm.weight"], sd[a + "k_norm.weight"] = f(b[q + "qn_w"]), f(b[q + "kn_w"])
        return sd

    def get_input_embeddings_weight(self):
        return self._bufs["embed_f32"]

    # ------------------------------------------------------------------ raw forward / backward (no autograd)
    def forward_loss(self, input_ids, src_row, audio, kmask, label_rows, label_targets, n_label_rows, loss_scale,
                     want_logits=False, pos=None):
        """Returns (loss[1] f32, nll[n] f32, logits or None, ctx) -- ctx feeds ``backward_from_ctx``."""
        if self._w is None:
            raise _lib.Ta355Error("LM weights not loaded")
        L_ = _lib.lib()
        B, L = input_ids.shape
        dev = self.device_
        tape = torch.empty(L_.ta_lm_tape_bytes(C.byref(self._w), B, L, n_label_rows), device=dev, dtype=torch.uint8)
        ws = torch.empty(L_.ta_lm_workspace_bytes(C.byref(self._w), B, L, n_label_rows), device=dev, dtype=torch.uint8)
        loss = torch.zeros(1, device=dev, dtype=F32)
        nll = torch.empty(max(n_label_rows, 1), device=dev, dtype=F32)
        logits = torch.empty((B * L, self.vocab_pad), device=dev, dtype=BF16) if want_logits else None
        _lib.check(L_.ta_lm_forward_loss(C.byref(self._w), ptr(input_ids), ptr(src_row), ptr(audio), ptr(kmask), ptr(pos),
                                         B, L, ptr(label_rows), ptr(label_targets), n_label_rows, loss_scale, ptr(loss),
                                         ptr(nll), ptr(logits), ptr(tape), ptr(ws), ws.numel(), stream()),
                   "ta_lm_forward_loss")
        ctx = dict(tape=tape, ws=ws, B=B, L=L, src_row=src_row, kmask=kmask, pos=pos, label_rows=label_rows,
                   n_label_rows=n_label_rows)
        return loss, nll, logits, ctx

    def backward_from_ctx(self, ctx, n_audio_rows, want_d_embeds=False):
        """-> (d_audio f32 [n_audio_rows, D], d_embeds f32 [B*L, D] or None) for d(loss) = 1."""
        D, dev = self.config.hidden_size, self.device_
        d_audio = torch.empty((n_audio_rows, D), device=dev, dtype=F32)
        d_emb = torch.empty((ctx["B"] * ctx["L"], D), device=dev, dtype=F32) if want_d_embeds else None
        _lib.check(_lib.lib().ta_lm_backward(C.byref(self._w), ptr(ctx["src_row"]), ptr(ctx["kmask"]), ptr(ctx["pos"]),
                                             ctx["B"], ctx["L"], ptr(ctx["label_rows"]), ctx["n_label_rows"],
                                             ptr(d_audio), n_audio_rows, ptr(d_emb), ptr(ctx["tape"]), ptr(ctx["ws"]),
                                             ctx["ws"].numel(), stream()), "ta_lm_backward")
        return d_audio, d_emb


class FrozenLMLoss(torch.autograd.Function):
    """loss = CE(frozen_LM(embed(ids) with <audio> rows := audio_embeds)); grad flows to audio_embeds only."""

    @staticmethod
    def forward(ctx, audio_embeds, lm, input_ids, src_row, kmask, label_rows, label_targets, n_label_rows, loss_scale,
                want_logits):
        a = audio_embeds.detach().to(F32).contiguous()
        loss, nll, logits, c = lm.forward_loss(input_ids, src_row, a, kmask, label_rows, label_targets, n_label_rows,
                                               loss_scale, want_logits)
        ctx.lm, ctx.c, ctx.n_audio = lm, c, a.shape[0]
        ctx.mark_non_differentiable(nll)
        if logits is None:
            logits = torch.empty(0, device=a.device)
        ctx.mark_non_differentiable(logits)
        return loss.reshape(()), nll, logits

    @staticmethod
    def backward(ctx, g_loss, _g_nll, _g_logits):
        d_audio, _ = ctx.lm.backward_from_ctx(ctx.c, ctx.n_audio)
        ctx.c = None
        return (d_audio * g_loss,) + (None,) * 9
