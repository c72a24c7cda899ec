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
import os

import torch

from . import _lib
from .asr_config import LMConfig
from .ops import BF16, F32, ptr, stream


def _pad128(n):
    return (n + 127) // 128 * 128


# LoRA adapters of linears that share an input are stacked into one GROUP (csrc/lora.hip): group -> peft target names
LORA_GROUPS = (("qkv", ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj")), ("o", ("self_attn.o_proj",)),
               ("gu", ("mlp.gate_proj", "mlp.up_proj")), ("d", ("mlp.down_proj",)))


class Qwen3MI355X(torch.nn.Module):
    def __init__(self, config: LMConfig, device="cuda"):
        super().__init__()
        self.config = config
        self.device_ = torch.device(device)
        self.vocab_pad = _pad128(config.vocab_size)
        self._bufs = {}
        self._w = None
        self._layers_arr = None
        self.lora_rank = 0
        self.lora_groups = 0
        self.lora_targets = set()
        self._lora_bound = None
        self.train_base = False          # full decoder fine-tuning (freeze_language_model=False)
        self.want_train_base = False     # ... requested before the weights exist: enabled by _finalize
        self._ft_bound = None
        self._ft_versions = None
        self.accumulate_into_grad = False   # trainer opt-in: weight gradients are added straight into Parameter.grad
        self._res_f32 = self._dx_f32 = False   # storage of the residual / d(x) streams (ta_lm_weights.res_f32 / dx_f32): bf16
        self._tape_modes = {}

    # ------------------------------------------------------------------ LoRA (stage 2; asr_modeling.py:289-301)
    # storage of the forward residual stream (+ its tape rows) and of the backward d(x) stream: fields of THIS model's weights handle
    # (include/ta355.h, ABI 4) -- rounds 1-5 kept them in process-wide library state
    @property
    def res_f32(self) -> bool:
        return self._res_f32

    @res_f32.setter
    def res_f32(self, v):
        self._res_f32 = bool(v)
        if self._w is not None:
            self._w.res_f32 = int(self._res_f32)

    @property
    def dx_f32(self) -> bool:
        return self._dx_f32

    @dx_f32.setter
    def dx_f32(self, v):
        self._dx_f32 = bool(v)
        if self._w is not None:
            self._w.dx_f32 = int(self._dx_f32)

    def _lora_dims(self):
        """group -> [(peft target, out_features)], in_features"""
        c = self.config
        D, F, nq, nkv, hd = c.hidden_size, c.intermediate_size, c.num_attention_heads, c.num_key_value_heads, c.head_dim
        outs = {"self_attn.q_proj": nq * hd, "self_attn.k_proj": nkv * hd, "self_attn.v_proj": nkv * hd,
                "self_attn.o_proj": D, "mlp.gate_proj": F, "mlp.up_proj": F, "mlp.down_proj": D}
        ins = {"qkv": D, "o": nq * hd, "gu": D, "d": F}
        return {g: ([(t, outs[t]) for t in ts], ins[g]) for g, ts in LORA_GROUPS}

    def lora_parameters(self):
        return [getattr(self, f"lora_{ab}_{g}") for g, _ in LORA_GROUPS for ab in ("la", "lb")] if self.lora_rank else []

    @torch.no_grad()
    def enable_lora(self, rank=8, alpha=32, dropout=0.0, target_modules=None, seed=0):
        """peft ``LoraConfig(r, lora_alpha, target_modules, bias='none')`` on every decoder layer (tiny_audio/asr_modeling.py:289-301,
        defaults tiny_audio/asr_config.py:72-75: r = 8, alpha = 32, all 7 linears): lora_A ~ kaiming_uniform(a=sqrt(5)) =
        U(+-1/sqrt(in)), lora_B = 0, y += (alpha/r) B A x.  The trainable fp32 masters are 8 Parameters [n_layers, ...] in the
        stacked group layout of include/ta355.h.

        ``rank``: any r with 3 r <= 64 (the q|k|v group's three adapters share one 64-wide K tile).
        ``target_modules``: any subset of the 7 linears (peft suffix names).  A group none of whose members is targeted is
        switched off in the C library (``ta_lm_weights.lora_groups``); inside a live group the members that are not targeted
        keep A = B = 0, which makes both of their gradients exactly zero (dA = s (dy B)^T x, dB = dy^T (x A^T)), so AdamW
        never moves them; ``export_lora_state_dict`` / ``lora_param_count`` only see the targeted ones."""
        if self.train_base or self.want_train_base:
            raise NotImplementedError("LoRA on top of a trainable base LM is not built")
        if not (1 <= int(rank) and 3 * int(rank) <= 64):
            raise NotImplementedError("lora_rank must satisfy 3 * rank <= 64 (one 64-wide K tile per adapter group)")
        if dropout:
            raise NotImplementedError("lora_dropout > 0 is not built (reference default 0.0, asr_config.py:74)")
        all_t = [t for _, ts in LORA_GROUPS for t in ts]
        if target_modules is None:
            targets = set(all_t)
        else:
            short = {t.split(".")[-1]: t for t in all_t}
            unknown = [t for t in target_modules if t.split(".")[-1] not in short]
            if unknown or not list(target_modules):
                raise ValueError(f"lora_target_modules {unknown or '[]'}: expected a non-empty subset of {sorted(short)}")
            targets = {short[t.split(".")[-1]] for t in target_modules}
        rank = int(rank)
        Lyr, dev = self.config.num_hidden_layers, self.device_
        gen = torch.Generator(device="cpu"); gen.manual_seed(seed)
        self.lora_groups = 0
        for gi, (g, (members, fin)) in enumerate(self._lora_dims().items()):
            bound = 1.0 / math.sqrt(fin)
            a = (torch.rand((Lyr, len(members) * rank, fin), generator=gen, dtype=F32) * 2 - 1) * bound
            for j, (t, _) in enumerate(members):
                if t not in targets:
                    a[:, j * rank:(j + 1) * rank] = 0
                else:
                    self.lora_groups |= 1 << gi
            b = torch.zeros((Lyr, sum(o for _, o in members), rank), dtype=F32)
            setattr(self, f"lora_la_{g}", torch.nn.Parameter(a.to(dev)))
            setattr(self, f"lora_lb_{g}", torch.nn.Parameter(b.to(dev)))
        self.lora_rank, self.lora_alpha, self.lora_targets = rank, alpha, targets
        self._lora_bound = None
        self._finalize_lora_scale()
        return self

    def lora_param_count(self):
        """peft's trainable-parameter count: r (in + out) per targeted linear and layer (the stacked masters also hold the
        zero blocks of members that are not targeted)."""
        if not self.lora_rank:
            return 0
        n = 0
        for g, (members, fin) in self._lora_dims().items():
            n += sum(self.lora_rank * (fin + o) for t, o in members if t in self.lora_targets)
        return n * self.config.num_hidden_layers

    def _finalize_lora_scale(self):
        if self._w is not None and self.lora_rank:
            self._w.lora_rank, self._w.lora_scale = self.lora_rank, float(self.lora_alpha) / self.lora_rank
            self._w.lora_groups = self.lora_groups

    @torch.no_grad()
    def load_lora_state_dict(self, sd):
        """Accepts peft adapter naming (``...model.layers.N.self_attn.q_proj.lora_A[.default].weight``) or the bare
        ``model.layers.N.<target>.lora_A`` keys of oracle/weights.py:init_lora."""
        r = self.lora_rank
        found = {}
        for k, v in sd.items():
            k2 = k.replace(".default", "")
            if k2.endswith(".weight"):
                k2 = k2[: -len(".weight")]
            i = k2.find("layers.")
            if i >= 0 and (k2.endswith(".lora_A") or k2.endswith(".lora_B")):
                found[k2[i:]] = torch.as_tensor(v).to(dtype=F32)
        for g, (members, fin) in self._lora_dims().items():
            la, lb = getattr(self, f"lora_la_{g}"), getattr(self, f"lora_lb_{g}")
            for i in range(self.config.num_hidden_layers):
                row = 0
                for j, (t, o) in enumerate(members):
                    if t in self.lora_targets:
                        la.data[i, j * r:(j + 1) * r].copy_(found[f"layers.{i}.{t}.lora_A"])
                        lb.data[i, row:row + o].copy_(found[f"layers.{i}.{t}.lora_B"])
                    row += o
        return self

    def export_lora_state_dict(self, prefix="base_model.model.model.", suffix=".weight"):
        """peft adapter_model naming by default; ``prefix='model.', suffix=''`` gives the oracle's keys."""
        r, sd = self.lora_rank, {}
        for g, (members, fin) in self._lora_dims().items():
            la, lb = getattr(self, f"lora_la_{g}").detach(), getattr(self, f"lora_lb_{g}").detach()
            for i in range(self.config.num_hidden_layers):
                row = 0
                for j, (t, o) in enumerate(members):
                    if t in self.lora_targets:
                        sd[f"{prefix}layers.{i}.{t}.lora_A{suffix}"] = la[i, j * r:(j + 1) * r].clone()
                        sd[f"{prefix}layers.{i}.{t}.lora_B{suffix}"] = lb[i, row:row + o].clone()
                    row += o
        return sd

    def _bind_lora(self):
        """Point the per-layer struct fields at the current storage of the 8 master Parameters (a flat-buffer
        optimizer may have re-homed ``.data`` since the last call)."""
        ps = self.lora_parameters()
        key = tuple(p.data_ptr() for p in ps)
        if key == self._lora_bound:
            return
        for p in ps:
            assert p.dtype == F32 and p.is_contiguous()
        for g, _ in LORA_GROUPS:
            for ab in ("la", "lb"):
                p = getattr(self, f"lora_{ab}_{g}")
                step = p[0].numel() * 4
                for i in range(self.config.num_hidden_layers):
                    setattr(self._layers_arr[i], f"{ab}_{g}", p.data_ptr() + i * step)
        self._lora_bound = key

    # ------------------------------------------------------------------ full decoder fine-tuning (8(f) rank 4)
    # (kind, per-layer struct field or None, needs bf16 images, decays)
    FT_KINDS = (("wqkv", True), ("wo", True), ("wgu", True), ("wd", True), ("ln_in_w", False), ("ln_post_w", False),
                ("qn_w", False), ("kn_w", False))

    def ft_parameters(self):
        """The fp32 masters of the trainable LM: 8 stacked per-layer Parameters [n_layers, ...] + final norm + embedding."""
        if not self.train_base:
            return []
        return [getattr(self, "ft_" + k) for k, _ in self.FT_KINDS] + [self.ft_norm_w, self.ft_embed]

    @torch.no_grad()
    def enable_full_finetune(self):
        """freeze_language_model=False (tiny_audio/asr_config.py:77; configs/experiments/embedded.yaml:23): every LM weight
        trains.  Masters are fp32 Parameters (stacked over the layers, q|k|v and gate|up fused like the kernels' images);
        the kernels keep reading bf16 images (W and W^T) that ``refresh_images`` rebuilds after an optimizer step, the
        norm scales and the embedding table are read from the masters directly."""
        if self.lora_rank:
            raise NotImplementedError("LoRA on top of a trainable base LM is not built")
        if self._w is None:
            raise _lib.Ta355Error("load or initialise the LM weights before enable_full_finetune()")
        c, b = self.config, self._bufs
        Lyr = c.num_hidden_layers
        src = getattr(self, "_fp32_src", None) or {}
        for kind, _img in self.FT_KINDS:
            rows = [src.get(f"layers.{i}.{kind}", b[f"layers.{i}.{kind}"]).to(F32) for i in range(Lyr)]
            p = torch.nn.Parameter(torch.stack(rows, 0).contiguous())
            p._no_decay = not _img                      # norm scales: the no-decay group (scripts/train.py:397-432)
            setattr(self, "ft_" + kind, p)
        self.ft_norm_w = torch.nn.Parameter(b["norm_w"].to(F32).clone()); self.ft_norm_w._no_decay = True
        self.ft_embed = torch.nn.Parameter(b["embed_f32"].clone())
        self._fp32_src = None
        self.train_base = True
        self._w.train_base = 1
        self._ft_bound = self._ft_versions = None
        self._bind_ft()
        return self

    def _bind_ft(self):
        """Point the kernels at the masters they read directly (norm scales, embedding) -- a flat-buffer optimizer may have
        re-homed ``.data`` -- and rebuild the bf16 images if any master changed since they were built."""
        ps = self.ft_parameters()
        key = tuple(p.data_ptr() for p in ps)
        if key != self._ft_bound:
            for p in ps:
                assert p.dtype == F32 and p.is_contiguous()
            for kind in ("ln_in_w", "ln_post_w", "qn_w", "kn_w"):
                p = getattr(self, "ft_" + kind)
                step = p[0].numel() * 4
                for i in range(self.config.num_hidden_layers):
                    setattr(self._layers_arr[i], kind, p.data_ptr() + i * step)
            self._w.norm_w = self.ft_norm_w.data_ptr()
            self._w.embed_f32 = self.ft_embed.data_ptr()
            # the buffer table follows the masters too (the initial copies are dropped: the embedding alone is 0.6 GB)
            b = self._bufs
            b["embed_f32"], b["norm_w"] = self.ft_embed.data, self.ft_norm_w.data
            for kind in ("ln_in_w", "ln_post_w", "qn_w", "kn_w"):
                m = getattr(self, "ft_" + kind).data
                for i in range(self.config.num_hidden_layers):
                    b[f"layers.{i}.{kind}"] = m[i]
            self._ft_bound = key
            self._ft_versions = None
        ver = tuple(p._version for p in ps)
        if ver != self._ft_versions:
            self.refresh_images()

    @torch.no_grad()
    def refresh_images(self):
        """bf16 W / W^T images of the four matrices of every layer and of the tied embedding, from the fp32 masters
        (in place: the kernels' pointers stay valid).  Call after every optimizer step that writes the masters through
        raw pointers; torch optimizers are detected through the Parameters' version counters."""
        b = self._bufs
        gpu = self.device_.type == "cuda" and not _lib.DRY_RUN
        L_ = _lib.lib() if gpu else None

        def images(master, wb, wt):                      # fp32 [N, K] -> bf16 [N, K] and bf16 [K, N]
            if not gpu:
                wb.copy_(master); wt[:, :master.shape[0]].copy_(wb.t()); return
            N, K = master.shape
            _lib.check(L_.ta_cast_f32_bf16(ptr(master), ptr(wb), master.numel(), stream()), "ta_cast_f32_bf16")
            _lib.check(L_.ta_transpose_to_bf16(ptr(master), 1, K, 0, 0, ptr(wt), wt.shape[1], N, K, stream()), "ta_transpose_to_bf16")
        for kind, img in self.FT_KINDS:
            if not img:
                continue
            m = getattr(self, "ft_" + kind)
            for i in range(self.config.num_hidden_layers):
                images(m[i], b[f"layers.{i}.{kind}"], b[f"layers.{i}.{kind}_t"])
        V = self.config.vocab_size
        images(self.ft_embed, b["embed_bf16"][:V], b["embed_t_bf16"])       # rows V..vocab_pad stay zero in both images
        self._ft_versions = tuple(p._version for p in self.ft_parameters())

    def ft_state_dict_hf(self):
        """The masters under the reference's parameter names (``model.layers.N.self_attn.q_proj.weight`` ...; tiny_audio/asr_modeling.py:409-421;
        the tied ``lm_head.weight`` alias of the embedding is not repeated: safetensors refuses shared storage)."""
        c = self.config
        nq, nkv, hd, F_ = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.intermediate_size
        sd = {"model.embed_tokens.weight": self.ft_embed.detach(), "model.norm.weight": self.ft_norm_w.detach()}
        for i in range(c.num_hidden_layers):
            p = f"model.layers.{i}."
            a = p + "self_attn."
            w = self.ft_wqkv.detach()[i]
            sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"] = w[: nq * hd], w[nq * hd:(nq + nkv) * hd], w[(nq + nkv) * hd:]
            sd[a + "o_proj.weight"] = self.ft_wo.detach()[i]
            gu = self.ft_wgu.detach()[i]
            sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = gu[:F_], gu[F_:]
            sd[p + "mlp.down_proj.weight"] = self.ft_wd.detach()[i]
            sd[p + "input_layernorm.weight"], sd[p + "post_attention_layernorm.weight"] = self.ft_ln_in_w.detach()[i], self.ft_ln_post_w.detach()[i]
            sd[a + "q_norm.weight"], sd[a + "k_norm.weight"] = self.ft_qn_w.detach()[i], self.ft_kn_w.detach()[i]
        return sd

    @torch.no_grad()
    def load_ft_state_dict_hf(self, sd):
        """Inverse of ``ft_state_dict_hf`` (keys may carry a ``language_model.`` prefix; ``lm_head.weight`` is ignored:
        it is the embedding)."""
        c = self.config
        g = lambda k: torch.as_tensor(sd[k] if k in sd else sd["language_model." + k]).to(device=self.device_, dtype=F32)
        self.ft_embed.copy_(g("model.embed_tokens.weight")); self.ft_norm_w.copy_(g("model.norm.weight"))
        for i in range(c.num_hidden_layers):
            p = f"model.layers.{i}."
            a = p + "self_attn."
            self.ft_wqkv[i].copy_(torch.cat([g(a + "q_proj.weight"), g(a + "k_proj.weight"), g(a + "v_proj.weight")], 0))
            self.ft_wo[i].copy_(g(a + "o_proj.weight"))
            self.ft_wgu[i].copy_(torch.cat([g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight")], 0))
            self.ft_wd[i].copy_(g(p + "mlp.down_proj.weight"))
            self.ft_ln_in_w[i].copy_(g(p + "input_layernorm.weight")); self.ft_ln_post_w[i].copy_(g(p + "post_attention_layernorm.weight"))
            self.ft_qn_w[i].copy_(g(a + "q_norm.weight")); self.ft_kn_w[i].copy_(g(a + "k_norm.weight"))
        self.refresh_images()
        return self

    # ------------------------------------------------------------------ weights
    def _rope_tables(self):
        c = self.config
        inv = 1.0 / (c.rope_theta ** (torch.arange(0, c.head_dim, 2, dtype=torch.float32) / c.head_dim))  # modeling_qwen3.py:117
        freqs = torch.arange(c.max_position_embeddings, dtype=torch.float32)[:, None] * inv[None, :]
        return freqs.cos().contiguous(), freqs.sin().contiguous()

    keep_fp32 = False     # set before loading weights that will be fine-tuned: the fp32 originals seed the masters

    def _pack_matrix(self, name, w):
        """w fp32 [out, in] on device -> bf16 copy + transposed bf16 copy."""
        if self.keep_fp32:
            if getattr(self, "_fp32_src", None) is None:
                self._fp32_src = {}
            self._fp32_src[name] = w
        wb = w.to(BF16).contiguous()
        self._bufs[name] = wb
        self._bufs[name + "_t"] = wb.t().contiguous()

    def _set_embedding(self, emb):
        c, dev = self.config, self.device_
        V, D = emb.shape
        # Hub checkpoints carry padded embedding matrices (Qwen3-0.6B / 1.7B: 151 936 rows); the reference calls
        # resize_token_embeddings(len(tokenizer)) after adding <audio> (tiny_audio/asr_modeling.py:160-171), which keeps
        # the FIRST vocab_size rows -- row vocab_size-1 becomes the <audio> token.  Do the same here.
        if D != c.hidden_size or V < c.vocab_size:
            raise ValueError(f"embed_tokens is [{V}, {D}]; the config needs at least [{c.vocab_size}, {c.hidden_size}]")
        if V > c.vocab_size:
            emb = emb[: c.vocab_size]
            V = c.vocab_size
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
                if not f.startswith(("la_", "lb_")):
                    setattr(arr[i], f, b[f"layers.{i}.{f}"].data_ptr())
        w = _lib.LmWeights(vocab=c.vocab_size, vocab_pad=self.vocab_pad, hidden=c.hidden_size, ffn=c.intermediate_size,
                           n_layers=L, heads=c.num_attention_heads, kv_heads=c.num_key_value_heads, head_dim=c.head_dim,
                           max_pos=c.max_position_embeddings, eps=c.rms_norm_eps, res_f32=int(self._res_f32),
                           dx_f32=int(self._dx_f32))
        for f in ("embed_f32", "embed_bf16", "embed_t_bf16", "norm_w", "rope_cos", "rope_sin"):
            setattr(w, f, b[f].data_ptr())
        w.layers = C.cast(arr, C.POINTER(_lib.LmLayer))
        self._layers_arr, self._w, self._lora_bound = arr, w, None
        self._finalize_lora_scale()
        if self.train_base or self.want_train_base:      # (re)seed the masters from the weights just loaded
            self.train_base = False
            self.enable_full_finetune()

    def export_state_dict_hf(self):
        """Back to the reference's parameter names as fp32 numpy (used by bench.py's CPU-baseline leg)."""
        c, b = self.config, self._bufs
        nq, nkv, hd, F = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.intermediate_size
        f = lambda t: t.detach().float().cpu().numpy()
        if self.train_base:                       # fine-tuned: the fp32 masters, not the bf16 images
            return {k: f(v) for k, v in self.ft_state_dict_hf().items()}
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

    @torch.no_grad()
    def set_embedding_weight(self, weight):
        """Replace the tied token embedding (``set_input_embeddings`` / ``set_output_embeddings`` of the reference,
        tiny_audio/asr_modeling.py:372-382): the fp32 lookup table and both bf16 lm_head images are rebuilt and the weights
        handle re-pointed.  With a trainable base LM the fp32 master takes the values and the images follow at the next forward."""
        w = torch.as_tensor(weight).detach()
        if self.train_base:
            self.ft_embed.copy_(w[: self.config.vocab_size].to(self.ft_embed))
            self._ft_versions = None
            return
        self._set_embedding(w)
        if self._w is not None:
            for f in ("embed_f32", "embed_bf16", "embed_t_bf16"):
                setattr(self._w, f, self._bufs[f].data_ptr())

    # ------------------------------------------------------------------ raw forward / backward (no autograd)
    def forward_loss(self, input_ids, src_row, audio, kmask, label_rows, label_targets, n_label_rows, loss_scale,
                     want_logits=False, pos=None):
        """Returns (loss[1] f32, nll[n] f32, logits or None, ctx) -- ctx feeds ``backward_from_ctx``."""
        if self._w is None:
            raise _lib.Ta355Error("LM weights not loaded")
        L_ = _lib.lib()
        B, L = input_ids.shape
        dev = self.device_
        if self.lora_rank:
            self._bind_lora()
        if self.train_base:
            self._bind_ft()
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
                   n_label_rows=n_label_rows, ids=input_ids)
        # The tape's residual-stream rows are in the storage dtype this handle carried NOW: the backward must read them as that even
        # if the owner flips ``res_f32`` in between.  Keyed by the tape's address, since the custom operator hands the backward only
        # tensors.
        if len(self._tape_modes) > 64:
            self._tape_modes.clear()
        self._tape_modes[tape.data_ptr()] = (int(self._w.res_f32), int(self._w.dx_f32))
        return loss, nll, logits, ctx

    def backward_from_ctx(self, ctx, n_audio_rows, want_d_embeds=False, want_d_audio=True):
        """-> (d_audio f32 [n_audio_rows, D] or None, d_embeds f32 [B*L, D] or None, trainable-LM gradients or None) for
        d(loss) = 1.  The third item is the list of adapter gradients (order of ``lora_parameters()``) with LoRA, the list
        of weight gradients (order of ``ft_parameters()``) with a trainable base LM -- or ``[None] * 10`` when
        ``accumulate_into_grad`` is set and they were added straight into ``Parameter.grad``."""
        D, dev = self.config.hidden_size, self.device_
        d_audio = torch.empty((n_audio_rows, D), device=dev, dtype=F32) if want_d_audio else None
        d_emb = torch.empty((ctx["B"] * ctx["L"], D), device=dev, dtype=F32) if want_d_embeds else None
        lg, lg_arr, wg, keep = None, None, None, None
        nl = self.config.num_hidden_layers
        if self.lora_rank:
            self._bind_lora()
            lg = [torch.empty_like(p) for p in self.lora_parameters()]
            lg_arr = (_lib.LmLoraGrads * nl)()
            k = 0
            for g, _ in LORA_GROUPS:
                for ab in ("la", "lb"):
                    step = lg[k][0].numel() * 4
                    for i in range(nl):
                        setattr(lg_arr[i], f"d{ab}_{g}", lg[k].data_ptr() + i * step)
                    k += 1
        if self.train_base:
            ps = self.ft_parameters()
            direct = self.accumulate_into_grad and all(p.grad is not None and p.grad.is_contiguous() for p in ps)
            bufs = [p.grad for p in ps] if direct else [torch.zeros_like(p) for p in ps]
            arr = (_lib.LmLayerWgrads * nl)()
            for (kind, _), g in zip(self.FT_KINDS, bufs):
                field = {"wqkv": "dwqkv", "wo": "dwo", "wgu": "dwgu", "wd": "dwd", "ln_in_w": "dln_in", "ln_post_w": "dln_post",
                         "qn_w": "dqn", "kn_w": "dkn"}[kind]
                step = g[0].numel() * 4
                for i in range(nl):
                    setattr(arr[i], field, g.data_ptr() + i * step)
            wg = _lib.LmWgrads(layers=C.cast(arr, C.POINTER(_lib.LmLayerWgrads)), dnorm=bufs[8].data_ptr(), dembed=bufs[9].data_ptr())
            keep = (arr, bufs)
            lg = [None] * len(ps) if direct else bufs
        w = self._w
        recorded = self._tape_modes.pop(ctx["tape"].data_ptr(), None)
        if recorded is not None and recorded != (int(w.res_f32), int(w.dx_f32)):
            w = _lib.LmWeights.from_buffer_copy(self._w)          # same pointers, the modes the forward ran in
            w.res_f32, w.dx_f32 = recorded
        _lib.check(_lib.lib().ta_lm_backward(C.byref(w), ptr(ctx["src_row"]), ptr(ctx["kmask"]), ptr(ctx["pos"]),
                                             ctx["B"], ctx["L"], ptr(ctx["label_rows"]), ctx["n_label_rows"],
                                             ptr(d_audio), n_audio_rows, ptr(d_emb), lg_arr,
                                             None if wg is None else C.byref(wg), ptr(ctx.get("ids")), ptr(ctx["tape"]),
                                             ptr(ctx["ws"]), ctx["ws"].numel(), stream()), "ta_lm_backward")
        del keep
        return d_audio, d_emb, lg


    # ------------------------------------------------------------------ greedy decoding
    @torch.no_grad()
    def greedy_decode(self, input_ids, src_row, audio, attention_mask=None, max_new_tokens=128, eos_ids=(), pad_id=0,
                      sync_every=8, use_graph=True, repetition_penalty=1.0, no_repeat_ngram_size=0, min_new_tokens=0, sampling=None):
        """-> int64 [B, n_new]; see ``greedy_decode_iter`` (this drains it, polling the device every ``sync_every``
        steps only)."""
        out = None
        for out in self.greedy_decode_iter(input_ids, src_row, audio, attention_mask, max_new_tokens, eos_ids, pad_id,
                                           sync_every, use_graph, per_token=False, repetition_penalty=repetition_penalty, min_new_tokens=min_new_tokens, sampling=sampling,
                                           no_repeat_ngram_size=no_repeat_ngram_size):
            pass
        return out

    def greedy_decode_iter(self, input_ids, src_row, audio, attention_mask=None, max_new_tokens=128, eos_ids=(), pad_id=0,
                           sync_every=8, use_graph=True, per_token=True, repetition_penalty=1.0, no_repeat_ngram_size=0, min_new_tokens=0,
                           processors_see_prompt=True, sampling=None):
        """HF greedy search with a KV cache (what ``language_model.generate`` does for the reference's generation
        config, tiny_audio/asr_config.py:103-111): prompt pass, then one token per clip per step until every clip has
        emitted an eos id or ``max_new_tokens`` is reached.  -> int64 [B, n_new] (prompt stripped; finished clips are
        padded with ``pad_id``).  All per-step state lives on the device; the host only polls the number of
        unfinished clips every ``sync_every`` steps.

        A generator: with ``per_token`` it yields the int64 [B] host tensor of every step's tokens as soon as that step
        has run (one device sync per token: the streaming mode of tiny_audio/asr_modeling.py:648-760) and stops when
        every clip is finished; the last item yielded is always the full [B, n_new] device tensor."""
        if self._w is None:
            raise _lib.Ta355Error("LM weights not loaded")
        L_ = _lib.lib()
        c, dev = self.config, self.device_
        B, L = input_ids.shape
        max_new = int(max_new_tokens)
        if max_new <= 0:
            yield torch.empty((B, 0), dtype=torch.int64, device=dev)
            return
        Lmax = L + max_new
        if Lmax > c.max_position_embeddings:
            raise ValueError(f"prompt ({L}) + max_new_tokens ({max_new}) exceeds max_position_embeddings")
        i32, i64 = torch.int32, torch.int64
        att = torch.ones((B, L), dtype=i32, device=dev) if attention_mask is None else attention_mask.to(device=dev, dtype=i32)
        kmask = torch.zeros((B, Lmax), dtype=i32, device=dev)
        kmask[:, :L] = att
        pos_full = (att.cumsum(-1) - 1).clamp(min=0).to(i32).contiguous()        # HF: position_ids from the mask
        pos = att.sum(-1).to(i32).contiguous()                                     # position of the first new token
        idx = torch.arange(L, device=dev, dtype=i32)[None, :].expand(B, L)
        last = (idx * att).max(dim=-1).values                                      # last valid prompt token of each clip
        last_rows = (torch.arange(B, device=dev, dtype=i32) * L + last.to(i32)).contiguous()
        shape = (c.num_hidden_layers, B, c.num_key_value_heads, Lmax, c.head_dim)
        kc, vc = torch.empty(shape, dtype=BF16, device=dev), torch.empty(shape, dtype=BF16, device=dev)
        lora_img = None
        if self.train_base:
            self._bind_ft()
        if self.lora_rank:
            self._bind_lora()
            lora_img = torch.empty(L_.ta_lm_lora_image_bytes(C.byref(self._w)), dtype=torch.uint8, device=dev)
        ws = torch.empty(max(L_.ta_lm_prefill_workspace_bytes(C.byref(self._w), B, L),
                             L_.ta_lm_decode_workspace_bytes(C.byref(self._w), B)), dtype=torch.uint8, device=dev)
        logits = torch.empty((B, self.vocab_pad), dtype=F32, device=dev)
        amax = torch.zeros(B, dtype=i64, device=dev)
        finished = torch.zeros(B, dtype=i32, device=dev)
        next_ids = torch.zeros(B, dtype=i64, device=dev)
        out_seq = torch.full((B, max_new), int(pad_id), dtype=i64, device=dev)
        step_dev = torch.zeros(1, dtype=i32, device=dev)
        slot_dev = torch.full((1,), L, dtype=i32, device=dev)
        alive = torch.full((1,), B, dtype=i32, device=dev)
        eos = torch.tensor(list(eos_ids) or [-1], dtype=i64, device=dev)
        n_eos = len(list(eos_ids))
        ids = input_ids.to(device=dev, dtype=i64).contiguous()
        a = None if audio is None else audio.detach().to(F32).contiguous()

        rep, ngram, min_new = float(repetition_penalty), int(no_repeat_ngram_size), int(min_new_tokens or 0)

        def advance():
            if rep != 1.0 or ngram > 0:            # HF logits processors over prompt ids + generated tokens (device-side, graph-safe)
                # processors_see_prompt=False: the generated tokens only -- what HF's processors see when generate() is given
                # inputs_embeds WITHOUT input_ids (the reference's generate_streaming, tiny_audio/asr_modeling.py:723-729)
                _lib.check(L_.ta_logits_process(ptr(logits), self.vocab_pad, c.vocab_size, ptr(ids), L if processors_see_prompt else 0, ptr(out_seq), max_new,
                                                ptr(step_dev), B, rep, ngram, stream()), "ta_logits_process")
            if min_new > 0 and n_eos:               # HF MinNewTokensLengthLogitsProcessor: no eos before min_new_tokens tokens exist
                _lib.check(L_.ta_logits_suppress_until(ptr(logits), self.vocab_pad, c.vocab_size, ptr(eos), n_eos, min_new, ptr(step_dev), B,
                                                       stream()), "ta_logits_suppress_until")
            if sampling is not None:                # do_sample: HF's warpers (temperature, top-k, top-p), then one multinomial draw per clip
                temp, top_k, top_p, seed = sampling
                _lib.check(L_.ta_logits_warp(ptr(logits), self.vocab_pad, c.vocab_size, B, float(temp), int(top_k), float(top_p), stream()),
                           "ta_logits_warp")
                _lib.check(L_.ta_sample_f32(ptr(logits), self.vocab_pad, c.vocab_size, B, int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(step_dev),
                                            ptr(amax), stream()), "ta_sample_f32")
            else:
                _lib.check(L_.ta_argmax_f32(ptr(logits), self.vocab_pad, c.vocab_size, B, ptr(amax), stream()), "ta_argmax_f32")
            _lib.check(L_.ta_greedy_advance(ptr(amax), ptr(eos), n_eos, int(pad_id), ptr(finished), ptr(next_ids), ptr(out_seq),
                                            max_new, ptr(step_dev), ptr(slot_dev), ptr(pos), ptr(kmask), Lmax, B, ptr(alive),
                                            stream()), "ta_greedy_advance")

        _lib.check(L_.ta_lm_prefill(C.byref(self._w), ptr(ids), ptr(src_row), ptr(a), ptr(att.contiguous()), ptr(pos_full), B, L,
                                    ptr(kc), ptr(vc), Lmax, ptr(last_rows), ptr(logits), ptr(lora_img), ptr(ws), ws.numel(),
                                    stream()), "ta_lm_prefill")
        advance()

        def step():
            _lib.check(L_.ta_lm_decode_step(C.byref(self._w), ptr(next_ids), ptr(pos), ptr(kmask), ptr(slot_dev), B, ptr(kc),
                                            ptr(vc), Lmax, ptr(logits), ptr(lora_img), ptr(ws), ws.numel(), stream()),
                       "ta_lm_decode_step")
            advance()

        # A decode step is ~340 tiny launches whose arguments never change (all per-step state is device-resident):
        # run the first one eagerly, capture the second into a hipGraph, replay it for the rest.
        graph = None
        want_graph = use_graph and dev.type == "cuda" and not _lib.DRY_RUN and max_new > 3      # (``use_graph=False``: eager steps)
        if per_token:
            yield out_seq[:, 0].cpu()
        for t in range(1, max_new):
            if (per_token or t % sync_every == 0) and int(alive.item()) == 0:
                break
            if graph is not None:
                graph.replay()
            elif want_graph and t == 2:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    step()
                graph.replay()                      # capture records, it does not run
            else:
                step()
            if per_token:
                yield out_seq[:, t].cpu()
        seq = out_seq.cpu()
        # HF stops right after the step in which the last clip finished: trim the surplus (all-pad) columns
        n_new = max_new
        if n_eos:
            is_eos = torch.isin(seq, torch.tensor(list(eos_ids), dtype=i64))
            first = torch.where(is_eos.any(-1), is_eos.to(torch.int8).argmax(-1) + 1, torch.full((B,), max_new + 1))
            n_new = min(max_new, int(first.max()))
        yield out_seq[:, :n_new]


class FrozenLMLoss:
    """loss = CE(frozen_LM(embed(ids) with <audio> rows := audio_embeds)) through ``torch.ops.ta355.lm_forward_loss``
    (torch_ops.py); grad flows to audio_embeds and to the trainable LM tensors passed as trailing inputs -- the LoRA masters
    (``*lm.lora_parameters()``) or, with a trainable base LM, its fp32 masters (``*lm.ft_parameters()``).

    accumulate_into_grad (full decoder fine-tuning under ASRTrainer): the kernels add the LM's weight gradients straight
    into the trainer's flat buffer, so d(loss) cannot be applied to them afterwards.  ASRTrainer is the only caller that
    sets the flag and it always back-propagates the CE with d(loss) == 1 (the auxiliary term is added outside this node,
    the token normalisation happens in the optimizer kernel); any other use must leave it off."""

    @staticmethod
    def apply(audio_embeds, lm, input_ids, src_row, kmask, label_rows, label_targets, n_label_rows, loss_scale,
              want_logits, *trainable, pos=None):
        from . import torch_ops
        loss, nll, logits, _tape, _ws = torch.ops.ta355.lm_forward_loss(
            audio_embeds, list(trainable), torch_ops.register_module(lm), input_ids, src_row, kmask, label_rows, label_targets,
            int(n_label_rows), float(loss_scale), bool(want_logits), pos)
        # a fresh tensor: the outputs of a multi-output custom op may not be modified in place, and HF Trainer does
        # `loss *= ...` on what the model returns (TF:trainer.py compute_loss)
        return loss.clone(), nll, logits
