"""ASRModel on MI355X: drop-in for the training path of ``tiny_audio/asr_modeling.py``.

Same constructor argument (an ASRConfig), same ``forward(input_ids, input_features, audio_attention_mask,
attention_mask, labels, audio_token_counts, ...)`` -> object with ``.loss`` / ``.logits``, same sub-module
names (``audio_tower``, ``projector``, ``language_model``), projector-only ``state_dict()`` with the
reference's key names (asr_modeling.py:398-422).  Encoder and LM are frozen; ``loss.backward()`` produces
gradients for the projector parameters only, through two HIP composites (LM dX backward, projector backward).

There is no CPU path: without libta355.so / a GPU every entry point raises.
"""
from __future__ import annotations

from collections.abc import Mapping
from typing import Iterator, Optional

import torch
import torch.nn as nn

from . import _lib, ops
from .asr_config import ASRConfig, compute_encoder_output_length
from .encoder import GlmAsrEncoderMI355X
from .language_model import FrozenLMLoss, Qwen3MI355X
from .ops import F32
from .projectors import PROJECTOR_CLASSES


class CausalLMOutput(dict):
    """What ``ASRModel.forward`` returns.  Like transformers' ``ModelOutput`` (``CausalLMOutputWithPast``): the dict holds only
    the fields that are not None, in order -- ``loss``, ``logits`` -- with attribute, key, index and slice access (HF
    ``Trainer.compute_loss`` reads ``outputs["loss"]`` / ``outputs[0]``; ``Trainer.prediction_step`` builds
    ``tuple(v for k, v in outputs.items() if k != "loss")`` and concatenates it over batches, so no None and no Python scalar
    may sit among the items).  This path's extras are plain attributes, not items: ``nll`` (per label token),
    ``n_label_tokens`` (int), ``aux_loss``, ``loss_ce`` (the LM's cross-entropy alone: loss = loss_ce + aux_loss)."""

    _items = ("loss", "logits")
    _extras = ("nll", "n_label_tokens", "aux_loss", "loss_ce")

    def __init__(self, loss=None, logits=None, nll=None, n_label_tokens=None, aux_loss=None, loss_ce=None):
        super().__init__()
        if isinstance(loss, Mapping):                 # ModelOutput.__init__ accepts a mapping as its first argument, and library code
            src = loss                                # rebuilds outputs that way: accelerate's convert_to_fp32 / send_to_device do
            loss, logits = src.get("loss"), src.get("logits")     # type(data)({k: f(v) ...}) around model.forward under bf16=True
            unknown = [k for k in src if k not in self._items]
            if unknown:
                raise KeyError(f"CausalLMOutput has no item {unknown[0]!r}")
            if isinstance(src, CausalLMOutput):       # (a rebuild from a plain dict cannot carry the attribute extras: they become None)
                nll, n_label_tokens, aux_loss, loss_ce = src.nll, src.n_label_tokens, src.aux_loss, src.loss_ce
        for k, v in (("loss", loss), ("logits", logits)):
            if v is not None:
                dict.__setitem__(self, k, v)
        for k, v in (("nll", nll), ("n_label_tokens", n_label_tokens), ("aux_loss", aux_loss), ("loss_ce", loss_ce)):
            object.__setattr__(self, k, v)

    def __getattr__(self, k):                     # reached only when the normal lookup fails: an item, or an absent item
        if k in CausalLMOutput._items:
            return dict.get(self, k)
        raise AttributeError(k)

    def __getitem__(self, k):
        if isinstance(k, (int, slice)):
            return tuple(self.values())[k]
        return dict.__getitem__(self, k)

    def to_tuple(self):
        return tuple(self.values())

    def __reduce__(self):                             # copy / pickle keep the extras
        return (CausalLMOutput, (dict.get(self, "loss"), dict.get(self, "logits"), self.nll, self.n_label_tokens, self.aux_loss, self.loss_ce))


def _is_cjk(ch: str) -> bool:
    """CJK ideograph blocks (a streamer releases such characters one by one: they carry no spaces)."""
    cp = ord(ch)
    return any(lo <= cp <= hi for lo, hi in ((0x4E00, 0x9FFF), (0x3400, 0x4DBF), (0x20000, 0x2A6DF), (0x2A700, 0x2B73F),
                                             (0x2B740, 0x2B81F), (0x2B820, 0x2CEAF), (0xF900, 0xFAFF), (0x2F800, 0x2FA1F)))


class _TextPieces:
    """Incremental detokeniser with the release rule of transformers' TextStreamer (generation/streamers.py
    ``put``/``end``): decode the tokens of the current line, release a finished line whole, a trailing CJK character at
    once, otherwise everything up to the last space; ``flush`` releases the rest."""

    def __init__(self, tokenizer):
        self.tok, self.ids, self.shown = tokenizer, [], 0

    def push(self, token_id: int) -> str:
        self.ids.append(int(token_id))
        text = self.tok.decode(self.ids, skip_special_tokens=True)
        if text.endswith("\n"):
            out, self.ids, self.shown = text[self.shown:], [], 0
        elif text and _is_cjk(text[-1]):
            out, self.shown = text[self.shown:], len(text)
        else:
            cut = text.rfind(" ") + 1
            out = text[self.shown:cut] if cut > self.shown else ""
            self.shown = max(self.shown, cut)
        return out

    def flush(self) -> str:
        out = self.tok.decode(self.ids, skip_special_tokens=True)[self.shown:] if self.ids else ""
        self.ids, self.shown = [], 0
        return out


class _ThinkGate:
    """Drops ``<think>...</think>`` spans from a stream of text pieces (tiny_audio/asr_modeling.py:737-757)."""

    def __init__(self):
        self.buf, self.inside = "", False

    def feed(self, text: str):
        self.buf += text
        while "<think>" in self.buf:
            self.inside = True
            before, self.buf = self.buf.split("<think>", 1)
            if before:
                yield before
        while self.inside and "</think>" in self.buf:
            self.inside = False
            self.buf = self.buf.split("</think>", 1)[1]
        if not self.inside and self.buf:
            out, self.buf = self.buf, ""
            yield out

    def flush(self) -> str:
        out = "" if self.inside else self.buf
        self.buf = ""
        return out


class ASRModel(nn.Module):
    config_class = ASRConfig
    main_input_name = "input_features"
    TRANSCRIBE_PROMPT = "Transcribe the speech to text"

    def __init__(self, config: ASRConfig, device="cuda", init="random", seed=0, **kwargs):
        super().__init__()
        self.config = config
        self.device_ = torch.device(device)
        self.audio_tower = GlmAsrEncoderMI355X(config.audio_config, device=device)        # asr_modeling.py:140
        self.language_model = Qwen3MI355X(config.text_config, device=device)              # :143
        self.audio_token_id = config.audio_token_id
        self.projector = self._create_projector(config).to(device=device, dtype=F32)      # :163
        if not getattr(config, "freeze_language_model", True):
            # full decoder fine-tuning (asr_modeling.py:251-253; embedded.yaml:23): the LM's fp32 masters become Parameters
            # as soon as its weights exist (now for init="random", at load_state_dict_hf otherwise)
            self.language_model.keep_fp32 = self.language_model.want_train_base = True
        if init == "random":
            self.audio_tower.random_init(seed)
            self.language_model.random_init(seed + 1)
        if getattr(config, "use_lora", False):
            self._setup_lora(config, seed)
        if getattr(config, "freeze_projector", False):
            self.projector.requires_grad_(False)
        self._drop_seed = 0x5EED + seed
        self.tokenizer = kwargs.get("tokenizer")        # optional: only needed when generate() must build the prompt
        if self.tokenizer is not None and hasattr(self.tokenizer, "convert_tokens_to_ids"):
            # the reference reads the placeholder's id from the tokenizer (tiny_audio/asr_modeling.py:160-171), config.json
            # does not carry it
            tid = self.tokenizer.convert_tokens_to_ids("<audio>")
            if tid is not None and int(tid) >= 0:
                self.audio_token_id = config.audio_token_id = int(tid)
        self.system_prompt = getattr(config, "system_prompt", None)
        # the reference builds its WhisperFeatureExtractor here (tiny_audio/asr_modeling.py:146, :190-201; padding disabled for
        # GLM-ASR); this one computes the log-mel on the device.  ``feature_extractor=`` hands in another one.
        self.feature_extractor = kwargs.get("feature_extractor") or self._create_feature_extractor(config)

    def _apply_stream_modes(self):
        """``config.model_dtype`` decides where the residual streams are STORED, as it does in the reference: "bfloat16" (the
        reference ASRConfig's default, tiny_audio/asr_config.py:41: bf16 modules) -> bf16 streams; "float32" (the training
        recipe, configs/config.yaml:14-18 with bf16 autocast) -> fp32 streams.  Compute is bf16 MFMA with fp32 accumulation in
        both (the reference's ``bf16: true``); the trainable masters are fp32 in both.  ``config.residual_dtype`` overrides."""
        rd = getattr(self.config, "residual_dtype", None) or getattr(self.config, "model_dtype", "bfloat16")
        f32 = str(rd).replace("torch.", "") in ("float32", "fp32", "float")
        # per-model state since round 6 (fields of the two weights handles, include/ta355.h ABI 4): another ASRModel of another
        # model_dtype in the same process, or a decoding thread, is not affected
        if self.audio_tower.res_f32 != f32:
            self.audio_tower.res_f32 = f32
        lm = self.language_model
        if lm.res_f32 != f32 or lm.dx_f32 != f32:
            lm.res_f32 = lm.dx_f32 = f32

    def _setup_lora(self, config, seed=0):
        """Stage-2 adapters on the LM (tiny_audio/asr_modeling.py:289-301: LoraConfig(r, lora_alpha,
        target_modules, lora_dropout, bias="none", task_type="CAUSAL_LM"))."""
        self.language_model.enable_lora(rank=config.lora_rank, alpha=config.lora_alpha, dropout=config.lora_dropout,
                                        target_modules=config.lora_target_modules, seed=seed + 2)

    def _create_feature_extractor(self, config):
        from .asr_processing import LogMelFeatureExtractor
        return LogMelFeatureExtractor(int(getattr(config.audio_config, "num_mel_bins", 128)), self.device_)

    def get_processor(self):
        """tiny_audio/asr_modeling.py:384-396: the processor that pairs with this model (its feature extractor, its tokenizer, its
        projector's length rule, its encoder's conv geometry)."""
        from .asr_processing import ASRProcessor
        if self.tokenizer is None:
            raise ValueError("get_processor needs a tokenizer: construct ASRModel(..., tokenizer=tok) or set model.tokenizer")
        return ASRProcessor(feature_extractor=self.feature_extractor, tokenizer=self.tokenizer, projector=self.projector,
                            encoder_conv_layers=self.config.encoder_conv_layers)

    # ---- the PreTrainedModel surface HF tooling touches (tiny_audio/asr_modeling.py:359-382, :535-546)
    def get_input_embeddings(self):
        """A frozen ``nn.Embedding`` VIEW of the LM's fp32 lookup table (no copy); the training forward does not go through it
        (the lookup is fused with the <audio> scatter, ``ta_embed_scatter``)."""
        w = self.language_model.ft_embed if self.language_model.train_base else self.language_model.get_input_embeddings_weight()
        emb = nn.Embedding(w.shape[0], w.shape[1], _weight=w.detach(), _freeze=True)
        return emb

    def set_input_embeddings(self, value):
        """Replace the (tied) token embedding: the lookup table and both lm_head images are rebuilt from ``value.weight``."""
        self.language_model.set_embedding_weight(value.weight if hasattr(value, "weight") else value)

    def get_output_embeddings(self):
        """The tied lm_head as a bias-free ``nn.Linear`` VIEW of the same table (Qwen3 ties them: SURVEY section 8)."""
        w = self.language_model.ft_embed if self.language_model.train_base else self.language_model.get_input_embeddings_weight()
        head = nn.Linear(w.shape[1], w.shape[0], bias=False, device="meta")
        head.weight = nn.Parameter(w.detach(), requires_grad=False)
        return head

    def set_output_embeddings(self, value):
        self.set_input_embeddings(value)

    def _set_gradient_checkpointing(self, enable: bool = True, gradient_checkpointing_func=None):
        """The reference forwards this to the LM so that its activations are recomputed in the backward (:359-370).  Here the LM
        keeps a 40 KB-per-token tape (7.4 GB at B = 32) out of 288 GB and nothing is recomputed: accepted and
        recorded, so ``TrainingArguments(gradient_checkpointing=True)`` runs unchanged; results are identical either way."""
        self.gradient_checkpointing = bool(enable)

    def gradient_checkpointing_enable(self, gradient_checkpointing_kwargs=None):
        self._set_gradient_checkpointing(True)

    def gradient_checkpointing_disable(self):
        self._set_gradient_checkpointing(False)

    def prepare_inputs_for_generation(self, *args, **kwargs):
        """HF's ``GenerationMixin`` protocol (:535-546: audio features only on the step with ``cache_position[0] == 0``).  This
        model owns its KV cache and decode loop (``generate`` / ``generate_streaming``), so there is no per-step ``forward`` for
        the protocol to feed -- raising beats silently returning inputs that ``forward`` would refuse."""
        raise NotImplementedError("ASRModel on MI355X decodes with its own device-resident loop: call generate() / "
                                  "generate_streaming(); HF's step-wise GenerationMixin protocol is not spoken")

    def _create_projector(self, config):
        projector_type = getattr(config, "projector_type", "mlp")
        cls = PROJECTOR_CLASSES.get(projector_type)
        if cls is None:
            raise ValueError(f"Unknown projector_type: {projector_type}. Valid options: {list(PROJECTOR_CLASSES.keys())}")
        return cls(config)

    # frozen sub-models hold plain device buffers, not Parameters: parameters()/state_dict() cover the projector, the LoRA
    # adapters and -- with freeze_language_model=False -- the LM's fp32 masters
    def state_dict(self, *args, **kwargs):
        sd = {f"projector.{k}": v for k, v in self.projector.state_dict().items()}
        if self.language_model.lora_rank:       # peft adapter naming under the reference's attribute name
            sd.update(self.language_model.export_lora_state_dict(prefix="language_model.base_model.model.model."))
        if self.language_model.train_base:      # fine-tuned LM: saved with the projector (asr_modeling.py:409-421)
            sd.update({"language_model." + k: v for k, v in self.language_model.ft_state_dict_hf().items()})
        return sd

    def load_state_dict(self, sd, strict=True):
        sub = {k[len("projector."):]: v for k, v in sd.items() if k.startswith("projector.")}
        out = self.projector.load_state_dict(sub, strict=strict)
        if self.language_model.lora_rank and any(".lora_A" in k for k in sd):
            self.language_model.load_lora_state_dict(sd)
        if self.language_model.train_base and "language_model.model.embed_tokens.weight" in sd:
            self.language_model.load_ft_state_dict_hf(sd)
        if hasattr(self.projector, "_pack_versions"):
            self.projector._pack_versions = None
        return out

    def train(self, mode: bool = True):
        """Frozen sub-modules never enter train mode (asr_modeling.py:344-357)."""
        super().train(mode)
        self.audio_tower.train(False)
        self.language_model.train(False)
        return self

    def _compute_encoder_output_lengths(self, audio_attention_mask):
        return compute_encoder_output_length(audio_attention_mask.sum(dim=-1), self.config.encoder_conv_layers)

    def _frame_keep_mask(self, B, S, frame_keep=None):
        """Whole-frame Bernoulli keep mask of _maybe_drop_audio_tokens (asr_modeling.py:458-479); applied inside
        the encoder's final LayerNorm kernel.  ``frame_keep`` injects a mask (parity tests)."""
        if frame_keep is not None:
            return frame_keep
        p = float(getattr(self.config, "audio_token_dropout", 0.0))
        if not self.training or p <= 0.0:
            return None
        self._drop_seed += 1
        return ops.bernoulli_keep(B * S, 1.0 - p, self._drop_seed, self.device_)

    def _encode_audio(self, audio_features, frame_keep=None, after_encoder=None):
        """-> projector output [B, N, llm_dim] fp32 (packing into <audio> rows happens in the LM op).
        ``after_encoder``: called between the frozen encoder and the projector -- the last point of a step that has not
        read a trainable weight yet (ASRTrainer applies a deferred optimizer update there)."""
        B, _, T = audio_features.shape
        S = self.audio_tower.output_length(T)
        keep = self._frame_keep_mask(B, S, frame_keep)
        hidden = self.audio_tower(audio_features, frame_keep=keep).last_hidden_state      # no_grad inside
        if after_encoder is not None:
            after_encoder()
        return self.projector(hidden)

    def forward(self, input_ids: Optional[torch.Tensor] = None, input_features: Optional[torch.Tensor] = None,
                audio_attention_mask: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                labels: Optional[torch.Tensor] = None, audio_token_counts: Optional[torch.Tensor] = None,
                num_items_in_batch=None, return_logits: bool = True, frame_keep=None, after_encoder=None,
                position_ids: Optional[torch.Tensor] = None, past_key_values=None, inputs_embeds=None,
                use_cache: Optional[bool] = None, cache_position=None, **kwargs):
        """Training/eval forward (tiny_audio/asr_modeling.py:481-533).

        ``position_ids`` [B, L] (or [1, L] / [L], broadcast over the batch): the RoPE position of every token, handed to the LM
        as the reference does (asr_modeling.py:517-526 -> TF:models/qwen3/modeling_qwen3.py:386-389, default arange(L)).
        A collator that LEFT-pads (trl's DataCollatorForChatML) must pass the positions it wants: nothing is inferred from the
        attention mask, exactly as in the reference.  ``inputs_embeds`` / ``past_key_values`` / ``use_cache=True`` /
        ``cache_position`` belong to HF's incremental-decoding protocol, which this forward does not speak (``generate`` owns
        the KV cache here): they raise instead of being ignored.

        ``num_items_in_batch``: as in HF Trainer -- loss = sum(nll) / num_items_in_batch (default: the number of
        label tokens in this batch, i.e. the mean).  ``return_logits=False`` skips materialising the [B, L, V]
        logits (the HF Trainer discards them on the training path).  ``label_meta=(rows, targets, n)`` (through ``**kwargs``:
        a NAMED parameter containing "label" would enter HF Trainer's ``label_names`` -- find_labels scans the signature -- and
        Trainer.predict would then treat every batch as unlabelled) lets a
        collator that already knows the label positions on the host avoid one device->host sync.
        """
        dev = self.device_
        if inputs_embeds is not None or past_key_values is not None or use_cache or cache_position is not None:
            bad = [n for n, v in (("inputs_embeds", inputs_embeds), ("past_key_values", past_key_values), ("use_cache", use_cache or None),
                                  ("cache_position", cache_position)) if v is not None]
            raise NotImplementedError(f"ASRModel.forward on MI355X does not take {', '.join(bad)}: the training / eval forward "
                                      "runs from input_ids (+ input_features) without a KV cache; use generate() for decoding")
        if input_ids is None:
            raise ValueError("input_ids is required")
        self._apply_stream_modes()
        ids = input_ids.to(device=dev, dtype=torch.int64).contiguous()
        B, L = ids.shape
        pos = None
        if position_ids is not None:
            pos = torch.as_tensor(position_ids).to(device=dev, dtype=torch.int32)
            if pos.dim() == 1:
                pos = pos[None, :]
            if pos.dim() != 2 or pos.shape[1] != L or pos.shape[0] not in (1, B):
                raise ValueError(f"position_ids must be [B, L] (or [1, L] / [L]); got {tuple(pos.shape)} for input_ids {tuple(ids.shape)}")
            # the RoPE kernels index the cos / sin tables with these: a position outside [0, max_position_embeddings) would read past
            # them (the reference raises an index error there).  One host check on this opt-in path; the default arange needs none.
            max_pos = int(self.config.text_config.max_position_embeddings)
            lo, hi = (int(v) for v in torch.aminmax(pos)) if pos.numel() else (0, 0)
            if lo < 0 or hi >= max_pos:
                raise ValueError(f"position_ids must lie in [0, {max_pos}) (max_position_embeddings); got [{lo}, {hi}]")
            pos = pos.expand(B, L).contiguous().reshape(-1)
        audio, src_row = None, None
        if input_features is None and after_encoder is not None:
            after_encoder()
        if input_features is not None:
            y = self._encode_audio(input_features.to(dev), frame_keep, after_encoder)     # [B, N, D]
            N = y.shape[1]
            if audio_token_counts is None:
                audio_token_counts = (ids == self.audio_token_id).sum(dim=-1)
            counts = audio_token_counts.to(device=dev, dtype=torch.int64).contiguous()
            src_row = ops.audio_index(ids, counts, N, self.audio_token_id)
            audio = y.reshape(B * N, -1)
        kmask = None if attention_mask is None else attention_mask.to(device=dev, dtype=torch.int32).contiguous()
        n_lab, rows, targets = 0, None, None
        if labels is not None:
            label_meta = kwargs.pop("label_meta", None)
            if label_meta is not None:
                rows, targets, n_lab = label_meta
            else:
                rows, targets, n = ops.label_rows(labels.to(device=dev, dtype=torch.int64).contiguous())
                n_lab = int(n.item())                                                      # one host sync
        scale = 1.0 / float(num_items_in_batch if num_items_in_batch is not None else max(n_lab, 1))
        if audio is None:
            audio = torch.zeros((1, self.config.llm_dim), device=dev, dtype=F32)
        loss, nll, logits = FrozenLMLoss.apply(audio, self.language_model, ids, src_row, kmask, rows, targets, n_lab,
                                               scale, bool(return_logits),
                                               *(self.language_model.lora_parameters() or self.language_model.ft_parameters()), pos=pos)
        V = self.config.text_config.vocab_size
        logits = logits.reshape(B, L, -1)[:, :, :V] if return_logits else None
        aux, loss_ce = None, loss
        if labels is None:
            loss = loss_ce = None
        elif hasattr(self.projector, "get_aux_loss"):
            aux = self.projector.get_aux_loss()
            if aux is not None and aux.numel() > 0:
                loss = loss + aux.to(loss.device)                                          # asr_modeling.py:528-531
        return CausalLMOutput(loss=loss, logits=logits, nll=nll[:n_lab] if labels is not None else None,
                              n_label_tokens=n_lab, aux_loss=aux, loss_ce=loss_ce)

    # ------------------------------------------------------------------ checkpoints
    def save_pretrained(self, save_directory, **kwargs):
        """model.safetensors (projector.*) + config.json (+ PEFT adapter files with LoRA), the reference's layout
        (tiny_audio/asr_modeling.py:769-852)."""
        from .checkpoint import save_pretrained
        save_pretrained(self, str(save_directory))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *args, **kwargs):
        """tiny_audio/asr_modeling.py:59-131 for a local directory."""
        from .checkpoint import load_pretrained
        return load_pretrained(cls, str(pretrained_model_name_or_path), **kwargs)

    # ------------------------------------------------------------------ generation
    def _generation_setting(self, name, default, overrides):
        v = overrides.pop(name, None)
        if v is None:
            v = getattr(self.config, name, None)
        return default if v is None else v

    def _get_num_audio_tokens(self, audio_attention_mask):
        """tiny_audio/asr_modeling.py:548-560"""
        lens = self._compute_encoder_output_lengths(audio_attention_mask)
        return int(self.projector.get_output_length(int(lens.max().item())))

    def _prepare_generation(self, input_ids, input_features, audio_attention_mask, attention_mask, system_prompt, kw):
        """Everything ``generate`` and ``generate_streaming`` share (tiny_audio/asr_modeling.py:580-640, :669-716):
        generation settings, audio -> encoder -> projector, the chat prompt with one <audio> placeholder per projected
        frame, and the placeholder -> audio-row index.  -> kwargs of ``Qwen3MI355X.greedy_decode_iter``."""
        if input_features is None:
            raise ValueError("input_features required for generation")
        if audio_attention_mask is None:
            raise ValueError("audio_attention_mask required for generation")
        self._apply_stream_modes()
        max_new = int(self._generation_setting("max_new_tokens", 128, kw))
        if int(self._generation_setting("num_beams", 1, kw)) != 1:
            raise NotImplementedError("num_beams 1 only (the reference's generation config, asr_config.py:103-111); beam search is not built")
        # do_sample / temperature / top_k / top_p (asr_config.py:78-81; round 4): HF's warpers on the device, one multinomial draw per
        # clip and step from a Philox stream keyed by ``seed`` (a generate() keyword here).  Without one every call draws a FRESH key
        # from torch's global generator, so repeated calls on the same audio give different samples and torch.manual_seed() makes a
        # run reproducible -- the behaviour of HF's sampling, which advances the global RNG
        sampling = None
        if bool(self._generation_setting("do_sample", False, kw)):
            temp = self._generation_setting("temperature", None, kw)
            top_k = self._generation_setting("top_k", None, kw)
            top_p = self._generation_setting("top_p", None, kw)
            seed = kw.pop("seed", None)
            sampling = (1.0 if temp is None else float(temp), 0 if top_k is None else int(top_k), 1.0 if top_p is None else float(top_p),
                        int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed))
            if not sampling[0] > 0 or sampling[1] < 0 or not 0 < sampling[2] <= 1:
                raise ValueError("temperature must be > 0, top_k >= 0, 0 < top_p <= 1")
        else:
            for k_ in ("temperature", "top_k", "top_p", "seed"):
                kw.pop(k_, None)
        min_new = int(self._generation_setting("min_new_tokens", 0, kw) or 0)   # HF MinNewTokensLengthLogitsProcessor (round 4)
        # the reference's other two knobs (asr_config.py:84-86): HF logits processors on the device, in front of the argmax
        rep = float(self._generation_setting("repetition_penalty", 1.0, kw))
        ngram = int(self._generation_setting("no_repeat_ngram_size", 0, kw))
        kw.pop("length_penalty", None)                                  # beam-search only: no effect on greedy search
        eos_ids = kw.pop("eos_token_id", None)
        if eos_ids is None:       # <|im_end|> and <|endoftext|> (asr_modeling.py:163-167); the latter is Qwen's pad token
            eos_ids = [self.config.eos_token_id, self.config.pad_token_id]
        eos_ids = [int(e) for e in (eos_ids if isinstance(eos_ids, (list, tuple)) else [eos_ids]) if e is not None]
        pad_id = int(kw.pop("pad_token_id", self.config.pad_token_id))
        dev = self.device_
        feats = input_features.to(dev)
        B = feats.shape[0]
        amask = audio_attention_mask.to(dev)
        enc_len = self._compute_encoder_output_lengths(amask)
        counts = self.projector.get_output_length(enc_len).to(device=dev, dtype=torch.int64).contiguous()
        y = self._encode_audio(feats)                                                   # [B, N, D]
        N = y.shape[1]
        if input_ids is None:
            if self.tokenizer is None:
                raise ValueError("input_ids required: no tokenizer is attached to build the chat prompt")
            from .asr_processing import ASRProcessor
            messages = ASRProcessor.build_messages(self._get_num_audio_tokens(amask), None, system_prompt or self.system_prompt)
            input_ids = ASRProcessor.tokenize_messages(self.tokenizer, messages, add_generation_prompt=True)
            if input_ids.shape[0] == 1 and B > 1:
                input_ids = input_ids.expand(B, -1)
            attention_mask = torch.ones_like(input_ids)
        ids = input_ids.to(device=dev, dtype=torch.int64).contiguous()
        src_row = ops.audio_index(ids, counts, N, self.audio_token_id)
        return dict(input_ids=ids, src_row=src_row, audio=y.reshape(B * N, -1), attention_mask=attention_mask,
                    max_new_tokens=max_new, eos_ids=eos_ids, pad_id=pad_id, repetition_penalty=rep, no_repeat_ngram_size=ngram,
                    min_new_tokens=min_new, sampling=sampling)

    @torch.no_grad()
    def generate(self, input_ids: Optional[torch.Tensor] = None, input_features: Optional[torch.Tensor] = None,
                 audio_attention_mask: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                 system_prompt: Optional[str] = None, **generate_kwargs) -> torch.Tensor:
        """Transcription token ids [B, n_new] (prompt stripped), as ASRModel.generate of the reference
        (tiny_audio/asr_modeling.py:562-646): audio -> encoder -> projector -> <audio> rows of the prompt embeddings ->
        greedy search on the LM (num_beams 1, do_sample False: asr_config.py:103-111), with the
        reference's ``repetition_penalty`` / ``no_repeat_ngram_size`` settings as device-side logits processors."""
        was_training = self.training
        self.eval()
        try:
            args = self._prepare_generation(input_ids, input_features, audio_attention_mask, attention_mask, system_prompt,
                                            dict(generate_kwargs))
            return self.language_model.greedy_decode(**args)
        finally:
            self.train(was_training)

    def generate_streaming(self, input_features: torch.Tensor, audio_attention_mask: torch.Tensor,
                           system_prompt: Optional[str] = None, input_ids: Optional[torch.Tensor] = None,
                           return_token_ids: bool = False, **generate_kwargs) -> Iterator:
        """Partial transcript text, piece by piece, while the clip is still being decoded (ASRModel.generate_streaming,
        tiny_audio/asr_modeling.py:648-760).  The reference runs ``language_model.generate`` on a thread feeding a
        ``TextIteratorStreamer``; here the decode loop itself is a generator (one device sync per token), so no thread
        is needed.  Text is released at word boundaries the way that streamer does (complete lines at once, otherwise
        up to the last space; CJK characters immediately), special tokens are skipped and ``<think>...</think>`` spans
        are dropped (:737-757).  One clip at a time, like the reference's streamer.

        ``return_token_ids=True`` yields the raw int token id of every step instead (no tokenizer needed)."""
        if input_features.shape[0] != 1:
            raise ValueError("generate_streaming handles one clip at a time (TextIteratorStreamer: batch size 1 only)")
        if not return_token_ids and self.tokenizer is None:
            raise ValueError("generate_streaming needs a tokenizer to produce text (or return_token_ids=True)")
        was_training = self.training
        self.eval()
        try:
            with torch.no_grad():
                args = self._prepare_generation(input_ids, input_features, audio_attention_mask, None, system_prompt,
                                                dict(generate_kwargs))
            eos = set(args["eos_ids"])
            pieces = _TextPieces(self.tokenizer) if not return_token_ids else None
            gate = _ThinkGate()
            # (the reference's streaming call passes inputs_embeds only, so HF's repetition / n-gram processors start from an empty
            # input_ids there and see the generated tokens alone -- generate() passes the prompt ids as well)
            for col in self.language_model.greedy_decode_iter(per_token=True, processors_see_prompt=False, **args):
                if col.dim() != 1:                       # the closing full [B, n_new] tensor
                    break
                tok = int(col[0])
                if return_token_ids:
                    yield tok
                    if tok in eos:
                        break
                    continue
                for out in gate.feed(pieces.push(tok)):
                    yield out
                if tok in eos:
                    break
            if pieces is not None:
                for out in gate.feed(pieces.flush()):
                    yield out
                tail = gate.flush()
                if tail:
                    yield tail
        finally:
            self.train(was_training)
