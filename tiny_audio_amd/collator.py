"""Training collator with the audio side on the GPU.

Batch contract of the reference's ``DataCollator`` (scripts/train.py:240-348): ``input_ids, attention_mask, labels``
(prompt and padding = -100; the assistant's text and its ``<|im_end|>`` unmasked) plus ``input_features,
audio_attention_mask, audio_token_counts``.  Differences by design: the log-mel features come from the ta355 kernel on
the device (the reference computes them in CPU dataloader workers and names that its bottleneck,
configs/experiments/embedded.yaml:37-41), and the chat-ML text collation -- which the reference delegates to
``trl.DataCollatorForChatML`` (not installed here) -- is restated in ``ChatMLTextCollator`` against any tokenizer that
offers ``apply_chat_template``: prompt = all messages but the last rendered with the generation prompt, the message = all of
them; the first len(prompt tokens) labels are masked; everything is left-padded; trl's ``prompts`` / ``prompt_attention_mask``
keys are emitted too.
"""
from __future__ import annotations

import re
from typing import Any, Callable, Optional

import numpy as np
import torch

from .asr_config import DEFAULT_ENCODER_CONV_LAYERS, compute_encoder_output_length

TRANSCRIBE_PROMPT = "Transcribe the speech to text"          # scripts/train.py:52
DESCRIBE_PROMPT = "Describe all the information you can hear"
MAX_AUDIO_SECONDS = 30.0                                      # scripts/train.py:269-272

# Annotation markers that only the training splits carry (scripts/train.py:55-76): Gigaspeech punctuation tags, TEDLIUM
# <unk>, EdAcc / Earnings22 event tags; TEDLIUM's bracketed editorial notes.
_MARKERS = ("comma", "period", "exclamationpoint", "questionmark", "sil", "music", "noise", "other", "unk", "overlap",
            "laugh", "dtmf", "foreign", "no-speech", "lipsmack", "clear_throat", "inaudible", "crosstalk")
_MARKER_RE = re.compile(r"\s*<(?:" + "|".join(re.escape(m) for m in _MARKERS) + r")>", re.IGNORECASE)
_BRACKET_RE = re.compile(r"\s*\[[^\]]*\]")


def normalize_label(raw_text: Optional[str]) -> str:
    """scripts/train.py:79-97: lowercase, drop corpus markers and bracketed notes, '%' -> ' percent', collapse blanks."""
    text = (raw_text or "").strip().lower()
    text = _BRACKET_RE.sub("", _MARKER_RE.sub("", text))
    text = text.replace("%", " percent").replace("per cent", "percent")
    return " ".join(text.split())


class ChatMLTextCollator:
    """messages -> ``input_ids, attention_mask, labels, prompts, prompt_attention_mask``: the batch of trl 0.29.1's
    ``DataCollatorForChatML`` (pinned in the reference's poetry.lock:7213-7214, called at scripts/train.py:265,344), restated
    step by step since trl is not installable here:

    1. prompt text  = chat template over all messages but the last, WITH the generation prompt, rendered to a string;
       message text = chat template over all messages, without it;
    2. both strings are tokenised separately with ``add_special_tokens=False`` (message truncated to ``max_length``, prompt to the
       message's token count);
    3. labels = ``ignore_index`` for the first ``len(prompt tokens)`` positions, the message's own ids after it -- the split is
       the prompt's TOKEN COUNT, not a search for a common prefix;
    4. all five tensors are LEFT-padded (ids with ``pad_token_id``, masks with 0, labels with ``ignore_index``), whatever the
       tokenizer's ``padding_side`` says.

    The extra keys ``prompts`` / ``prompt_attention_mask`` reach ``ASRModel.forward`` through ``**kwargs`` and are ignored there,
    as in the reference.  A tokenizer object that cannot be called on a string (the whitespace stub of the host tests) is driven
    through ``apply_chat_template(tokenize=True)`` instead; same steps 3-4.  Pinned on ``tests/golden/chatml_collation.json``
    (transformers' own ``apply_chat_template`` + fast-tokenizer outputs for a ChatML template)."""

    def __init__(self, tokenizer, max_length: Optional[int] = 2048, padding_side: str = "left", ignore_index: int = -100):
        self.tok, self.max_length, self.padding_side, self.ignore_index = tokenizer, max_length, padding_side, ignore_index
        if getattr(tokenizer, "pad_token_id", None) is None:
            raise ValueError("tokenizer needs a pad token")

    def _encode(self, messages, add_generation_prompt, max_length):
        """-> (ids, attention mask) of one rendering."""
        tok = self.tok
        if callable(tok):
            text = tok.apply_chat_template(messages, tokenize=False, add_generation_prompt=add_generation_prompt)
            enc = tok(text, truncation=True, max_length=max_length, padding=False, return_tensors=None, add_special_tokens=False)
            ids = list(enc["input_ids"])
            return ids, list(enc["attention_mask"]) if "attention_mask" in enc else [1] * len(ids)
        ids = tok.apply_chat_template(messages, tokenize=True, add_generation_prompt=add_generation_prompt)
        ids = list(ids["input_ids"] if isinstance(ids, dict) else ids)
        if max_length is not None:
            ids = ids[:max_length]
        return ids, [1] * len(ids)

    def _pad(self, rows, value):
        L = max(len(r) for r in rows)
        out = torch.full((len(rows), L), value, dtype=torch.int64)
        for i, r in enumerate(rows):
            if r:
                out[i, (L - len(r) if self.padding_side == "left" else 0):(L if self.padding_side == "left" else len(r))] = torch.tensor(r)
        return out

    def __call__(self, examples):
        ids, att, p_ids, p_att, lab = [], [], [], [], []
        for ex in examples:
            msgs = ex["messages"]
            full, full_mask = self._encode(msgs, False, self.max_length)
            prompt, prompt_mask = self._encode(msgs[:-1], True, len(full))
            ids.append(full); att.append(full_mask); p_ids.append(prompt); p_att.append(prompt_mask)
            lab.append([self.ignore_index] * min(len(prompt), len(full)) + full[len(prompt):])
        pad = int(self.tok.pad_token_id)
        return {"input_ids": self._pad(ids, pad), "attention_mask": self._pad(att, 0), "labels": self._pad(lab, self.ignore_index),
                "prompts": self._pad(p_ids, pad), "prompt_attention_mask": self._pad(p_att, 0)}


class DataCollator:
    """Drop-in for scripts/train.py:DataCollator with the feature extraction on the device."""

    def __init__(self, tokenizer: Any, feature_extractor: Any, sample_rate: int, system_prompt: Optional[str] = None,
                 projector: Any = None, encoder_conv_layers: Optional[list] = None, text_collator: Optional[Callable] = None):
        self.tokenizer, self.feature_extractor, self.sample_rate = tokenizer, feature_extractor, sample_rate
        self.system_prompt, self.projector = system_prompt, projector
        self.encoder_conv_layers = encoder_conv_layers or DEFAULT_ENCODER_CONV_LAYERS
        self.text_collator = text_collator or ChatMLTextCollator(tokenizer, max_length=2048)

    def _extract_audio_arrays(self, features):
        """The reference's row filter (scripts/train.py:274-311): drop empty / non-finite audio, labels that normalise to
        nothing, clips longer than 30 s; stereo is averaged."""
        arrays, kept = [], []
        for f in features:
            try:
                audio = f["audio"]["array"]
                audio = audio.numpy() if hasattr(audio, "numpy") else np.asarray(audio)
                audio = audio.squeeze()
                if audio.ndim > 1:
                    audio = audio.mean(axis=0)
                ok = (audio.size > 0 and bool(np.isfinite(audio).all()) and bool(normalize_label(f.get("text") or ""))
                      and audio.size / self.sample_rate <= MAX_AUDIO_SECONDS)
                if ok:
                    arrays.append(audio.astype(np.float32)); kept.append(f)
            except Exception:  # noqa: BLE001  (a malformed row is dropped, as in the reference)
                continue
            finally:
                f["audio"] = None
        if not arrays:
            raise ValueError("No valid audio samples in batch")
        return arrays, kept

    def _make_messages(self, num_audio_tokens: int, prompt: str, response: str) -> dict:
        messages = []
        if self.system_prompt:
            messages.append({"role": "system", "content": self.system_prompt})
        messages.append({"role": "user", "content": "<audio>" * num_audio_tokens + " " + prompt})
        messages.append({"role": "assistant", "content": response})
        return {"messages": messages}

    def _build_sample(self, feature: dict, num_audio_tokens: int) -> dict:
        return self._make_messages(num_audio_tokens, TRANSCRIBE_PROMPT, normalize_label(feature.get("text") or ""))

    def __call__(self, features):
        arrays, kept = self._extract_audio_arrays(features)
        audio = self.feature_extractor(arrays, sampling_rate=self.sample_rate, padding="longest", return_attention_mask=True,
                                       return_tensors="pt")
        mel_lengths = audio["attention_mask"].sum(dim=-1)
        enc_lengths = compute_encoder_output_length(mel_lengths, self.encoder_conv_layers)
        counts = self.projector.get_output_length(enc_lengths).to(torch.long)
        batch = self.text_collator([self._build_sample(f, n) for f, n in zip(kept, counts.tolist())])
        batch["input_features"] = audio["input_features"]
        batch["audio_attention_mask"] = audio["attention_mask"]
        batch["audio_token_counts"] = counts
        return batch


class MultiTaskDataCollator(DataCollator):
    """scripts/train.py:351-365: ASR + SIFT rows, no system prompt."""

    def __init__(self, *args, **kwargs):
        kwargs["system_prompt"] = ""
        super().__init__(*args, **kwargs)

    def _build_sample(self, feature: dict, num_audio_tokens: int) -> dict:
        if feature.get("task") == "sift":
            return self._make_messages(num_audio_tokens, DESCRIBE_PROMPT, (feature.get("sift_response") or feature.get("text") or "").strip())
        return self._make_messages(num_audio_tokens, TRANSCRIBE_PROMPT, (feature.get("text") or "").strip().lower())
