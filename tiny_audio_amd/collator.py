"""Training collator with the audio side on the GPU.

Batch contract of the reference's ``DataCollator`` (scripts/train.py:240-348): ``input_ids, attention_mask, labels``
(prompt and padding = -100; the assistant's text and its ``<|im_end|>`` unmasked) plus ``input_features,
audio_attention_mask, audio_token_counts``.  Differences by design: the log-mel features come from the ta355 kernel on
the device (the reference computes them in CPU dataloader workers and names that its bottleneck,
configs/experiments/embedded.yaml:37-41), and the chat-ML text collation -- which the reference delegates to
``trl.DataCollatorForChatML`` (not installed here) -- is restated in ``ChatMLTextCollator`` against any tokenizer that
offers ``apply_chat_template``: prompt = all messages but the last rendered with the generation prompt, completion =
the rest of the full rendering; prompt tokens are masked.
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
    """messages -> input_ids / attention_mask / labels with the prompt masked (trl DataCollatorForChatML semantics)."""

    def __init__(self, tokenizer, max_length: int = 2048, padding_side: str = "left", ignore_index: int = -100):
        self.tok, self.max_length, self.padding_side, self.ignore_index = tokenizer, max_length, padding_side, ignore_index
        if getattr(tokenizer, "pad_token_id", None) is None:
            raise ValueError("tokenizer needs a pad token")

    def _render(self, messages, add_generation_prompt):
        ids = self.tok.apply_chat_template(messages, tokenize=True, add_generation_prompt=add_generation_prompt)
        ids = ids["input_ids"] if isinstance(ids, dict) else ids
        return list(ids)

    def __call__(self, examples):
        rows = []
        for ex in examples:
            msgs = ex["messages"]
            prompt = self._render(msgs[:-1], True)
            full = self._render(msgs, False)
            if full[: len(prompt)] != prompt:                      # template re-renders the prefix differently: find the split
                n = 0
                while n < min(len(prompt), len(full)) and prompt[n] == full[n]:
                    n += 1
                prompt = full[:n]
            full = full[: self.max_length]
            labels = [self.ignore_index] * min(len(prompt), len(full)) + full[len(prompt):]
            rows.append((full, labels))
        L = max(len(r[0]) for r in rows)
        pad = int(self.tok.pad_token_id)
        ids = torch.full((len(rows), L), pad, dtype=torch.int64)
        att = torch.zeros((len(rows), L), dtype=torch.int64)
        lab = torch.full((len(rows), L), self.ignore_index, dtype=torch.int64)
        for i, (full, labels) in enumerate(rows):
            sl = slice(L - len(full), L) if self.padding_side == "left" else slice(0, len(full))
            ids[i, sl] = torch.tensor(full); att[i, sl] = 1; lab[i, sl] = torch.tensor(labels)
        return {"input_ids": ids, "attention_mask": att, "labels": lab}


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
