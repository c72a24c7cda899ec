"""Synthetic training batches of SURVEY.md section 8(d) for bench.py and the scripts (no tokenizer / dataset offline).

The token stream has the shape ``DataCollator.__call__`` (scripts/train.py:324-348) produces for a transcription
sample: ``[prompt prefix | <audio> * n | prompt suffix | transcript | <|im_end|> | pad ...]`` with labels -100
everywhere except the transcript and the closing ``<|im_end|>``.  The label positions are known on the host, as
they are to a collator that builds the labels on the CPU.
"""
from __future__ import annotations

import numpy as np


def token_batch(B, n_audio, vocab, audio_id, pad_id, eos_id, L=None, n_prefix=3, n_suffix=24, n_text=35):
    """-> (input_ids [B,L] i64, attention_mask [B,L] i64, labels [B,L] i64, audio_token_counts [B] i64,
    n_label_tokens int).  Clip b draws its ids from ``RandomState(99 + b)``."""
    need = n_prefix + n_audio + n_suffix + n_text + 1
    L = L or need
    if L < need:
        raise ValueError(f"seq_len {L} < {need} tokens of one sample")
    ids = np.full((B, L), pad_id, dtype=np.int64)
    att = np.zeros((B, L), dtype=np.int64)
    lab = np.full((B, L), -100, dtype=np.int64)
    hi = min(vocab, audio_id) - 30
    a0, s0, t0 = n_prefix, n_prefix + n_audio, n_prefix + n_audio + n_suffix
    for b in range(B):
        rng = np.random.RandomState(99 + b)
        ids[b, :a0] = rng.randint(0, hi, n_prefix)
        ids[b, a0:s0] = audio_id
        ids[b, s0:t0] = rng.randint(0, hi, n_suffix)
        ids[b, t0:t0 + n_text] = rng.randint(0, hi, n_text)
        ids[b, t0 + n_text] = eos_id
        lab[b, t0:t0 + n_text + 1] = ids[b, t0:t0 + n_text + 1]
        att[b, :t0 + n_text + 1] = 1
    # shifted labels: position p predicts token p+1, so every labelled token except one at column 0 is a target
    n_label = int((lab[:, 1:] != -100).sum())
    return ids, att, lab, np.full(B, n_audio, dtype=np.int64), n_label


def synthetic_wave(b: int, n: int = 160000) -> np.ndarray:
    """wav[b] = 0.1 * N(0, 1) from ``RandomState(1234 + b)``, float32 -- the recipe of the reference's
    tests/test_data_collator.py:68-75 (the oracle keeps its own copy of this two-liner for the tests)."""
    return (0.1 * np.random.RandomState(1234 + b).standard_normal(n)).astype(np.float32)
