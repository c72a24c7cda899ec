"""Greedy generation, numpy.  TEST INFRASTRUCTURE ONLY.

tiny_audio/asr_modeling.py:562-646 (ASRModel.generate): encode audio, embed the prompt, masked_scatter the
projector rows into the <audio> positions, then ``language_model.generate(input_ids=, inputs_embeds=,
attention_mask=, generation_config=)`` and strip the prompt.  The generation config is greedy
(tiny_audio/asr_config.py:103-111: num_beams 1, do_sample False, repetition_penalty 1.0, no_repeat_ngram_size 0,
min_new_tokens 0, max_new_tokens 128) with eos = [<|im_end|>, <|endoftext|>] (asr_modeling.py:163-168).

HF greedy search (TF:generation/utils.py ``_sample`` with do_sample=False):
    next = argmax(logits[:, -1].float());  next = next * unfinished + pad * (1 - unfinished)
    sequences = cat(sequences, next);  unfinished &= ~isin(next, eos);  stop when unfinished.max() == 0
    or max_new_tokens reached.
No KV cache here: the whole sequence is re-run every step (small configs only); the cache is an optimisation that
does not change the arithmetic.
"""
from __future__ import annotations

import numpy as np

from . import encoder as enc
from . import projectors as proj
from . import qwen3
from .model import gather_audio_embeds, masked_scatter_rows


def prompt_embeds(batch, W, cfg):
    """inputs_embeds of the prompt with the audio rows in place (asr_modeling.py:586-629), eval mode."""
    ids = batch["input_ids"]
    emb = W["lm"]["model.embed_tokens.weight"][ids]
    hs = enc.encoder_forward(batch["input_features"], W["encoder"], cfg["enc"])
    if cfg.get("projector_type", "mlp") == "mlp":
        y, _ = proj.mlp_forward(hs, W["projector"], cfg.get("k", 4))
    else:
        y, _, _ = proj.moe_forward(hs, W["projector"], cfg.get("k", 4), training=False)
    is_audio = ids == cfg["audio_token_id"]
    counts = batch.get("audio_token_counts")
    if counts is None:
        counts = is_audio.sum(-1)
    return masked_scatter_rows(emb, is_audio, gather_audio_embeds(y, counts))


def apply_logits_processors(scores, seq, repetition_penalty=1.0, no_repeat_ngram_size=0):
    """TF:generation/logits_process.py on one step's scores [B, V] given the sequences so far ``seq`` [B, n] (the prompt ids
    followed by the generated tokens: the reference hands input_ids to generate for this, tiny_audio/asr_modeling.py:625-633).
    RepetitionPenaltyLogitsProcessor.__call__: score = gather(scores, seq); score = where(score < 0, score * p, score / p);
    scatter back.  NoRepeatNGramLogitsProcessor.__call__ (_calc_banned_ngram_tokens): if n + 1 >= ngram, every token that
    followed an earlier occurrence of the last ngram-1 tokens is set to -inf."""
    scores = scores.copy()
    B, n = seq.shape
    if repetition_penalty != 1.0:
        for b in range(B):
            sc = scores[b, seq[b]]
            scores[b, seq[b]] = np.where(sc < 0, sc * np.float32(repetition_penalty), sc / np.float32(repetition_penalty))
    g = int(no_repeat_ngram_size)
    if g > 0 and n + 1 >= g:
        for b in range(B):
            prefix = tuple(seq[b, n - (g - 1):]) if g > 1 else ()
            for i in range(n - g + 1):
                if tuple(seq[b, i:i + g - 1]) == prefix:
                    scores[b, seq[b, i + g - 1]] = -np.inf
    return scores


def warp_logits(scores, temperature=1.0, top_k=0, top_p=1.0):
    """HF's sampling warpers in HF's order (TF:generation/logits_process.py TemperatureLogitsWarper, TopKLogitsWarper,
    TopPLogitsWarper; reached with generation_config.do_sample, tiny_audio/asr_config.py:78-81): scores / T; everything below the
    k-th largest score -> -inf (ties at the k-th value stay); ascending sort, softmax, cumulative sum, cumulative <= 1 - top_p -> -inf
    except the last (largest) entry.  float32 throughout, as torch does it."""
    x = np.asarray(scores, np.float32).copy()
    if temperature != 1.0:
        x = (x / np.float32(temperature)).astype(np.float32)
    V = x.shape[-1]
    if 0 < top_k < V:
        kth = np.sort(x, axis=-1)[:, V - top_k][:, None]
        x = np.where(x < kth, np.float32(-np.inf), x)
    if top_p < 1.0:
        order = np.argsort(x, axis=-1, kind="stable")
        srt = np.take_along_axis(x, order, axis=-1)
        e = np.exp(srt - srt[:, -1:], dtype=np.float32)
        probs = (e / e.sum(-1, keepdims=True, dtype=np.float32)).astype(np.float32)
        cum = np.cumsum(probs, axis=-1, dtype=np.float32)
        rm_sorted = cum <= np.float32(1.0 - top_p)
        rm_sorted[:, -1] = False
        rm = np.zeros_like(rm_sorted)
        np.put_along_axis(rm, order, rm_sorted, axis=-1)
        x = np.where(rm, np.float32(-np.inf), x)
    return x


def greedy_generate(batch, W, cfg, max_new_tokens=128, eos_ids=(), pad_id=0, return_margins=False, repetition_penalty=1.0,
                    no_repeat_ngram_size=0, processors_see_prompt=True, min_new_tokens=0):
    """-> generated token ids [B, n_new] (prompt stripped), n_new <= max_new_tokens.  The prompt must be unpadded
    (attention_mask all ones), which is what ASRModel.generate builds.  ``return_margins`` also returns the
    top-1 minus top-2 logit gap of every decision (how robust the argmax is to bf16 rounding)."""
    att = batch.get("attention_mask")
    assert att is None or bool(np.all(att == 1)), "oracle generate: unpadded prompts only"
    x = prompt_embeds(batch, W, cfg)
    B = x.shape[0]
    embed = W["lm"]["model.embed_tokens.weight"]
    unfinished = np.ones(B, dtype=bool)
    out, margins = [], []
    for _ in range(max_new_tokens):
        logits, _ = qwen3.lm_forward(x, np.ones(x.shape[:2], np.int64), W["lm"], cfg["lm"], keep_cache=False,
                                     lora=W.get("lora"), lora_scale=cfg.get("lora_scale", 0.0))
        last = logits[:, -1].astype(np.float32)
        if repetition_penalty != 1.0 or no_repeat_ngram_size > 0:
            # processors_see_prompt=False: the reference's generate_streaming hands language_model.generate inputs_embeds WITHOUT
            # input_ids (tiny_audio/asr_modeling.py:723-729), so HF starts from an empty id sequence: generated tokens only
            head = [np.asarray(batch["input_ids"], np.int64)] if processors_see_prompt else [np.zeros((B, 0), np.int64)]
            seq = np.concatenate(head + [o[:, None] for o in out], axis=1)
            last = apply_logits_processors(last, seq, repetition_penalty, no_repeat_ngram_size)
        if len(out) < min_new_tokens and len(eos_ids):
            # HF MinNewTokensLengthLogitsProcessor (TF:generation/logits_process.py; generation_config.min_new_tokens,
            # tiny_audio/asr_config.py:83): every eos id scores -inf until min_new_tokens tokens have been generated
            last = last.copy()
            last[:, np.asarray(list(eos_ids), dtype=np.int64)] = -np.inf
        nxt = last.argmax(-1)
        srt = np.sort(last, axis=-1)
        margins.append(srt[:, -1] - srt[:, -2])
        nxt = np.where(unfinished, nxt, pad_id)
        out.append(nxt)
        unfinished &= ~np.isin(nxt, np.asarray(list(eos_ids), dtype=np.int64))
        if not unfinished.any():
            break
        x = np.concatenate([x, embed[nxt][:, None, :]], axis=1)
    seq = np.stack(out, axis=1).astype(np.int64)
    return (seq, np.stack(margins, axis=1)) if return_margins else seq
