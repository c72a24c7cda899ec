#!/usr/bin/env python3
"""WER of a trained tiny-audio checkpoint on MI355X with the REAL frozen weights -- the one-command recipe for a box that
has the checkpoints (this build container is offline: no pretrained weights, no tokenizer files, so the +-0.5 WER criterion
of BASELINE.json cannot be measured here).

    python scripts/eval_real_weights.py \
        --audio-model-dir /models/GLM-ASR-Nano-2512 --text-model-dir /models/Qwen3-0.6B \
        --checkpoint /models/tiny-audio --manifest test.jsonl [--whisper-dir /models/whisper-tiny] [--reference-wer 6.1]

manifest: one JSON object per line, {"audio": "<16 kHz mono wav path>", "text": "<reference transcript>"}.
Mirrors scripts/eval/evaluators/base.py:100-150 of the reference: greedy generate (ASRModel.generate), the pipeline's text
post-processing (<think> stripping, repetition truncation: tiny_audio/asr_pipeline.py:232-330), Whisper's
EnglishTextNormalizer + the project's three spelling fixes on both sides (scripts/eval/audio.py:59-96), corpus WER.
"""
import argparse
import json
import os
import sys
import wave

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def read_wav(path):
    with wave.open(path, "rb") as w:
        if w.getframerate() != 16000 or w.getsampwidth() != 2:
            raise ValueError(f"{path}: need 16 kHz 16-bit PCM")
        x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
        return x.reshape(-1, w.getnchannels()).mean(axis=1) if w.getnchannels() > 1 else x


def make_normalizer(whisper_dir):
    """scripts/eval/audio.py:59-96."""
    from transformers.models.whisper.english_normalizer import EnglishTextNormalizer
    spelling = {}
    if whisper_dir:
        with open(os.path.join(whisper_dir, "normalizer.json")) as f:
            spelling = json.load(f)
    base = EnglishTextNormalizer(spelling)
    fixes = {"okay": "ok", "all right": "alright", "kinda": "kind of"}

    def norm(text):
        text = base(text)
        for a, b in fixes.items():
            text = text.replace(a, b)
        return text
    return norm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--audio-model-dir", required=True)
    ap.add_argument("--text-model-dir", required=True)
    ap.add_argument("--checkpoint", required=True, help="directory with model.safetensors + config.json (+ adapter files)")
    ap.add_argument("--manifest", required=True)
    ap.add_argument("--whisper-dir", default=None, help="openai/whisper-tiny snapshot (normalizer.json: British->American spellings)")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--max-new-tokens", type=int, default=128)
    ap.add_argument("--reference-wer", type=float, default=None, help="WER (percent) of the reference run on the same manifest")
    a = ap.parse_args()
    from transformers import AutoTokenizer
    from tiny_audio_amd import hub_weights
    from tiny_audio_amd.asr_modeling import ASRModel
    from tiny_audio_amd.asr_processing import ASRProcessor, LogMelFeatureExtractor
    from tiny_audio_amd.eval_text import postprocess_tokens, word_error_rate

    tok = AutoTokenizer.from_pretrained(a.text_model_dir)
    if tok.convert_tokens_to_ids("<audio>") in (None, tok.unk_token_id):         # tiny_audio/asr_modeling.py:160-171
        tok.add_special_tokens({"additional_special_tokens": ["<audio>"]})
    model = ASRModel.from_pretrained(a.checkpoint, device="cuda", init="none", tokenizer=tok,
                                     encoder_state_dict=hub_weights.encoder_state_dict(a.audio_model_dir),
                                     lm_state_dict=hub_weights.lm_state_dict(a.text_model_dir))
    model.eval()
    fe = LogMelFeatureExtractor(128, "cuda")
    proc = model.get_processor()
    rows = [json.loads(l) for l in open(a.manifest) if l.strip()]
    norm = make_normalizer(a.whisper_dir)
    eos = [tok.convert_tokens_to_ids("<|im_end|>"), tok.convert_tokens_to_ids("<|endoftext|>")]
    refs, hyps = [], []
    order = sorted(range(len(rows)), key=lambda i: os.path.getsize(rows[i]["audio"]))      # batches of similar length
    for s in range(0, len(order), a.batch):
        idx = order[s:s + a.batch]
        f = fe([read_wav(rows[i]["audio"]) for i in idx], sampling_rate=16000)
        out = model.generate(input_features=f["input_features"], audio_attention_mask=f["attention_mask"],
                             max_new_tokens=a.max_new_tokens).cpu().tolist()
        for i, toks in zip(idx, out):
            hyps.append(postprocess_tokens(toks, [e for e in eos if e is not None], lambda t: tok.decode(t, skip_special_tokens=True)))
            refs.append(rows[i]["text"])
    wer = 100.0 * word_error_rate(refs, hyps, normalize=norm)
    rec = {"wer_percent": round(wer, 3), "utterances": len(refs)}
    if a.reference_wer is not None:
        rec.update(reference_wer_percent=a.reference_wer, within_0p5=abs(wer - a.reference_wer) <= 0.5)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
