"""Frozen-model weights from local snapshots of their hub repositories (no network: ``HF_HUB_OFFLINE``).

The reference fetches the two frozen models by id at construction (tiny_audio/asr_modeling.py:203-254):
``AutoModelForSeq2SeqLM.from_pretrained("zai-org/GLM-ASR-Nano-2512").audio_tower`` and
``AutoModelForCausalLM.from_pretrained("Qwen/Qwen3-0.6B")``, then ``resize_token_embeddings(len(tokenizer))`` after the
``<audio>`` token is added.  Here the same tensors are read straight from the snapshot directories' safetensors files
(single file or sharded with ``model.safetensors.index.json``) and handed to ``load_state_dict_hf`` -- which also performs
the resize (first ``vocab_size`` rows of ``embed_tokens``).
"""
from __future__ import annotations

import json
import os
from typing import Callable, Dict, Iterable, Optional

import torch

_AUDIO_PREFIXES = ("audio_tower.", "model.audio_tower.", "audio_encoder.", "encoder.")


def _shards(path: str) -> Iterable[str]:
    idx = os.path.join(path, "model.safetensors.index.json")
    if os.path.exists(idx):
        with open(idx) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
        return [os.path.join(path, f) for f in files]
    one = os.path.join(path, "model.safetensors")
    if os.path.exists(one):
        return [one]
    files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"no safetensors weights under {path}")
    return [os.path.join(path, f) for f in files]


def read_tensors(path: str, keep: Optional[Callable[[str], bool]] = None) -> Dict[str, torch.Tensor]:
    """Every tensor (fp32, CPU) of a snapshot directory whose name passes ``keep``."""
    from safetensors import safe_open
    out = {}
    for file in _shards(path):
        with safe_open(file, framework="pt", device="cpu") as f:
            for k in f.keys():
                if keep is None or keep(k):
                    out[k] = f.get_tensor(k).to(torch.float32)
    return out


def encoder_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """GlmAsrEncoder state dict (``conv1.weight``, ``layers.N.self_attn.q_proj.weight`` ...) from a GLM-ASR snapshot: the
    tensors under the model's ``audio_tower`` (tiny_audio/asr_modeling.py:221-231 keeps only that sub-module)."""
    sd = read_tensors(path, lambda k: k.startswith(_AUDIO_PREFIXES) or k.startswith(("conv1.", "conv2.", "layers.", "norm.")))
    out = {}
    for k, v in sd.items():
        for p in _AUDIO_PREFIXES:
            if k.startswith(p):
                k = k[len(p):]
                break
        out[k] = v
    if "conv1.weight" not in out:
        raise KeyError(f"{path}: no audio_tower.conv1.weight -- not a GLM-ASR checkpoint?")
    return out


def lm_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """Qwen3ForCausalLM state dict (``model.embed_tokens.weight`` ... ; a tied ``lm_head.weight`` is dropped).  The embedding
    keeps the checkpoint's row count; ``Qwen3MI355X.load_state_dict_hf`` cuts it to the tokenizer's vocabulary."""
    sd = read_tensors(path, lambda k: k.startswith("model.") or k == "lm_head.weight")
    head = sd.pop("lm_head.weight", None)
    if "model.embed_tokens.weight" not in sd:
        raise KeyError(f"{path}: no model.embed_tokens.weight -- not a causal-LM checkpoint?")
    # Qwen3MI355X ties the output head to embed_tokens (Qwen3-0.6B / 1.7B: tie_word_embeddings = true).  A checkpoint with an
    # UNTIED head (the larger Qwen3 models) would load silently and produce wrong logits: refuse it.
    tied = _config_says_tied(path)
    if tied is False or (head is not None and (head.shape != sd["model.embed_tokens.weight"].shape
                                               or not torch.equal(head, sd["model.embed_tokens.weight"]))):
        raise ValueError(f"{path}: lm_head.weight is not tied to model.embed_tokens.weight (tie_word_embeddings = false); "
                         "Qwen3MI355X implements the tied head only")
    return sd


def _config_says_tied(path: str):
    """tie_word_embeddings of the snapshot's config.json (top level or text_config); None when there is no config."""
    import json
    cfg = os.path.join(path, "config.json")
    if not os.path.exists(cfg):
        return None
    with open(cfg) as fh:
        c = json.load(fh)
    v = c.get("tie_word_embeddings", c.get("text_config", {}).get("tie_word_embeddings"))
    return None if v is None else bool(v)
