"""Checkpoint interchange with the reference.

What the reference writes (tiny_audio/asr_modeling.py:398-422, 769-852) and reads back (:59-131):

* ``model.safetensors``  -- ``ASRModel.state_dict()``: the trainable weights only, ``projector.*`` (the frozen encoder
  and LM are re-fetched from their own hub repos by id);
* ``config.json``        -- ``ASRConfig`` incl. the nested ``audio_config`` / ``text_config``;
* ``adapter_model.safetensors`` + ``adapter_config.json`` -- the PEFT LoRA adapter when ``use_lora`` (peft 0.19.1:
  keys ``base_model.model.model.layers.N.<module>.lora_{A,B}.weight``, adapter name stripped on save).

Here the same files are produced / consumed for the MI355X model, so a projector (and adapter) trained on either side
loads on the other.  Tokenizer / feature-extractor files and the copied ``asr_*.py`` sources of the reference's
``save_pretrained`` are Hub packaging, not part of the hot path, and are not written.
"""
from __future__ import annotations

import json
import os
from typing import Optional

import torch

from .asr_config import ASRConfig

MODEL_FILE, CONFIG_FILE = "model.safetensors", "config.json"
ADAPTER_FILE, ADAPTER_CONFIG_FILE = "adapter_model.safetensors", "adapter_config.json"
_PEFT_PREFIX = "base_model.model.model."


def _jsonable(v):
    if isinstance(v, (str, int, float, bool)) or v is None:
        return v
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, dict):
        return {str(k): _jsonable(x) for k, x in v.items()}
    return str(v)


def config_to_json(config: ASRConfig) -> dict:
    d = {k: _jsonable(v) for k, v in config.to_dict().items()}
    d.update(model_type="asr_model", architectures=["ASRModel"], vocab_size=config.text_config.vocab_size)
    return d


def config_from_json(d: dict) -> ASRConfig:
    d = dict(d)
    for k in ("model_type", "architectures", "transformers_version", "auto_map", "custom_pipelines", "encoder"):
        d.pop(k, None)                 # HF bookkeeping; "encoder" is the reference's alias of audio_config
    return ASRConfig(**d)


def save_pretrained(model, save_directory: str) -> None:
    from safetensors.torch import save_file
    os.makedirs(save_directory, exist_ok=True)
    lm = model.language_model
    sd = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in model.state_dict().items() if ".lora_" not in k}
    save_file(sd, os.path.join(save_directory, MODEL_FILE), metadata={"format": "pt"})
    with open(os.path.join(save_directory, CONFIG_FILE), "w") as f:
        json.dump(config_to_json(model.config), f, indent=2, sort_keys=True)
    if lm.lora_rank:
        ad = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in lm.export_lora_state_dict(prefix=_PEFT_PREFIX).items()}
        save_file(ad, os.path.join(save_directory, ADAPTER_FILE), metadata={"format": "pt"})
        cfg = model.config
        with open(os.path.join(save_directory, ADAPTER_CONFIG_FILE), "w") as f:
            json.dump({"peft_type": "LORA", "task_type": "CAUSAL_LM", "r": int(cfg.lora_rank), "lora_alpha": int(cfg.lora_alpha),
                       "lora_dropout": float(cfg.lora_dropout), "target_modules": list(cfg.lora_target_modules), "bias": "none",
                       "fan_in_fan_out": False, "inference_mode": True, "base_model_name_or_path": ""}, f, indent=2)


def load_pretrained(model_cls, pretrained_path: str, device="cuda", config: Optional[ASRConfig] = None, init="random",
                    encoder_state_dict=None, lm_state_dict=None, **kwargs):
    """Build the model from ``config.json`` and overlay ``model.safetensors`` (strict=False, as the reference does) and
    the PEFT adapter.  The frozen encoder / LM weights are separate checkpoints in the reference (hub ids in the
    config); pass their HF state dicts via ``encoder_state_dict`` / ``lm_state_dict``, otherwise they are randomly
    initialised (``init``)."""
    from safetensors.torch import load_file
    if config is None:
        with open(os.path.join(pretrained_path, CONFIG_FILE)) as f:
            config = config_from_json(json.load(f))
    model = model_cls(config, device=device, init=init, **kwargs)
    if encoder_state_dict is not None:
        model.audio_tower.load_state_dict_hf(encoder_state_dict)
    if lm_state_dict is not None:
        model.language_model.load_state_dict_hf(lm_state_dict)
    mf = os.path.join(pretrained_path, MODEL_FILE)
    if os.path.exists(mf):
        sd = load_file(mf)
        lm_keys = [k for k in sd if k.startswith("language_model.") and ".lora_" not in k]
        if lm_keys and not model.language_model.train_base:
            # a fully fine-tuned decoder loaded for inference / as a frozen LM: its weights replace the base LM's
            # (tiny_audio/asr_modeling.py:96-106 overlays them on the freshly built base model)
            model.language_model.load_state_dict_hf({k[len("language_model."):]: v.to(torch.float32) for k, v in sd.items()
                                                     if k in lm_keys})
        model.load_state_dict({k: v.to(torch.float32) for k, v in sd.items()}, strict=False)
    af, ac = os.path.join(pretrained_path, ADAPTER_FILE), os.path.join(pretrained_path, ADAPTER_CONFIG_FILE)
    if getattr(config, "use_lora", False) and os.path.exists(ac):
        with open(ac) as f:
            a = json.load(f)
        if int(a.get("r", config.lora_rank)) != model.language_model.lora_rank:
            raise ValueError("adapter rank differs from config.lora_rank")
        have = {t.split(".")[-1] for t in model.language_model.lora_targets}
        if "target_modules" in a and {t.split(".")[-1] for t in a["target_modules"]} != have:
            raise ValueError("adapter target_modules differ from config.lora_target_modules")
        model.language_model.lora_alpha = int(a.get("lora_alpha", config.lora_alpha))
        model.language_model._finalize_lora_scale()
        model.language_model.load_lora_state_dict(load_file(af))
    return model
