"""ASRModel.forward + backward + one optimizer step (rows a4,a7,a8,a13 of
SURVEY.md section 8), numpy.

tiny_audio/asr_modeling.py:27-44  (_gather_audio_embeds)
tiny_audio/asr_modeling.py:434-456 (_encode_audio), :458-479 (frame dropout)
tiny_audio/asr_modeling.py:481-533 (forward: embed, masked_scatter, LM, + aux loss)
HF Trainer step semantics: TF:trainer.py:2040-2050,2141-2201 (sum-CE / global
label-token count), clip_grad_norm_(1.0), torch.optim.AdamW
(configs/training/production.yaml:5-9).
"""
from __future__ import annotations

import numpy as np

from . import encoder as enc
from . import projectors as proj
from . import qwen3


def gather_audio_embeds(audio_embeds, token_counts):
    """tiny_audio/asr_modeling.py:27-44: first counts[i] rows of sample i,
    zero-padded when counts[i] > rows available; packed [sum(counts), D]."""
    B, N, D = audio_embeds.shape
    counts = np.asarray(token_counts, dtype=np.int64)
    need = int(counts.max()) if counts.size else 0
    if need > N:
        audio_embeds = np.concatenate(
            [audio_embeds, np.zeros((B, need - N, D), dtype=audio_embeds.dtype)], axis=1)
        N = need
    mask = np.arange(N)[None, :] < counts[:, None]
    return audio_embeds[mask]


def masked_scatter_rows(inputs_embeds, is_audio, packed):
    """tiny_audio/asr_modeling.py:511-515: rows where input_ids == <audio> are
    replaced, in row-major order, by consecutive rows of ``packed``."""
    out = inputs_embeds.copy()
    idx = np.argwhere(is_audio)
    assert idx.shape[0] == packed.shape[0], (idx.shape, packed.shape)
    out[idx[:, 0], idx[:, 1]] = packed
    return out


def asr_forward(batch, W, cfg, keep_cache=True, frame_keep_mask=None, moe_noise=None, training=False,
                num_items_in_batch=None):
    """batch: input_ids, attention_mask, labels, input_features, audio_token_counts (+ optional position_ids [B, L] or [L],
    handed to the LM as tiny_audio/asr_modeling.py:517-526 does).
    W: dict(encoder=..., projector=..., lm=...); cfg: dict(enc=..., lm=...,
    projector_type, k, audio_token_id, hidden..).
    ``frame_keep_mask`` [B, S] is the injected Bernoulli keep mask of
    _maybe_drop_audio_tokens (zeros whole frames, no rescale)."""
    ids = batch["input_ids"]
    emb = W["lm"]["model.embed_tokens.weight"][ids]                       # :498
    hs = enc.encoder_forward(batch["input_features"], W["encoder"], cfg["enc"])   # :448-450
    if frame_keep_mask is not None:
        hs = hs * frame_keep_mask[:, :, None].astype(hs.dtype)            # :458-479
    aux = np.float32(0.0)
    if cfg.get("projector_type", "mlp") == "mlp":
        y, pc = proj.mlp_forward(hs, W["projector"], cfg.get("k", 4))
    else:
        y, aux, pc = proj.moe_forward(hs, W["projector"], cfg.get("k", 4), training=training,
                                      jitter_noise=moe_noise,
                                      aux_coef=cfg.get("router_aux_loss_coef", 0.01))
    counts = batch.get("audio_token_counts")
    is_audio = ids == cfg["audio_token_id"]
    if counts is None:
        counts = is_audio.sum(-1)
    packed = gather_audio_embeds(y, counts)
    x0 = masked_scatter_rows(emb, is_audio, packed)
    logits, lc = qwen3.lm_forward(x0, batch.get("attention_mask"), W["lm"], cfg["lm"], position_ids=batch.get("position_ids"),
                                  keep_cache=keep_cache, lora=W.get("lora"), lora_scale=cfg.get("lora_scale", 0.0))
    out = dict(logits=logits, audio_embeds=y, encoder_out=hs, inputs_embeds=x0, aux_loss=aux)
    if batch.get("labels") is not None:
        ce, dlogits, n_tok = qwen3.causal_lm_loss(logits, batch["labels"], num_items_in_batch)
        out.update(ce_loss=ce, loss=np.float32(ce + aux), n_label_tokens=n_tok)
        out["_dlogits"] = dlogits
    out["_cache"] = dict(pc=pc, lc=lc, counts=np.asarray(counts), is_audio=is_audio, y_shape=y.shape)
    return out


def asr_backward(out, W, cfg, batch=None):
    """Gradients of out['loss'] w.r.t. the projector parameters (encoder and LM frozen); with W['lora'] the adapter
    gradients are returned under 'lora.<name>' keys as well (stage-2 training, BASELINE configs[4]).
    With cfg['freeze_language_model'] False (full decoder fine-tuning, configs/experiments/embedded.yaml:23) every LM
    weight gradient is returned under 'language_model.<name>'; ``batch`` then supplies the input ids for the embedding
    lookup's share of the tied embedding matrix (tiny_audio/asr_modeling.py:498)."""
    c = out["_cache"]
    lg = {} if W.get("lora") is not None else None
    wg = {} if not cfg.get("freeze_language_model", True) else None
    dx0 = qwen3.lm_backward_dx(out["_dlogits"], W["lm"], cfg["lm"], c["lc"], W.get("lora"), cfg.get("lora_scale", 0.0), lg, wg)
    if wg is not None:
        ids = np.asarray(batch["input_ids"])
        text = ~c["is_audio"]                               # audio rows were overwritten by masked_scatter: no embedding grad
        ge = wg["model.embed_tokens.weight"]
        np.add.at(ge, ids[text], dx0[text])
    idx = np.argwhere(c["is_audio"])
    dpacked = dx0[idx[:, 0], idx[:, 1]]
    B, N, D = c["y_shape"]
    dy = np.zeros((B, max(N, int(c["counts"].max())), D), dtype=np.float32)
    mask = np.arange(dy.shape[1])[None, :] < c["counts"][:, None]
    dy[mask] = dpacked
    dy = dy[:, :N]
    if cfg.get("projector_type", "mlp") == "mlp":
        grads = proj.mlp_backward(dy, W["projector"], c["pc"])
    else:
        grads = proj.moe_backward(dy, W["projector"], c["pc"], d_aux=1.0)
    if lg is not None:
        grads.update({"lora." + k: v for k, v in lg.items()})
    if wg is not None:
        grads.update({"language_model." + k: v for k, v in wg.items()})
    return grads, dx0


def clip_grad_norm(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_: g *= max_norm / (total_norm + 1e-6), clamped to 1."""
    total = float(np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values())))
    coef = min(1.0, max_norm / (total + 1e-6))
    return {k: (g * np.float32(coef)).astype(np.float32) for k, g in grads.items()}, total


def adamw_step(params, grads, state, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, no_decay=()):
    """torch.optim.AdamW (decoupled decay); ``no_decay`` names get wd 0 like
    the norm/bias group of ASRTrainer.create_optimizer (scripts/train.py:427-432)."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    for k, p in params.items():
        g = grads[k].astype(np.float32)
        m = state.setdefault("m." + k, np.zeros_like(p))
        v = state.setdefault("v." + k, np.zeros_like(p))
        wd = 0.0 if (k in no_decay or k.endswith("bias") or "norm" in k) else weight_decay
        lr_k = lr(k) if callable(lr) else lr             # per-parameter LR: the split groups of scripts/train.py:384-437
        p *= np.float32(1.0 - lr_k * wd)
        m *= np.float32(b1); m += np.float32(1 - b1) * g
        v *= np.float32(b2); v += np.float32(1 - b2) * g * g
        mhat = m / np.float32(1 - b1 ** t)
        vhat = v / np.float32(1 - b2 ** t)
        p -= np.float32(lr_k) * mhat / (np.sqrt(vhat) + np.float32(eps))
    return params


def train_step(batch, W, cfg, state, lr=1e-3, max_grad_norm=1.0, weight_decay=0.0, decoder_lr=None, **fw):
    """One optimizer step.  With cfg['freeze_language_model'] False the LM weights train too, at ``decoder_lr``
    (configs/experiments/embedded.yaml:23-25)."""
    out = asr_forward(batch, W, cfg, training=True, **fw)
    grads, _ = asr_backward(out, W, cfg, batch)
    grads, gnorm = clip_grad_norm(grads, max_grad_norm)
    if cfg.get("freeze_language_model", True):
        adamw_step(W["projector"], grads, state, lr, weight_decay=weight_decay)
    else:
        params = dict(W["projector"])
        params.update({"language_model." + k: v for k, v in W["lm"].items()})
        dlr = lr if decoder_lr is None else decoder_lr
        adamw_step(params, grads, state, lambda k: dlr if k.startswith("language_model.") else lr, weight_decay=weight_decay)
    return float(out["loss"]), gnorm
