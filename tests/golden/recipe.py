"""Seeded input recipes shared by make_golden.py (reference side) and the tests
(oracle / HIP side).  Pure numpy; imports nothing from the reference."""
from __future__ import annotations

import numpy as np

from oracle import weights as OW

# ----------------------------------------------------------------------------- shared small config
SMALL = dict(
    enc=OW.enc_config(hidden=256, ffn=512, layers=2, heads=4),
    lm=OW.lm_config(vocab=1024, hidden=256, ffn=512, layers=2, heads=4, kv_heads=2, head_dim=128),
    k=4, proj_hidden=128, audio_token_id=1023, pad_id=1000, eos_id=1001,
)


def logmel_waves():
    """Three clips: white noise 2.5 s, a shorter noisy two-tone 1.7003 s (ragged, not a hop
    multiple after padding is decided by the longest), a chirp 2.5 s."""
    sr = 16000
    w0 = OW.synthetic_wave(0, 40000)
    n1 = 27205
    tt = np.arange(n1) / sr
    w1 = (0.3 * np.sin(2 * np.pi * 440 * tt) + 0.05 * np.sin(2 * np.pi * 3000 * tt)
          + 0.01 * np.random.RandomState(7).standard_normal(n1)).astype(np.float32)
    tt = np.arange(40000) / sr
    w2 = (0.2 * np.sin(2 * np.pi * (200 + 1500 * tt) * tt)).astype(np.float32)
    return [w0, w1, w2]


def encoder_input(B=2, T=200, seed=11):
    return (0.6 * np.random.RandomState(seed).standard_normal((B, 128, T))).astype(np.float32)


def proj_input(B=2, S=50, seed=21):
    E = SMALL["enc"]["hidden"]
    x = np.random.RandomState(seed).standard_normal((B, S, E)).astype(np.float32)
    N = (S - 4) // 4 + 1
    dy = np.random.RandomState(seed + 1).standard_normal((B, N, SMALL["lm"]["hidden"])).astype(np.float32)
    return x, dy


def lm_input(B=2, L=48, seed=31):
    cfg = SMALL["lm"]
    rng = np.random.RandomState(seed)
    x = (rng.standard_normal((B, L, cfg["hidden"])) / np.sqrt(cfg["hidden"])).astype(np.float32)
    att = np.ones((B, L), dtype=np.int64)
    att[1, 40:] = 0
    lab = np.full((B, L), -100, dtype=np.int64)
    lab[0, 30:48] = rng.randint(0, 1000, 18)
    lab[1, 28:40] = rng.randint(0, 1000, 12)
    return x, att, lab




def asr_tokens(counts):
    """Token stream for the whole-model fixtures (ragged audio-token counts)."""
    return OW.synthetic_tokens(2, list(counts), SMALL["lm"]["vocab"], SMALL["audio_token_id"],
                               SMALL["pad_id"], SMALL["eos_id"], n_text=20, n_suffix=8, ragged=True)


def gen_waves():
    """Two clips of equal length (2.0 s) for the generation fixture: equal audio-token counts, so the prompt is the
    unpadded [prefix | <audio>*n | suffix] that ASRModel.generate itself builds (asr_modeling.py:588-616)."""
    return [OW.synthetic_wave(7, 32000), OW.synthetic_wave(8, 32000)]


def gen_prompt(n_audio, B=2, n_prefix=3, n_suffix=8):
    ids = np.concatenate([np.arange(5, 5 + n_prefix), np.full(n_audio, SMALL["audio_token_id"]),
                          np.arange(40, 40 + n_suffix)]).astype(np.int64)
    return np.tile(ids[None, :], (B, 1))


GEN_BOOST = 16.0


def gen_lm_weights():
    """LM weights of the generation fixture: o_proj / down_proj scaled up so that the block outputs dominate the
    residual stream -- with the plain init a random tied-embedding LM just repeats its last input token."""
    w = OW.init_lm(SMALL["lm"], seed=1)
    for k in w:
        if k.endswith("o_proj.weight") or k.endswith("down_proj.weight"):
            w[k] = (w[k] * np.float32(GEN_BOOST)).astype(np.float32)
    return w


# LoRA fixtures (file, rank, alpha, lora_target_modules; None = all 7 linears): the reference default and the other values of
# tiny_audio/asr_config.py:72-75's knobs -- a partial q|k|v group, a rank that needs three 16-column blocks, a partial gate|up group
LORA_CASES = (("lora_small.npz", 8, 32, None),
              ("lora_r4_qv_small.npz", 4, 32, ("q_proj", "v_proj")),
              ("lora_r16_small.npz", 16, 32, None),
              ("lora_r8_kou_small.npz", 8, 16, ("k_proj", "o_proj", "up_proj")))

QF = dict(heads=4, layers=2, window=15, downsample=5, eps=1e-12, ffn=512)       # reduced QFormer (hidden = encoder dim 256)


def qformer_input(B=2, S=50, seed=31):
    """S = 50 is not a multiple of the window (15): the last window is zero padded, as for the real S = 500."""
    E, D = SMALL["enc"]["hidden"], SMALL["lm"]["hidden"]
    x = np.random.RandomState(seed).standard_normal((B, S, E)).astype(np.float32)
    n_out = (S + 14) // 15 * 3
    dy = np.random.RandomState(seed + 1).standard_normal((B, n_out, D)).astype(np.float32)
    return x, dy


# ----------------------------------------------------------------------------- full decoder fine-tuning fixture
FULLFT_KEEP = {   # parameter -> slice of its gradient / final value that is stored (the fixture stays small)
    "model.layers.0.self_attn.q_proj.weight": (slice(0, 96), slice(None)),
    "model.layers.0.self_attn.k_proj.weight": (slice(128, 160), slice(None)),
    "model.layers.1.self_attn.v_proj.weight": (slice(0, 48), slice(None)),
    "model.layers.0.self_attn.o_proj.weight": (slice(0, 64), slice(None)),
    "model.layers.1.mlp.gate_proj.weight": (slice(0, 64), slice(None)),
    "model.layers.0.mlp.up_proj.weight": (slice(400, 464), slice(None)),
    "model.layers.1.mlp.down_proj.weight": (slice(0, 48), slice(None)),
    "model.embed_tokens.weight": (slice(None, None, 8), slice(None)),
}


def fullft_select(name, arr):
    if name in FULLFT_KEEP:
        return arr[FULLFT_KEEP[name]]
    return arr if arr.ndim == 1 else None           # every norm scale is kept whole


# ----------------------------------------------------------------------------- the benchmarked shape (round 5: recipe numerics)
# BASELINE configs[1]: GLM-ASR encoder defaults, Qwen3-0.6B, V = 151 670, MLP projector H = D = 1024 (SURVEY.md section 8 preamble)
FULL = dict(enc=OW.enc_config(), lm=OW.lm_config(), k=4, proj_hidden=1024,
            audio_token_id=151669, pad_id=151643, eos_id=151645)
FULL_L = 192
FULL_LOGIT_COL_STRIDE = 29          # stored logits columns: 0, 29, 58, ... (5 230 of 151 670)
FULL_GRAD_STRIDE = {"linear_1.weight": 16, "linear_2.weight": 4, "norm.weight": 1, "norm_2.weight": 1}


def full_clip_tokens():
    """The bench's token stream for ONE clip: 125 audio tokens, L = 192, 36 label positions."""
    return OW.synthetic_tokens(1, 125, FULL["lm"]["vocab"], FULL["audio_token_id"], FULL["pad_id"], FULL["eos_id"], L=FULL_L)


def full_logit_rows(att, lab):
    """Rows of the [L, V] logits kept in the fixture: every row that predicts a label, plus every 4th attended row."""
    att, lab = np.asarray(att).ravel(), np.asarray(lab).ravel()
    pred = np.nonzero(np.concatenate([lab[1:], [-100]]) != -100)[0]
    rows = sorted(set(pred.tolist()) | set(np.nonzero(att)[0][::4].tolist()))
    return np.asarray(rows, np.int64)


def full_grad_sample(name, g):
    return np.ascontiguousarray(np.asarray(g).ravel()[::FULL_GRAD_STRIDE[name]])


def lm_input_leftpad(B=2, L=48, pad=9, seed=41):
    """A LEFT-padded ragged batch with explicit position_ids (what trl's DataCollatorForChatML hands the model):
    clip 1 carries ``pad`` masked rows in front; its positions count the attended tokens from 0 (HF's convention:
    attention_mask.cumsum(-1) - 1, masked rows set to 1).  RoPE is translation invariant, so that alone would give the same
    answer as arange(L); clip 0 therefore gets NON-uniform positions (a gap of 37 after row 19, as when a cached prefix is
    skipped), which a forward that ignores position_ids cannot reproduce."""
    cfg = SMALL["lm"]
    rng = np.random.RandomState(seed)
    x = (rng.standard_normal((B, L, cfg["hidden"])) / np.sqrt(cfg["hidden"])).astype(np.float32)
    att = np.ones((B, L), dtype=np.int64)
    att[1, :pad] = 0
    pos = np.cumsum(att, -1) - 1
    pos[att == 0] = 1
    pos[0, 20:] += 37
    lab = np.full((B, L), -100, dtype=np.int64)
    lab[0, 30:48] = rng.randint(0, 1000, 18)
    lab[1, 33:48] = rng.randint(0, 1000, 15)
    return x, att, lab, pos.astype(np.int64)
