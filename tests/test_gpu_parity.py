"""-m gpu: the HIP hot path (composites through the C ABI + the Python drop-in modules) against
  (1) the committed golden vectors produced by the reference itself (tests/golden/*.npz, reduced widths), and
  (2) the numpy oracle on the same seeded inputs at the TRUE layer widths (reduced depth so the oracle takes seconds),
plus size-independent properties at BASELINE.json's full shapes.

Stated tolerances (bf16 MFMA operands, fp32 accumulate / residual / norms / softmax / CE):
  encoder / projector outputs: rel-to-max error <= 2e-2      logits: max-abs 0.1, RMS 0.02 (bf16 output)
  loss: relative 5e-3                                          projector gradients: cosine >= 0.999
  (logits: see logits_close)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import encoder as OE
from oracle import model as OM
from oracle import projectors as OP
from oracle import qwen3 as OQ
from oracle import weights as OW
from tests.golden import recipe as R

if torch.cuda.is_available():
    from tiny_audio_amd.asr_config import ASRConfig, EncoderConfig, LMConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    from tiny_audio_amd.encoder import GlmAsrEncoderMI355X
    from tiny_audio_amd.language_model import Qwen3MI355X
    from tiny_audio_amd.projectors import MLPAudioProjector
    from tiny_audio_amd.trainer import ASRTrainer, TrainingArguments
    from tiny_audio_amd import ops

DEV = "cuda"


def relmax(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def cosine(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))


def npy(t):
    return t.detach().float().cpu().numpy()


def logits_close(got, ref):
    """outputs.logits is returned in bf16 (as the reference's autocast lm_head does): with |logit| up to ~6 the
    bf16 quantum alone is 0.03, so the stated tolerance is max-abs <= 0.1 AND RMS error <= 0.02."""
    d = np.asarray(got, np.float64) - np.asarray(ref, np.float64)
    return float(np.abs(d).max()) < 0.1 and float(np.sqrt((d ** 2).mean())) < 0.02


def build_model(enc_cfg, lm_cfg, proj_hidden, wE, wL, wP, **kw):
    cfg = ASRConfig(audio_config=enc_cfg, text_config=lm_cfg, projector_hidden_dim=proj_hidden,
                    audio_token_id=kw.pop("audio_token_id"), **kw)
    m = ASRModel(cfg, device=DEV, init="none")
    m.audio_tower.load_state_dict_hf(wE)
    m.language_model.load_state_dict_hf(wL)
    m.load_state_dict({"projector." + k: torch.from_numpy(v) for k, v in wP.items()})
    return m


# ============================================================================ (1) golden vectors from the reference
def test_encoder_vs_golden(golden):
    g = golden("encoder_small.npz")
    cfg = R.SMALL["enc"]
    enc = GlmAsrEncoderMI355X(EncoderConfig(cfg), DEV).load_state_dict_hf(OW.init_encoder(cfg, 0))
    out = enc(torch.from_numpy(R.encoder_input()), return_f32=True).last_hidden_state
    assert relmax(npy(out), g["last_hidden_state"]) < 2e-2


def test_mlp_projector_vs_golden(golden):
    g = golden("projector_mlp.npz")
    S = R.SMALL
    E, D, H = S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"]
    cfg = ASRConfig(audio_config=S["enc"], text_config=S["lm"], projector_hidden_dim=H)
    p = MLPAudioProjector(cfg).to(DEV)
    p.load_state_dict({k: torch.from_numpy(v) for k, v in OW.init_mlp_projector(E, D, H).items()})
    x, dy = R.proj_input()
    y = p(torch.from_numpy(x).to(DEV).to(torch.bfloat16))
    assert relmax(npy(y), g["y"]) < 2e-2
    y.backward(torch.from_numpy(dy).to(DEV))
    for k, prm in p.named_parameters():
        assert cosine(npy(prm.grad), g["g." + k]) > 0.999, k
        assert relmax(npy(prm.grad), g["g." + k]) < 4e-2, k


def test_asr_model_vs_golden(golden):
    g = golden("asr_small.npz")
    S = R.SMALL
    E, D, H = S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"]
    m = build_model(S["enc"], S["lm"], H, OW.init_encoder(S["enc"], 0), OW.init_lm(S["lm"], 1),
                    OW.init_mlp_projector(E, D, H), audio_token_id=S["audio_token_id"])
    ids, att, lab, counts = R.asr_tokens(g["counts"])
    m.train()
    out = m(input_ids=torch.from_numpy(ids), input_features=torch.from_numpy(g["input_features"]),
            attention_mask=torch.from_numpy(att), labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts))
    out.loss.backward()
    ref_loss = float(g["mlp.loss"])
    assert abs(float(out.loss) - ref_loss) < 5e-3 * ref_loss
    valid = att.astype(bool)
    assert logits_close(npy(out.logits)[valid], g["mlp.logits"][valid])
    for k, prm in m.projector.named_parameters():
        assert cosine(npy(prm.grad), g["mlp.g." + k]) > 0.999, k


def test_three_training_steps_vs_golden(golden):
    """Row a13: AdamW + global-norm clip through the HIP optimizer kernels on the reference's 3-step fixture."""
    g, g3 = golden("asr_small.npz"), golden("train3_small.npz")
    S = R.SMALL
    E, D, H = S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"]
    m = build_model(S["enc"], S["lm"], H, OW.init_encoder(S["enc"], 0), OW.init_lm(S["lm"], 1),
                    OW.init_mlp_projector(E, D, H), audio_token_id=S["audio_token_id"])
    ids, att, lab, counts = R.asr_tokens(g["counts"])
    batch = dict(input_ids=torch.from_numpy(ids), input_features=torch.from_numpy(g["input_features"]),
                 attention_mask=torch.from_numpy(att), labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts))
    m.train()
    tr = ASRTrainer(m, TrainingArguments(learning_rate=1e-3, weight_decay=0.0, max_grad_norm=1.0))
    losses, gnorms = [], []
    for _ in range(3):
        tr.training_step(batch)
        losses.append(tr.last_loss()); gnorms.append(tr.last_grad_norm())
    np.testing.assert_allclose(losses, g3["losses"], rtol=5e-3)
    np.testing.assert_allclose(gnorms, g3["gnorms"], rtol=3e-2)
    assert losses[2] < losses[0]
    for k, prm in m.projector.named_parameters():
        d = np.abs(npy(prm) - g3["w." + k])
        assert d.mean() < 2e-4, (k, d.mean())            # each Adam step moves a weight by <= lr = 1e-3


# ============================================================================ (2) oracle at true widths, reduced depth
TRUE_ENC = OW.enc_config(layers=2)
TRUE_LM = OW.lm_config(vocab=5003, layers=2)        # true widths; small odd vocab exercises the padded-column masking
AID, PAD, EOS = 5002, 4990, 4991


@pytest.mark.parametrize("B,T", [(2, 301), (2, 207), (8, 77)], ids=["S151-unfused", "S104-fused", "S39-fused-unaligned"])
def test_encoder_true_width_vs_oracle(B, T, monkeypatch):
    """(2, 301): M = 302 is not a multiple of 8 -> the qkv_post path; the other two run the fused q|k|v path (rope in the
    GEMM epilogue, V^T from a GEMM, strided attention), with 16-B-aligned and element-aligned clip offsets in V^T."""
    w = OW.init_encoder(TRUE_ENC, 0)
    enc = GlmAsrEncoderMI355X(EncoderConfig(TRUE_ENC), DEV).load_state_dict_hf(w)
    x = (0.6 * np.random.RandomState(3).standard_normal((B, 128, T))).astype(np.float32)
    S = (T - 1) // 2 + 1
    ref = OE.encoder_forward(x, w, TRUE_ENC)
    out = enc(torch.from_numpy(x), return_f32=True).last_hidden_state
    assert out.shape == (B, S, 1280)
    assert relmax(npy(out), ref) < 2e-2 and cosine(npy(out), ref) > 0.9995
    keep = (np.random.RandomState(4).rand(B, S) < 0.8).astype(np.float32)
    outk = enc(torch.from_numpy(x), frame_keep=torch.from_numpy(keep).reshape(-1), return_f32=True).last_hidden_state
    assert relmax(npy(outk), ref * keep[:, :, None]) < 2e-2
    assert float(npy(outk)[keep == 0].__abs__().max()) == 0.0          # dropped frames are exactly zero, no rescale
    if (B * S) % 8 == 0:                                               # both paths compute the same function
        monkeypatch.setenv("TA355_ENC_QKV_FUSED", "0")
        out0 = enc(torch.from_numpy(x), return_f32=True).last_hidden_state
        assert relmax(npy(out0), ref) < 2e-2 and relmax(npy(out), npy(out0)) < 1.5e-2
        assert not np.array_equal(npy(out), npy(out0))                 # ... through different kernels


@pytest.mark.parametrize("hidden", [1024, 2048])
def test_mlp_projector_true_width_vs_oracle(hidden):
    E, D = 1280, 1024
    w = OW.init_mlp_projector(E, D, hidden)
    cfg = ASRConfig(audio_config=TRUE_ENC, text_config=TRUE_LM, projector_hidden_dim=hidden)
    p = MLPAudioProjector(cfg).to(DEV)
    p.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    rng = np.random.RandomState(5)
    x = rng.standard_normal((3, 102, E)).astype(np.float32)          # 102 frames: tail of 2 is dropped (N = 25)
    xb = torch.from_numpy(x).to(DEV).to(torch.bfloat16)
    y = p(xb)
    xr = npy(xb)
    ref, c = OP.mlp_forward(xr, w)
    assert y.shape == (3, 25, D) and relmax(npy(y), ref) < 2e-2
    dy = rng.standard_normal(ref.shape).astype(np.float32)
    y.backward(torch.from_numpy(dy).to(DEV))
    gr = OP.mlp_backward(dy, w, c)
    for k, prm in p.named_parameters():
        assert cosine(npy(prm.grad), gr[k]) > 0.999, k


def _true_batch(B=2, T=200):
    feats = (0.6 * np.random.RandomState(6).standard_normal((B, 128, T))).astype(np.float32)
    n_audio = ((T - 1) // 2 + 1 - 4) // 4 + 1
    counts = [n_audio, n_audio - 5][:B]
    ids, att, lab, counts = OW.synthetic_tokens(B, counts, TRUE_LM["vocab"], AID, PAD, EOS, n_text=14, n_suffix=6, ragged=True)
    return dict(input_ids=ids, attention_mask=att, labels=lab, input_features=feats, audio_token_counts=counts)


LM_17B = OW.lm_config(vocab=5003, layers=2, hidden=2048, ffn=6144)     # Qwen3-1.7B widths (transcription.yaml:14-16)


@pytest.mark.parametrize("lm_cfg", [TRUE_LM, LM_17B], ids=["qwen3-0.6b", "qwen3-1.7b"])
def test_full_model_true_width_vs_oracle(lm_cfg):
    TRUE_LM = lm_cfg
    D = lm_cfg["hidden"]
    wE, wL = OW.init_encoder(TRUE_ENC, 0), OW.init_lm(TRUE_LM, 1)
    wP = OW.init_mlp_projector(1280, D, 1024)
    m = build_model(TRUE_ENC, TRUE_LM, 1024, wE, wL, wP, audio_token_id=AID)
    b = _true_batch()
    m.train()
    out = m(**{k: torch.from_numpy(v) for k, v in b.items()})
    out.loss.backward()
    W = dict(encoder=wE, lm=wL, projector={k: v.copy() for k, v in wP.items()})
    cfg = dict(enc=TRUE_ENC, lm=TRUE_LM, projector_type="mlp", k=4, audio_token_id=AID)
    ref = OM.asr_forward(b, W, cfg, training=True)
    grads, _ = OM.asr_backward(ref, W, cfg)
    assert abs(float(out.loss) - float(ref["loss"])) < 5e-3 * float(ref["loss"])
    assert out.n_label_tokens == ref["n_label_tokens"]
    valid = b["attention_mask"].astype(bool)
    assert logits_close(npy(out.logits)[valid], ref["logits"][valid])
    for k, prm in m.projector.named_parameters():
        assert cosine(npy(prm.grad), grads[k]) > 0.999, k
    # sum-CE / num_items semantics of the HF Trainer (TF:loss/loss_utils.py:33-46)
    out77 = m(**{k: torch.from_numpy(v) for k, v in b.items()}, num_items_in_batch=77, return_logits=False)
    assert abs(float(out77.loss) * 77 - float(out.loss) * out.n_label_tokens) < 1e-3 * float(out.loss) * out.n_label_tokens
    assert out77.logits is None


def test_lm_text_only_and_dx_vs_oracle():
    """Row a9/a10 in isolation: inputs_embeds-level gradient of the frozen LM (no audio)."""
    wL = OW.init_lm(TRUE_LM, 1)
    lm = Qwen3MI355X(LMConfig(TRUE_LM), DEV).load_state_dict_hf(wL)
    rng = np.random.RandomState(8)
    B, L = 2, 70
    ids = rng.randint(0, 4900, (B, L)).astype(np.int64)
    att = np.ones((B, L), np.int64); att[1, 55:] = 0
    lab = np.full((B, L), -100, np.int64); lab[0, 40:70] = ids[0, 40:70]; lab[1, 30:55] = ids[1, 30:55]
    rows, tg, n = ops.label_rows(torch.from_numpy(lab).to(DEV))
    n = int(n.item())
    loss, nll, logits, ctx = lm.forward_loss(torch.from_numpy(ids).to(DEV), None, None, torch.from_numpy(att).to(DEV).int(),
                                             rows, tg, n, 1.0 / n, want_logits=True)
    x0 = wL["model.embed_tokens.weight"][ids]
    ref_logits, cache = OQ.lm_forward(x0, att, wL, TRUE_LM)
    ref_loss, dlogits, n_ref = OQ.causal_lm_loss(ref_logits, lab)
    assert n == n_ref and abs(float(loss) - float(ref_loss)) < 5e-3 * float(ref_loss)
    valid = att.astype(bool)
    got = npy(logits).reshape(B, L, -1)[:, :, :TRUE_LM["vocab"]]
    assert logits_close(got[valid], ref_logits[valid])
    _, d_emb, _ = lm.backward_from_ctx(ctx, 1, want_d_embeds=True)
    ref_dx = OQ.lm_backward_dx(dlogits, wL, TRUE_LM, cache)
    got_dx = npy(d_emb).reshape(B, L, -1)
    assert cosine(got_dx[valid], ref_dx[valid]) > 0.999
    assert relmax(got_dx[valid], ref_dx[valid]) < 5e-2


def test_lm_long_sequence_and_single_clip_vs_oracle():
    """Edge shapes: L = 260 (> 192: the tiled attention kernels instead of the whole-sequence one; 5 key tiles, ragged
    last tile) with left AND right padding in one batch, and B = 1 with L = 33 (single partial tile)."""
    wL = OW.init_lm(TRUE_LM, 1)
    lm = Qwen3MI355X(LMConfig(TRUE_LM), DEV).load_state_dict_hf(wL)
    rng = np.random.RandomState(12)
    for B, L in ((2, 260), (1, 33)):
        ids = rng.randint(0, 4900, (B, L)).astype(np.int64)
        att = np.ones((B, L), np.int64)
        lab = np.full((B, L), -100, np.int64)
        if B == 2:
            att[0, 230:] = 0; att[1, :17] = 0                        # right-padded clip 0, left-padded clip 1
            lab[0, 150:230] = ids[0, 150:230]; lab[1, 200:260] = ids[1, 200:260]
        else:
            lab[0, 20:33] = ids[0, 20:33]
        pos = np.clip(np.cumsum(att, -1) - 1, 0, None).astype(np.int32)
        rows, tg, n = ops.label_rows(torch.from_numpy(lab).to(DEV))
        n = int(n.item())
        loss, nll, logits, ctx = lm.forward_loss(torch.from_numpy(ids).to(DEV), None, None, torch.from_numpy(att).to(DEV).int(),
                                                 rows, tg, n, 1.0 / n, want_logits=True, pos=torch.from_numpy(pos).to(DEV))
        x0 = wL["model.embed_tokens.weight"][ids]
        ref_logits, cache = OQ.lm_forward(x0, att, wL, TRUE_LM, position_ids=None if B == 1 else pos)
        ref_loss, dlogits, n_ref = OQ.causal_lm_loss(ref_logits, lab)
        assert n == n_ref and abs(float(loss) - float(ref_loss)) < 5e-3 * float(ref_loss)
        valid = att.astype(bool)
        got = npy(logits).reshape(B, L, -1)[:, :, :TRUE_LM["vocab"]]
        assert logits_close(got[valid], ref_logits[valid])
        _, d_emb, _ = lm.backward_from_ctx(ctx, 1, want_d_embeds=True)
        ref_dx = OQ.lm_backward_dx(dlogits, wL, TRUE_LM, cache)
        assert cosine(npy(d_emb).reshape(B, L, -1)[valid], ref_dx[valid]) > 0.999


# ============================================================================ LoRA stage 2 (row a11, BASELINE configs[4])
def _lora_case(cfg, wL, lo, x_ids, att, lab):
    """HIP LM with adapters vs oracle: loss, logits, d(inputs_embeds), every adapter gradient."""
    lm = Qwen3MI355X(LMConfig(cfg), DEV).load_state_dict_hf(wL)
    lm.enable_lora(rank=8, alpha=32).load_lora_state_dict(lo)
    B, L = x_ids.shape
    rows, tg, n = ops.label_rows(torch.from_numpy(lab).to(DEV))
    n = int(n.item())
    loss, nll, logits, ctx = lm.forward_loss(torch.from_numpy(x_ids).to(DEV), None, None, torch.from_numpy(att).to(DEV).int(),
                                             rows, tg, n, 1.0 / n, want_logits=True)
    x0 = wL["model.embed_tokens.weight"][x_ids]
    ref_logits, cache = OQ.lm_forward(x0, att, wL, cfg, lora=lo, lora_scale=4.0)
    ref_loss, dlogits, n_ref = OQ.causal_lm_loss(ref_logits, lab)
    base_logits, _ = OQ.lm_forward(x0, att, wL, cfg)
    valid = att.astype(bool)
    got = npy(logits).reshape(B, L, -1)[:, :, :cfg["vocab"]]
    assert n == n_ref and abs(float(loss) - float(ref_loss)) < 5e-3 * float(ref_loss)
    assert logits_close(got[valid], ref_logits[valid])
    # the adapters must actually matter in this case, otherwise the comparison above proves nothing
    assert np.abs(ref_logits[valid] - base_logits[valid]).max() > 0.3
    _, d_emb, lg = lm.backward_from_ctx(ctx, 1, want_d_embeds=True, want_d_audio=False)
    ref_g = {}
    ref_dx = OQ.lm_backward_dx(dlogits, wL, cfg, cache, lo, 4.0, ref_g)
    got_dx = npy(d_emb).reshape(B, L, -1)
    assert cosine(got_dx[valid], ref_dx[valid]) > 0.999
    # export the packed gradient tensors through the same name mapping as the parameters
    for p_, g_ in zip(lm.lora_parameters(), lg):
        p_.data.copy_(g_)
    got_g = lm.export_lora_state_dict(prefix="model.", suffix="")
    assert set(got_g) == set(ref_g)
    for k in ref_g:
        assert cosine(npy(got_g[k]), ref_g[k]) > 0.998, k
        assert relmax(npy(got_g[k]), ref_g[k]) < 6e-2, k


@pytest.mark.parametrize("case", R.LORA_CASES, ids=lambda c: c[0][:-4])
def test_lora_vs_golden_config(golden, case):
    """Reduced configs of tests/golden/lora*_small.npz (values produced by the reference's Qwen3 + merged adapters): the reference
    default (r = 8, all 7 linears) and other values of tiny_audio/asr_config.py:72-75's lora_rank / lora_alpha /
    lora_target_modules -- members outside the target list stay exactly zero and are not exported."""
    fname, rank, alpha, targets = case
    g = golden(fname)
    cfg = R.SMALL["lm"]
    wL, lo = OW.init_lm(cfg, seed=1), OW.init_lora(cfg, rank=rank, seed=4, targets=targets)
    lm = Qwen3MI355X(LMConfig(cfg), DEV).load_state_dict_hf(wL)
    lm.enable_lora(rank=rank, alpha=alpha, target_modules=None if targets is None else list(targets)).load_lora_state_dict(lo)
    assert lm.lora_param_count() == sum(v.size for v in lo.values())          # peft's trainable-parameter count
    x, att, lab = R.lm_input()                                    # inputs_embeds-level fixture: feed as "audio" rows
    B, L, D = x.shape
    ids = torch.full((B, L), R.SMALL["audio_token_id"], dtype=torch.int64, device=DEV)
    src = torch.arange(B * L, dtype=torch.int32, device=DEV)
    rows, tg, n = ops.label_rows(torch.from_numpy(lab).to(DEV))
    n = int(n.item())
    audio = torch.from_numpy(x.reshape(B * L, D)).to(DEV)
    loss, nll, logits, ctx = lm.forward_loss(ids, src, audio, torch.from_numpy(att).to(DEV).int(), rows, tg, n, 1.0 / n,
                                             want_logits=True)
    assert abs(float(loss) - float(g["loss"])) < 5e-3 * float(g["loss"])
    d_audio, _, lg = lm.backward_from_ctx(ctx, B * L)
    valid = att.astype(bool)
    assert cosine(npy(d_audio).reshape(B, L, D)[valid], g["dx"][valid]) > 0.999
    masters = [p_.detach().clone() for p_ in lm.lora_parameters()]
    for p_, g_ in zip(lm.lora_parameters(), lg):
        p_.data.copy_(g_)
    got = lm.export_lora_state_dict(prefix="model.", suffix="")
    assert set(got) == set(lo)
    for k in [k[2:] for k in g.files if k.startswith("g.")]:
        assert cosine(npy(got[k]), g["g." + k]) > 0.998, k
    # what is not targeted has zero masters AND zero gradients, element for element
    n_live = sum(int((g_ != 0).sum()) for g_ in lg)
    assert n_live <= lm.lora_param_count() and n_live > 0.98 * lm.lora_param_count()
    for m_, g_ in zip(masters, lg):
        dead_rows = (m_ == 0).all(dim=-1)
        if targets is not None:
            assert bool((g_[dead_rows] == 0).all())


def test_lora_true_width_vs_oracle():
    wL, lo = OW.init_lm(TRUE_LM, 1), OW.init_lora(TRUE_LM, rank=8, seed=4)
    rng = np.random.RandomState(8)
    B, L = 2, 70
    ids = rng.randint(0, 4900, (B, L)).astype(np.int64)
    att = np.ones((B, L), np.int64); att[1, 55:] = 0
    lab = np.full((B, L), -100, np.int64); lab[0, 40:70] = ids[0, 40:70]; lab[1, 30:55] = ids[1, 30:55]
    _lora_case(TRUE_LM, wL, lo, ids, att, lab)


def test_lora_zero_b_is_identity_and_stage2_trains():
    """peft init (lora_B = 0): the adapted model equals the base model; stage 2 (frozen projector) updates only the
    adapters, lora_B first (dA = 0 while B = 0), and the loss decreases over a few AdamW steps."""
    S = R.SMALL
    wE, wL = OW.init_encoder(S["enc"], 0), OW.init_lm(S["lm"], 1)
    wP = OW.init_mlp_projector(S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"])
    kw = dict(audio_token_id=S["audio_token_id"], audio_token_dropout=0.0)
    base = build_model(S["enc"], S["lm"], S["proj_hidden"], wE, wL, wP, **kw)
    m = build_model(S["enc"], S["lm"], S["proj_hidden"], wE, wL, wP, use_lora=True, freeze_projector=True, **kw)
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    assert len(names) == 8 and all(n.startswith("language_model.lora_") for n in names)
    n_true = sum(p.numel() for _, p in m.named_parameters() if p.requires_grad)
    c = S["lm"]
    D, F, nq, nkv, hd = c["hidden"], c["ffn"], c["heads"], c["kv_heads"], c["head_dim"]
    per_layer = 8 * ((D + nq * hd) + 2 * (D + nkv * hd) + (nq * hd + D) + 2 * (D + F) + (F + D))
    assert n_true == c["layers"] * per_layer                       # exact peft parameter count, no padding
    feats = torch.from_numpy(R.encoder_input())
    B = feats.shape[0]
    counts = np.full(B, m.projector.get_output_length(m.audio_tower.output_length(feats.shape[2])), np.int64)
    ids, att, lab, counts = R.asr_tokens(counts)
    tb = dict(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(att), labels=torch.from_numpy(lab),
              audio_token_counts=torch.from_numpy(counts))
    o0 = base(input_features=feats, **tb)
    o1 = m(input_features=feats, **tb)
    assert abs(float(o0.loss.detach()) - float(o1.loss.detach())) < 1e-5      # CE sum is an atomic reduction: order-dependent ulps
    assert torch.equal(o0.logits, o1.logits)
    tr = ASRTrainer(m, TrainingArguments(learning_rate=2e-3, warmup_steps=0, max_steps=8, lr_scheduler_type="constant",
                                         weight_decay=0.0))
    a0 = m.language_model.lora_la_qkv.detach().clone()
    losses = [tr.training_step(dict(input_features=feats, **tb)) for _ in range(1)]
    assert torch.equal(m.language_model.lora_la_qkv.detach(), a0)  # dA = s B^T dW = 0 on the first step
    assert float(m.language_model.lora_lb_qkv.detach().abs().max()) > 0
    losses += [tr.training_step(dict(input_features=feats, **tb)) for _ in range(7)]
    assert losses[-1] < losses[0] - 0.05, losses
    assert all(p.grad is None or not p.requires_grad for p in m.projector.parameters())


# ============================================================================ QFormer projector (section 8(f) rank 4)
def _qformer_case(E, D, cfgq, x, dy, hidden_kw, keeps_np=None):
    from oracle import qformer as OQF
    from tiny_audio_amd.qformer_projector import QFormerAudioProjector
    w = OW.init_qformer_projector(E, D, layers=cfgq["layers"], ffn=cfgq["ffn"])
    cfg = ASRConfig(audio_config=dict(hidden=E, ffn=2 * E, layers=1, heads=E // 64), text_config=dict(hidden=D, ffn=2 * D, layers=1, heads=4, kv_heads=2, vocab=512),
                    projector_type="qformer", qformer_num_heads=cfgq["heads"], qformer_num_layers=cfgq["layers"],
                    qformer_intermediate_size=cfgq["ffn"], **hidden_kw)
    m = QFormerAudioProjector(cfg).to(DEV)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    m.eval()
    keeps_t = None if keeps_np is None else {k: torch.from_numpy(np.broadcast_to(v, v.shape).copy()).to(DEV) for k, v in keeps_np.items()}
    y = m(torch.from_numpy(x).to(DEV), keeps=keeps_t or {})
    ref, c = OQF.qformer_forward(x, w, cfgq, keeps=keeps_np)
    assert y.shape == ref.shape and relmax(npy(y), ref) < 2e-2 and cosine(npy(y), ref) > 0.9995
    (y * torch.from_numpy(dy).to(DEV)).sum().backward()
    grads = OQF.qformer_backward(dy, w, cfgq, c)
    for k, p_ in m.named_parameters():
        if k.endswith("key.bias"):                     # mathematically zero (softmax shift invariance): compare to the scale of its siblings
            assert np.abs(npy(p_.grad)).max() < 2e-2 * np.abs(grads[k.replace("key.bias", "value.bias")]).max(), k
        else:
            assert cosine(npy(p_.grad), grads[k]) > 0.995, k
            assert relmax(npy(p_.grad), grads[k]) < 8e-2, k
    return m


def test_qformer_projector_vs_golden_config(golden):
    g = golden("projector_qformer.npz")
    x, dy = R.qformer_input()
    E, D = R.SMALL["enc"]["hidden"], R.SMALL["lm"]["hidden"]
    m = _qformer_case(E, D, R.QF, x, dy, {})
    assert relmax(npy(m(torch.from_numpy(x).to(DEV), keeps={})), g["y"]) < 2e-2
    for k in [k[2:] for k in g.files if k.startswith("g.") and not k.endswith("key.bias")]:
        assert cosine(npy(dict(m.named_parameters())[k].grad), g["g." + k]) > 0.995, k


def test_qformer_projector_true_width_with_dropout_masks():
    """True widths (hidden 1280, 16 heads of 80, FFN 5120, S = 500 -> 34 windows -> 102 tokens per clip), one layer, with
    injected dropout keep masks at every site (the train-mode algebra); then the train-mode RNG path just has to run
    and give a different, finite output."""
    from oracle import qformer as OQF
    cfgq = dict(heads=16, layers=1, window=15, downsample=5, eps=1e-12, ffn=5120)
    rng = np.random.RandomState(2)
    B, S = 1, 500
    x = rng.standard_normal((B, S, 1280)).astype(np.float32)
    dy = rng.standard_normal((B, 102, 1024)).astype(np.float32)
    EB, M = 34, 102
    mk = lambda *s: ((rng.rand(*s) < 0.9) / 0.9).astype(np.float32)
    keeps = {"emb": mk(M, 1280).reshape(EB, 3, 1280), "l0.sa": mk(M, 1280).reshape(EB, 3, 1280), "l0.ca": mk(M, 1280).reshape(EB, 3, 1280),
             "l0.ffn": mk(M, 1280).reshape(EB, 3, 1280), "l0.sa_p": mk(EB, 16, 3, 3), "l0.ca_p": mk(EB, 16, 3, 15)}
    m = _qformer_case(1280, 1024, cfgq, x, dy, {}, keeps_np=keeps)
    assert m.get_output_length(500) == 102 == OQF.output_length(500)
    m.train()
    y1, y2 = m(torch.from_numpy(x).to(DEV)), m(torch.from_numpy(x).to(DEV))
    assert torch.isfinite(y1).all() and not torch.equal(y1, y2)
    m.eval()
    assert torch.equal(m(torch.from_numpy(x).to(DEV)), m(torch.from_numpy(x).to(DEV)))


# ============================================================================ MOSA projector (section 8(f) rank 4)
def _mosa_case(E, D, x, dy):
    from oracle import mosa as OMS
    from tiny_audio_amd.mosa_projector import MOSAProjector
    w = OW.init_mosa_projector(E, D)
    cfg = ASRConfig(audio_config=dict(hidden=E, ffn=2 * E, layers=1, heads=E // 64), text_config=dict(hidden=D, ffn=2 * D, layers=1, heads=4, kv_heads=2, vocab=512),
                    projector_type="mosa")
    m = MOSAProjector(cfg).to(DEV)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    y = m(torch.from_numpy(x).to(DEV))
    ref, c = OMS.mosa_forward(x, w)
    assert y.shape == ref.shape and relmax(npy(y), ref) < 2e-2 and cosine(npy(y), ref) > 0.9995
    (y * torch.from_numpy(dy).to(DEV)).sum().backward()
    grads = OMS.mosa_backward(dy, w, c)
    for k, p_ in m.named_parameters():
        assert cosine(npy(p_.grad), grads[k]) > 0.995, k
        # the router's ReLU has a kink at 0: pre-activations within bf16 rounding of 0 switch sides, which moves single
        # entries of the first router layer's gradient by a whole token's contribution (the direction is unaffected)
        assert relmax(npy(p_.grad), grads[k]) < (0.3 if k.startswith("router.0") else 8e-2), k
    return m


def test_mosa_projector_vs_golden_config(golden):
    g = golden("projector_mosa.npz")
    x, _ = R.proj_input()
    m = _mosa_case(R.SMALL["enc"]["hidden"], R.SMALL["lm"]["hidden"], x, g["dy"])
    P = dict(m.named_parameters())
    assert relmax(npy(m(torch.from_numpy(x).to(DEV))), g["y"]) < 2e-2
    for k in [k[2:] for k in g.files if k.startswith("g.")]:
        assert cosine(npy(P[k].grad), g["g." + k]) > 0.995, k
    assert cosine(npy(P["experts.2.fc1.weight"].grad)[:64], g["rows64_experts_2_fc1_weight"]) > 0.995


def test_mosa_projector_true_width():
    rng = np.random.RandomState(4)
    x = rng.standard_normal((2, 101, 1280)).astype(np.float32)               # odd length: T1 = 51, T2 = 26
    dy = rng.standard_normal((2, 26, 1024)).astype(np.float32)
    m = _mosa_case(1280, 1024, x, dy)
    assert m.get_output_length(500) == 125 and m.get_output_length(101) == 26


# ============================================================================ greedy generation (section 8(f) rank 1)
def _check_greedy_against_oracle(tokens, batch, W, cfg, eos_ids, pad_id, tol=0.12, processors_see_prompt=True, min_new_tokens=0,
                                 **processors):
    """Greedy parity that is robust to bf16 near-ties: feed the HIP path's OWN tokens to the fp32 oracle and require
    every decision to be the oracle's argmax or within `tol` logits of it (bf16 logits carry ~0.03 of rounding);
    pad-after-EOS and the stopping rule are checked exactly."""
    from oracle import generate as OG
    x = OG.prompt_embeds(batch, W, cfg)
    embed = W["lm"]["model.embed_tokens.weight"]
    B, n_new = tokens.shape
    unfinished = np.ones(B, bool)
    exact = 0
    for t in range(n_new):
        assert unfinished.any(), "generation continued after every clip had finished"
        logits, _ = OQ.lm_forward(x, np.ones(x.shape[:2], np.int64), W["lm"], cfg["lm"], keep_cache=False)
        last = logits[:, -1]
        if processors:
            seq = np.concatenate([np.asarray(batch["input_ids"], np.int64), tokens[:, :t]], axis=1) if processors_see_prompt else tokens[:, :t]
            last = OG.apply_logits_processors(last.astype(np.float32), seq, **processors)
        if t < min_new_tokens:                                # HF MinNewTokensLengthLogitsProcessor
            last = last.astype(np.float32).copy()
            last[:, list(eos_ids)] = -np.inf
            assert not np.isin(tokens[:, t], list(eos_ids)).any(), "an eos id before min_new_tokens"
        for b in range(B):
            if not unfinished[b]:
                assert tokens[b, t] == pad_id
            else:
                assert last[b].max() - last[b, tokens[b, t]] < tol, (b, t, int(tokens[b, t]), int(last[b].argmax()))
                exact += int(tokens[b, t] == last[b].argmax())
        unfinished &= ~np.isin(tokens[:, t], list(eos_ids))
        x = np.concatenate([x, embed[tokens[:, t]][:, None, :]], axis=1)
    return exact


def test_generate_vs_golden_and_oracle(golden):
    g = golden("generate_small.npz")
    S = R.SMALL
    E, D, H = S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"]
    wE, wL, wP = OW.init_encoder(S["enc"], 0), R.gen_lm_weights(), OW.init_mlp_projector(E, D, H)
    m = build_model(S["enc"], S["lm"], H, wE, wL, wP, audio_token_id=S["audio_token_id"], pad_token_id=S["pad_id"],
                    eos_token_id=S["eos_id"])
    kw = dict(input_ids=torch.from_numpy(g["input_ids"]), input_features=torch.from_numpy(g["input_features"]),
              audio_attention_mask=torch.from_numpy(g["audio_attention_mask"]),
              attention_mask=torch.ones(g["input_ids"].shape, dtype=torch.int64))
    W = dict(encoder=wE, lm=wL, projector=wP)
    cfg = dict(enc=S["enc"], lm=S["lm"], projector_type="mlp", k=S["k"], audio_token_id=S["audio_token_id"])
    batch = dict(input_ids=g["input_ids"], input_features=g["input_features"])
    a = m.generate(**kw, max_new_tokens=12).cpu().numpy()
    assert a.shape == (2, 12)
    exact = _check_greedy_against_oracle(a, batch, W, cfg, (S["eos_id"], S["pad_id"]), S["pad_id"])
    assert exact >= 20                                       # of 24 decisions; the rest are within-tolerance ties
    # the reference's own tokens: identical wherever the reference's decision margin exceeds the bf16 tolerance
    assert (a == g["tokens_a"]).mean() > 0.5 and (a[:, :3] == g["tokens_a"][:, :3]).all()
    eos_b = int(g["eos_b"])
    b = m.generate(**kw, max_new_tokens=12, eos_token_id=[eos_b, S["pad_id"]]).cpu().numpy()
    _check_greedy_against_oracle(b, batch, W, cfg, (eos_b, S["pad_id"]), S["pad_id"])
    for row in b:                                            # EOS semantics: nothing but pad after the first eos
        hit = np.nonzero(np.isin(row, [eos_b, S["pad_id"]]))[0]
        if hit.size:
            assert (row[hit[0] + 1:] == S["pad_id"]).all()
    assert b.shape[1] <= 12
    if b.shape[1] < 12:                                      # stopped early: every clip must have emitted an eos id
        assert np.isin(b, [eos_b, S["pad_id"]]).any(axis=1).all()
    # generation_config.min_new_tokens (tiny_audio/asr_config.py:83; round 4): no eos id before min_new tokens exist -- the fixture's
    # clip 0 would stop at step 6, the reference's run with min_new_tokens = 9 carries on
    mn = int(g["min_new_c"])
    c = m.generate(**kw, max_new_tokens=12, eos_token_id=[eos_b, S["pad_id"]], min_new_tokens=mn).cpu().numpy()
    assert c.shape == g["tokens_c"].shape and not np.isin(c[:, :mn], [eos_b, S["pad_id"]]).any()
    _check_greedy_against_oracle(c, batch, W, cfg, (eos_b, S["pad_id"]), S["pad_id"], min_new_tokens=mn)
    assert (c == g["tokens_c"]).mean() > 0.5 and (c[:, :3] == g["tokens_c"][:, :3]).all()
    with pytest.raises(NotImplementedError):
        m.generate(**kw, num_beams=4)
    with pytest.raises(ValueError):
        m.generate(input_ids=kw["input_ids"], input_features=kw["input_features"])


def test_generate_with_repetition_penalty_and_no_repeat_ngram(golden):
    """ta_logits_process in the decode loop (inside the captured hipGraph from step 2 on): every HIP decision is the fp32
    oracle's processed argmax (or within the bf16 near-tie tolerance of it) on the HIP path's own prefix; no bigram repeats
    under no_repeat_ngram_size = 2; and the reference's tokens are reproduced wherever its margins are decisive."""
    g = golden("generate_penalties_small.npz")
    S = R.SMALL
    E, D, H = S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"]
    wE, wL, wP = OW.init_encoder(S["enc"], 0), R.gen_lm_weights(), OW.init_mlp_projector(E, D, H)
    m = build_model(S["enc"], S["lm"], H, wE, wL, wP, audio_token_id=S["audio_token_id"], pad_token_id=S["pad_id"],
                    eos_token_id=S["eos_id"])
    kw = dict(input_ids=torch.from_numpy(g["input_ids"]), input_features=torch.from_numpy(g["input_features"]),
              audio_attention_mask=torch.from_numpy(g["audio_attention_mask"]),
              attention_mask=torch.ones(g["input_ids"].shape, dtype=torch.int64))
    W = dict(encoder=wE, lm=wL, projector=wP)
    cfg = dict(enc=S["enc"], lm=S["lm"], projector_type="mlp", k=S["k"], audio_token_id=S["audio_token_id"])
    batch = dict(input_ids=g["input_ids"], input_features=g["input_features"])
    L = g["input_ids"].shape[1]
    for name, opts in (("rep", dict(repetition_penalty=1.3)), ("ngram", dict(no_repeat_ngram_size=2)),
                       ("both", dict(repetition_penalty=1.3, no_repeat_ngram_size=2))):
        out = m.generate(**kw, max_new_tokens=16, **opts).cpu().numpy()
        assert out.shape == (2, 16)
        exact = _check_greedy_against_oracle(out, batch, W, cfg, (S["eos_id"], S["pad_id"]), S["pad_id"], **opts)
        assert exact >= 26, (name, exact)                      # of 32 decisions
        assert (out == g["tokens_" + name]).mean() > 0.5 and (out[:, :5] == g["tokens_" + name][:, :5]).all(), name
        if "no_repeat_ngram_size" in opts:
            for row in out:
                seq = list(g["input_ids"][0]) + list(row)
                big = list(zip(seq[L - 1:-1], seq[L:]))
                assert len(set(big)) == len(big)
    m.config.repetition_penalty = 1.3                            # the setting may also come from the config (asr_config.py:155-160)
    out_cfg = m.generate(**kw, max_new_tokens=16).cpu().numpy()
    assert np.array_equal(out_cfg, m.generate(**kw, max_new_tokens=16, repetition_penalty=1.3).cpu().numpy())


def test_generate_streaming_penalties_see_generated_tokens_only(golden):
    """the reference's generate_streaming gives HF inputs_embeds without input_ids, so its repetition / n-gram processors
    never see the prompt; generate() does.  Fixture: a prompt that contains the model's favourite tokens (the two modes part from
    the first decision on), tokens from the reference's own LM call in both modes."""
    g = golden("generate_penalties_small.npz")
    S = R.SMALL
    E, D, H = S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"]
    wE, wL, wP = OW.init_encoder(S["enc"], 0), R.gen_lm_weights(), OW.init_mlp_projector(E, D, H)
    m = build_model(S["enc"], S["lm"], H, wE, wL, wP, audio_token_id=S["audio_token_id"], pad_token_id=S["pad_id"],
                    eos_token_id=S["eos_id"])
    W = dict(encoder=wE, lm=wL, projector=wP)
    cfg = dict(enc=S["enc"], lm=S["lm"], projector_type="mlp", k=S["k"], audio_token_id=S["audio_token_id"])
    for name, opts in (("rep", dict(repetition_penalty=1.3)), ("both", dict(repetition_penalty=1.3, no_repeat_ngram_size=2))):
        full = m.generate(input_ids=torch.from_numpy(g["input_ids2"]), input_features=torch.from_numpy(g["input_features"]),
                          audio_attention_mask=torch.from_numpy(g["audio_attention_mask"]),
                          attention_mask=torch.ones(g["input_ids2"].shape, dtype=torch.int64), max_new_tokens=16, **opts).cpu().numpy()
        batch2 = dict(input_ids=g["input_ids2"], input_features=g["input_features"])
        _check_greedy_against_oracle(full, batch2, W, cfg, (S["eos_id"], S["pad_id"]), S["pad_id"], **opts)   # prompt + generated tokens
        for b in range(2):                                   # streaming: one clip at a time
            toks = list(m.generate_streaming(torch.from_numpy(g["input_features"][b:b + 1]), torch.from_numpy(g["audio_attention_mask"][b:b + 1]),
                                             input_ids=torch.from_numpy(g["input_ids2"][b:b + 1]), return_token_ids=True,
                                             max_new_tokens=16, **opts))
            got = np.array(toks, dtype=np.int64)[None, :]
            want = g["tokens2_stream_" + name][b:b + 1]
            assert (got[:, :2] == want[:, :2]).all(), (name, b, got, want)      # decisive margins: the reference's own streaming tokens
            assert (got[0, :got.shape[1]] != full[b, :got.shape[1]]).any(), (name, b)   # and not what generate() produces on this prompt
            batch = dict(input_ids=g["input_ids2"][b:b + 1], input_features=g["input_features"][b:b + 1])
            _check_greedy_against_oracle(got, batch, W, cfg, (S["eos_id"], S["pad_id"]), S["pad_id"], processors_see_prompt=False, **opts)


def test_logits_process_kernel_vs_oracle():
    """ta_logits_process alone, exact: random logits and sequences with repeated ids, every (penalty, n-gram) combination,
    step counter read from device memory."""
    from oracle import generate as OG
    from tiny_audio_amd import _lib
    from tiny_audio_amd.ops import ptr, stream
    rng = np.random.RandomState(3)
    B, V, Vp, L, max_new = 3, 200, 256, 9, 12
    for step in (0, 1, 5, 12):
        for pen, ng in ((1.0, 2), (1.7, 0), (1.7, 3), (0.6, 1), (1.2, 2)):
            logits = rng.standard_normal((B, Vp)).astype(np.float32) * 3
            prompt = rng.randint(0, 12, (B, L)).astype(np.int64)           # few distinct ids: many repeated n-grams
            gen = rng.randint(0, 12, (B, max_new)).astype(np.int64)
            seq = np.concatenate([prompt, gen[:, :step]], 1)
            ref = OG.apply_logits_processors(logits[:, :V], seq, pen, ng)
            lg = torch.from_numpy(logits).to(DEV)
            pr, gn = torch.from_numpy(prompt).to(DEV), torch.from_numpy(gen).to(DEV)
            st = torch.tensor([step], dtype=torch.int32, device=DEV)
            _lib.check(_lib.lib().ta_logits_process(ptr(lg), Vp, V, ptr(pr), L, ptr(gn), max_new, ptr(st), B, pen, ng, stream()), "lp")
            got = lg.cpu().numpy()
            np.testing.assert_array_equal(got[:, :V], ref, err_msg=f"step {step} penalty {pen} ngram {ng}")
            np.testing.assert_array_equal(got[:, V:], logits[:, V:])


def test_generate_true_width_ragged_prompts_and_cache_consistency():
    """True layer widths (2+2 layers).  (i) greedy parity vs the oracle; (ii) the KV-cache decode path must agree with a
    full re-run of the training-path forward on prompt+generated tokens (argmax of the last position);
    (iii) left-padded prompts (what HF batched generation expects) give each clip the tokens it gets alone."""
    wE, wL = OW.init_encoder(TRUE_ENC, 0), OW.init_lm(TRUE_LM, 1)
    for k in wL:
        if k.endswith("o_proj.weight") or k.endswith("down_proj.weight"):
            wL[k] = wL[k] * np.float32(8.0)
    wP = OW.init_mlp_projector(1280, 1024, 512)
    m = build_model(TRUE_ENC, TRUE_LM, 512, wE, wL, wP, audio_token_id=AID, pad_token_id=PAD, eos_token_id=EOS)
    x = (0.6 * np.random.RandomState(3).standard_normal((2, 128, 120))).astype(np.float32)        # S = 60 -> 15 audio tokens
    amask = np.ones((2, 120), np.int64)
    n_audio = 15
    ids = np.concatenate([np.arange(5, 8), np.full(n_audio, AID), np.arange(40, 46)])[None, :].repeat(2, 0).astype(np.int64)
    ids[1, -3:] = [77, 78, 79]
    kw = dict(input_features=torch.from_numpy(x), audio_attention_mask=torch.from_numpy(amask))
    out = m.generate(input_ids=torch.from_numpy(ids), attention_mask=torch.ones(ids.shape, dtype=torch.int64), **kw,
                     max_new_tokens=10).cpu().numpy()
    assert out.shape == (2, 10)
    W = dict(encoder=wE, lm=wL, projector=wP)
    cfg = dict(enc=TRUE_ENC, lm=TRUE_LM, projector_type="mlp", k=4, audio_token_id=AID)
    _check_greedy_against_oracle(out, dict(input_ids=ids, input_features=x), W, cfg, (EOS, PAD), PAD)
    # (ii) cache vs full forward of the training path
    full = np.concatenate([ids, out[:, :-1]], axis=1)
    o = m(input_ids=torch.from_numpy(full), input_features=torch.from_numpy(x), attention_mask=torch.ones(full.shape, dtype=torch.int64),
          audio_token_counts=torch.full((2,), n_audio))
    lg = npy(o.logits)
    for t in range(out.shape[1]):
        col = ids.shape[1] - 1 + t
        for b in range(2):
            assert lg[b, col].max() - lg[b, col, out[b, t]] < 0.12, (b, t)
    # (iii) left padding: clip 1 alone == clip 1 in a left-padded batch with a longer clip-0 prompt
    solo = m.generate(input_ids=torch.from_numpy(ids[1:2]), attention_mask=torch.ones((1, ids.shape[1]), dtype=torch.int64),
                      input_features=kw["input_features"][1:2], audio_attention_mask=kw["audio_attention_mask"][1:2],
                      max_new_tokens=6).cpu().numpy()
    ids_l = np.concatenate([np.full((2, 4), PAD), ids], axis=1); att_l = np.ones_like(ids_l); att_l[1, :4] = 0
    ids_l[0, :4] = [9, 10, 11, 12]
    both = m.generate(input_ids=torch.from_numpy(ids_l), attention_mask=torch.from_numpy(att_l), **kw, max_new_tokens=6).cpu().numpy()
    assert (both[1] == solo[0]).mean() >= 5 / 6            # identical up to one bf16 near-tie
    # (iv) streaming (asr_modeling.py:648-760): the same kernels, one sync per token -> the very same tokens
    streamed = list(m.generate_streaming(input_ids=torch.from_numpy(ids[1:2]), input_features=kw["input_features"][1:2],
                                         audio_attention_mask=kw["audio_attention_mask"][1:2], max_new_tokens=6,
                                         return_token_ids=True))
    assert streamed == solo[0].tolist()[:len(streamed)] and len(streamed) >= min(solo.shape[1], 6)


# ============================================================================ size-independent properties at full shapes
def test_full_size_properties():
    """BASELINE config: full depth (32 + 28 layers), 10 s clips, L = 192 -- too big for the oracle,
    so check invariants: finite loss near ln(V) for random weights, batch-permutation equivariance of per-token
    NLL, gradient linearity in the loss scale, zero gradient for a clip whose labels are all ignored."""
    torch.manual_seed(0)
    cfg = ASRConfig(audio_token_dropout=0.0)
    m = ASRModel(cfg, device=DEV, init="random", seed=0)
    B, L = 3, 192
    V = cfg.text_config.vocab_size
    feats = torch.randn(B, 128, 1000) * 0.5
    ids, att, lab, counts = OW.synthetic_tokens(B, 125, V, cfg.audio_token_id, cfg.pad_token_id, cfg.eos_token_id, L=L)
    lab[2] = -100                                               # third clip carries no labels
    tb = dict(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(att), labels=torch.from_numpy(lab),
              audio_token_counts=torch.from_numpy(counts))
    m.train()
    out = m(input_features=feats, **tb, return_logits=False)
    assert out.n_label_tokens == 72 and np.isfinite(float(out.loss))
    assert abs(float(out.loss) - np.log(V)) < 3.0
    out.loss.backward()
    g1 = {k: p.grad.clone() for k, p in m.projector.named_parameters()}
    nll1 = out.nll.clone()
    # permutation of the batch permutes per-token NLLs and leaves the (mean) gradient unchanged
    perm = [1, 0, 2]
    m.zero_grad()
    out2 = m(input_features=feats[perm], **{k: v[perm] for k, v in tb.items()}, return_logits=False)
    out2.loss.backward()
    assert torch.allclose(torch.cat([nll1[36:72], nll1[0:36]]), out2.nll, rtol=2e-3, atol=2e-3)
    for k, p in m.projector.named_parameters():
        assert cosine(npy(p.grad), npy(g1[k])) > 0.9995, k
    # gradient scales linearly with 1/num_items
    m.zero_grad()
    out3 = m(input_features=feats, **tb, return_logits=False, num_items_in_batch=36)
    out3.loss.backward()
    for k, p in m.projector.named_parameters():
        assert cosine(npy(p.grad), npy(g1[k])) > 0.99999 and abs(float(p.grad.norm() / g1[k].norm()) - 2.0) < 2e-3
    # a batch made only of the unlabelled clip: loss 0, all gradients exactly 0
    m.zero_grad()
    out4 = m(input_features=feats[2:], **{k: v[2:] for k, v in tb.items()}, return_logits=False)
    assert out4.n_label_tokens == 0 and float(out4.loss) == 0.0
    out4.loss.backward()
    for k, p in m.projector.named_parameters():
        assert float(p.grad.abs().max()) == 0.0, k


# ============================================================================ MoE projector (BASELINE configs[3])
def _moe(cfg_kw, w, **extra):
    from tiny_audio_amd.projectors import MoEAudioProjector
    cfg = ASRConfig(projector_type="moe", **cfg_kw, **extra)
    p = MoEAudioProjector(cfg).to(DEV)
    p.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return p


def test_moe_projector_vs_golden(golden):
    g = golden("projector_moe.npz")
    S = R.SMALL
    E, D, H = S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"]
    w = OW.init_moe_projector(E, D, H)
    p = _moe(dict(audio_config=S["enc"], text_config=S["lm"], projector_hidden_dim=H), w, router_jitter_noise=0.0)
    x, dy = R.proj_input()
    xb = torch.from_numpy(x).to(DEV).to(torch.bfloat16)
    p.eval()
    y = p(xb)
    assert relmax(npy(y), g["y_eval"]) < 2e-2 and float(p.get_aux_loss()) == 0.0
    p.train()
    y = p(xb)
    aux = p.get_aux_loss()
    assert relmax(npy(y), g["y_train"]) < 2e-2
    assert abs(float(aux) - float(g["aux_train"])) < 2e-2 * abs(float(g["aux_train"])) + 1e-7
    ((y * torch.from_numpy(dy).to(DEV)).sum() + 3.0 * aux).backward()
    for k, prm in p.named_parameters():
        ref = g["gt." + k]
        if np.abs(ref).max() == 0:
            assert float(prm.grad.abs().max()) == 0.0, k
        else:
            assert cosine(npy(prm.grad), ref) > 0.995, (k, cosine(npy(prm.grad), ref))


def test_moe_projector_true_width_vs_oracle_with_jitter():
    E, D, H = 1280, 1024, 1024
    w = OW.init_moe_projector(E, D, H)
    p = _moe(dict(audio_config=TRUE_ENC, text_config=TRUE_LM, projector_hidden_dim=H), w)
    rng = np.random.RandomState(9)
    x = rng.standard_normal((3, 101, E)).astype(np.float32)                      # 101 frames -> 25 tokens, tail dropped
    xb = torch.from_numpy(x).to(DEV).to(torch.bfloat16)
    noise = rng.uniform(0.99, 1.01, size=(75, 4)).astype(np.float32)
    p.train()
    y = p(xb, jitter_noise=torch.from_numpy(noise))
    aux = p.get_aux_loss()
    ref, ref_aux, c = OP.moe_forward(npy(xb), w, training=True, jitter_noise=noise)
    assert y.shape == (3, 25, D)
    # routing is discrete: with bf16 activations a near-tie may flip an expert for a token; compare token-wise
    bad = (np.abs(npy(y) - ref).max(-1) > 2e-2 * np.abs(ref).max()).mean()
    assert bad < 0.03, bad
    assert abs(float(aux) - float(ref_aux)) < 2e-2 * float(ref_aux)
    dy = rng.standard_normal(ref.shape).astype(np.float32)
    (y * torch.from_numpy(dy).to(DEV)).sum().add(aux).backward()
    gr = OP.moe_backward(dy, w, c, d_aux=1.0)
    for k, prm in p.named_parameters():
        assert cosine(npy(prm.grad), gr[k]) > 0.99, (k, cosine(npy(prm.grad), gr[k]))


def test_asr_model_moe_vs_golden(golden):
    g = golden("asr_small.npz")
    S = R.SMALL
    E, D, H = S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"]
    cfg = ASRConfig(audio_config=S["enc"], text_config=S["lm"], projector_hidden_dim=H, projector_type="moe",
                    audio_token_id=S["audio_token_id"], router_jitter_noise=0.0)
    m = ASRModel(cfg, device=DEV, init="none")
    m.audio_tower.load_state_dict_hf(OW.init_encoder(S["enc"], 0))
    m.language_model.load_state_dict_hf(OW.init_lm(S["lm"], 1))
    m.load_state_dict({"projector." + k: torch.from_numpy(v) for k, v in OW.init_moe_projector(E, D, H).items()})
    ids, att, lab, counts = R.asr_tokens(g["counts"])
    m.train()
    out = m(input_ids=torch.from_numpy(ids), input_features=torch.from_numpy(g["input_features"]),
            attention_mask=torch.from_numpy(att), labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts))
    out.loss.backward()
    assert abs(float(out.loss) - float(g["moe.loss"])) < 5e-3 * float(g["moe.loss"])
    assert abs(float(out.aux_loss) - float(g["moe.aux"])) < 2e-2 * abs(float(g["moe.aux"])) + 1e-7
    for k in [k[len("moe.g."):] for k in g.files if k.startswith("moe.g.")]:
        got = dict(m.projector.named_parameters())[k].grad
        assert cosine(npy(got), g["moe.g." + k]) > 0.995, k


# ============================================================================ full decoder fine-tuning (8(f) rank 4)
def _ft_grads_hf(m):
    """Parameter.grad of the stacked masters, under the reference's names."""
    lm = m.language_model
    saved = [p.data for p in lm.ft_parameters()]
    for p in lm.ft_parameters():
        p.data = p.grad
    try:
        return {k: npy(v) for k, v in lm.ft_state_dict_hf().items()}
    finally:
        for p, d in zip(lm.ft_parameters(), saved):
            p.data = d


def test_full_finetune_vs_golden(golden):
    """freeze_language_model=False on the reference's own fixture: one backward (every LM weight gradient) and 3 steps of
    the split-group optimizer (projector lr 1e-3, decoder lr 1e-4), bf16 kernels vs the reference's fp32 autograd."""
    g, gf = golden("asr_small.npz"), golden("fullft_small.npz")
    S = R.SMALL
    E, D, H = S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"]
    mk = lambda: build_model(S["enc"], S["lm"], H, OW.init_encoder(S["enc"], 0), OW.init_lm(S["lm"], 1),
                             OW.init_mlp_projector(E, D, H), audio_token_id=S["audio_token_id"], freeze_language_model=False)
    ids, att, lab, counts = R.asr_tokens(g["counts"])
    batch = dict(input_ids=torch.from_numpy(ids), input_features=torch.from_numpy(g["input_features"]),
                 attention_mask=torch.from_numpy(att), labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts))
    m = mk()
    assert m.language_model.train_base and sum(p.numel() for p in m.parameters() if p.requires_grad) == int(gf["n_trainable"])
    m.train()
    out = m(**batch)
    out.loss.backward()
    assert abs(float(out.loss) - float(gf["loss"])) < 5e-3 * float(gf["loss"])
    gl = _ft_grads_hf(m)
    n = 0
    for k in gf.files:
        if not k.startswith("g.language_model."):
            continue
        name = k[len("g.language_model."):]
        mine = R.fullft_select(name, gl[name])
        assert cosine(mine, gf[k]) > 0.995 and relmax(mine, gf[k]) < 6e-2, (name, cosine(mine, gf[k]), relmax(mine, gf[k]))
        n += 1
    assert n >= 17
    for k, prm in m.projector.named_parameters():
        assert cosine(npy(prm.grad), gf["g.projector." + k]) > 0.999, k
    m = mk(); m.train()
    tr = ASRTrainer(m, TrainingArguments(learning_rate=1e-3, weight_decay=0.0, max_grad_norm=1.0), decoder_learning_rate=1e-4)
    losses, gnorms = [], []
    for _ in range(3):
        tr.training_step(batch)
        losses.append(tr.last_loss()); gnorms.append(tr.last_grad_norm())
    np.testing.assert_allclose(losses, gf["losses"], rtol=5e-3)
    np.testing.assert_allclose(gnorms, gf["gnorms"], rtol=3e-2)
    assert losses[2] < losses[0]
    sd = {k: npy(v) for k, v in m.language_model.ft_state_dict_hf().items()}
    for k in gf.files:
        if k.startswith("w.language_model."):
            name = k[len("w.language_model."):]
            d = np.abs(R.fullft_select(name, sd[name]) - gf[k])
            assert d.mean() < 3e-5, (name, d.mean())        # 3 steps * lr 1e-4 of travel per element at most


def test_full_finetune_true_width_vs_oracle():
    """True Qwen3-0.6B widths (2 layers): every weight gradient of the trainable LM against the fp32 oracle, including the
    tied embedding (lm_head share over the vocabulary + input-lookup share) and gradient accumulation over two calls."""
    wE, wL = OW.init_encoder(TRUE_ENC, 0), OW.init_lm(TRUE_LM, 1)
    wP = OW.init_mlp_projector(1280, 1024, 1024)
    m = build_model(TRUE_ENC, TRUE_LM, 1024, wE, wL, wP, audio_token_id=AID, freeze_language_model=False)
    b = _true_batch()
    m.train()
    tb = {k: torch.from_numpy(v) for k, v in b.items()}
    out = m(**tb)
    out.loss.backward()
    W = dict(encoder=wE, lm=wL, projector={k: v.copy() for k, v in wP.items()})
    cfg = dict(enc=TRUE_ENC, lm=TRUE_LM, projector_type="mlp", k=4, audio_token_id=AID, freeze_language_model=False)
    ref = OM.asr_forward(b, W, cfg, training=True)
    grads, _ = OM.asr_backward(ref, W, cfg, b)
    assert abs(float(out.loss) - float(ref["loss"])) < 5e-3 * float(ref["loss"])
    gl = _ft_grads_hf(m)
    assert set(gl) == {k[len("language_model."):] for k in grads if k.startswith("language_model.")}
    for k, v in gl.items():
        r = grads["language_model." + k]
        assert cosine(v, r) > 0.995, (k, cosine(v, r))
        assert abs(np.linalg.norm(v) - np.linalg.norm(r)) < 3e-2 * np.linalg.norm(r), k
    for k, prm in m.projector.named_parameters():
        assert cosine(npy(prm.grad), grads[k]) > 0.999, k
    # a second micro-batch accumulates (autograd adds the returned gradients)
    out = m(**tb); out.loss.backward()
    g2 = _ft_grads_hf(m)
    for k in ("model.layers.1.mlp.down_proj.weight", "model.embed_tokens.weight", "model.layers.0.self_attn.q_norm.weight"):
        assert relmax(g2[k], 2 * gl[k]) < 2e-2, k
