"""-m gpu, round 5: the numerics contract of the headline (VERDICT r04 "Next round" item 1) and the boundary (item 6).

(a) The benchmarked model (32 + 28 layers, V = 151 670, MLP H = D = 1024) on numpy-seeded weights, one 10 s clip of the bench batch,
    in BOTH residual-stream storage modes, against tests/golden/asr_full_recipe.npz -- outputs of the REFERENCE itself run three
    ways on the same weights: fp32, fp32 modules + bf16 autocast (the training recipe, configs/config.yaml:14-18 +
    configs/training/production.yaml:49), bf16 modules (ASRConfig's default, tiny_audio/asr_config.py:41).

    Gates, fixed BEFORE the first GPU run of this test from the fixture alone:
      * BASELINE.md section 4 as written where the reference's own bf16 step meets it: loss relative <= 5e-3, projector-gradient
        cosine >= 0.999 against the reference's fp32 run;
      * "logits atol 5e-2" is NOT met by the reference's own recipe at this depth -- its autocast run sits 0.095 (max-abs) /
        0.0166 (RMS) from its fp32 run, its bf16-module run 0.122 / 0.0228 -- so the logits / NLL gates are RELATIVE to the
        reference regime a storage mode mirrors (fp32 streams <-> autocast, bf16 streams <-> bf16 modules), on the same row /
        column sample: RMS <= 1.25x, max-abs <= 1.4x (an extreme-value statistic) of that regime's own distance from fp32.
    Everything measured is written to gpurun_out/r05_recipe_drift.json next to the reference's figures.

(b) The oracle-based full-depth tests of round 3 (MLP / MoE / LoRA on GPU-random weights) in the fp32-stream mode as well
    (they run in the default bf16-stream mode in tests/test_gpu_round3.py).

(c) ``position_ids`` reach the LM (ABI ``pos``), ``inputs_embeds`` / ``past_key_values`` / ``use_cache=True`` raise.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import features as OF
from oracle import qwen3 as OQ
from oracle import weights as OW
from tests.golden import recipe as R

if torch.cuda.is_available():
    from tiny_audio_amd import ops
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    from tiny_audio_amd.asr_processing import LogMelFeatureExtractor

DEV = "cuda"
LOSS_REL, GRAD_COS = 5e-3, 0.999
RMS_FACTOR, MAXABS_FACTOR = 1.25, 1.4              # x the mirrored reference regime's own distance from fp32
MIRROR = {"f32": "autocast", "bf16": "bf16"}       # storage mode -> the reference regime that stores the same way


def npy(t):
    return t.detach().float().cpu().numpy()


def cosine(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))


def _record(name, rec, fname="r05_recipe_drift.json"):
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, fname)
        cur = {}
        if os.path.exists(path):
            with open(path) as fh:
                cur = json.load(fh)
        cur[name] = rec
        with open(path, "w") as fh:
            json.dump(cur, fh, indent=1, sort_keys=True)
    except OSError:
        pass
    print(f"[recipe drift] {name}: {json.dumps(rec)}")


_FULL_W = {}


def _full_weights():
    """numpy-seeded weights at the benchmarked shape (oracle.weights; ~45 s of single-threaded RandomState), shared by both modes"""
    if not _FULL_W:
        F = R.FULL
        _FULL_W.update(enc=OW.init_encoder(F["enc"], 0), lm=OW.init_lm(F["lm"], 1), proj=OW.init_mlp_projector(1280, 1024, 1024))
    return _FULL_W


# ============================================================================ (a) both storage modes vs the reference's three runs
@pytest.mark.parametrize("streams", ["bf16", "f32"])
def test_benchmarked_model_vs_reference_recipe_fixture(golden, streams):
    g = golden("asr_full_recipe.npz")
    W = _full_weights()
    cfg = ASRConfig(model_dtype="float32" if streams == "f32" else "bfloat16", audio_token_dropout=0.0)
    assert cfg.text_config.vocab_size == 151670 and cfg.audio_config.num_hidden_layers == 32 and cfg.text_config.num_hidden_layers == 28
    m = ASRModel(cfg, device=DEV, init="none")
    m.audio_tower.load_state_dict_hf(W["enc"])
    m.language_model.load_state_dict_hf(W["lm"])
    m.load_state_dict({"projector." + k: torch.from_numpy(v) for k, v in W["proj"].items()})
    m.train()
    ids, att, lab, counts = R.full_clip_tokens()
    f = LogMelFeatureExtractor(128, DEV)([OW.synthetic_wave(0)], sampling_rate=16000)
    out = m(input_ids=torch.from_numpy(ids), input_features=f["input_features"], attention_mask=torch.from_numpy(att),
            labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts), return_logits=True)
    assert (m.audio_tower.res_f32, m.language_model.res_f32, m.language_model.dx_f32) == (streams == "f32",) * 3
    out.loss.backward()
    torch.cuda.synchronize()
    rows = g["rows"]
    np.testing.assert_array_equal(rows, R.full_logit_rows(att[0], lab[0]))
    lg = npy(out.logits[0])[rows][:, ::R.FULL_LOGIT_COL_STRIDE].astype(np.float64)
    nll = npy(out.nll).astype(np.float64)
    ref = {k: g[f"{k}.logits_sample"].astype(np.float64) for k in ("fp32", "autocast", "bf16")}
    rms = lambda d: float(np.sqrt((d ** 2).mean()))
    mab = lambda d: float(np.abs(d).max())
    regime = MIRROR[streams]
    rec = {"streams": streams, "mirrors_reference_regime": regime,
           "loss_hip": float(out.loss.detach()), "loss_ref_fp32": float(g["fp32.loss"]), "loss_ref_regime": float(g[f"{regime}.loss"]),
           "logits_maxabs_vs_fp32": mab(lg - ref["fp32"]), "logits_rms_vs_fp32": rms(lg - ref["fp32"]),
           "logits_maxabs_vs_regime": mab(lg - ref[regime]), "logits_rms_vs_regime": rms(lg - ref[regime]),
           "ref_regime_logits_maxabs_vs_fp32_same_sample": mab(ref[regime] - ref["fp32"]),
           "ref_regime_logits_rms_vs_fp32_same_sample": rms(ref[regime] - ref["fp32"]),
           "ref_regime_logits_maxabs_vs_fp32_all_rows_x_V": float(g[f"{regime}.logits_maxabs_vs_fp32"]),
           "ref_regime_logits_rms_vs_fp32_all_rows_x_V": float(g[f"{regime}.logits_rms_vs_fp32"]),
           "nll_maxabs_vs_fp32": mab(nll - g["fp32.nll"]), "nll_rms_vs_fp32": rms(nll - g["fp32.nll"]),
           "ref_regime_nll_maxabs_vs_fp32": float(g[f"{regime}.nll_maxabs_vs_fp32"]), "ref_regime_nll_rms_vs_fp32": float(g[f"{regime}.nll_rms_vs_fp32"])}
    rec["loss_rel_vs_fp32"] = abs(rec["loss_hip"] - rec["loss_ref_fp32"]) / rec["loss_ref_fp32"]
    gc, gc_reg, gc_ref = {}, {}, {}
    for k, p in m.projector.named_parameters():
        mine = R.full_grad_sample(k, npy(p.grad))
        gc[k] = cosine(mine, g["fp32.g." + k]); gc_reg[k] = cosine(mine, g[f"{regime}.g.{k}"])
        gc_ref[k] = float(g[f"{regime}.gcos_vs_fp32.{k}"])
        rec[f"gnorm_ratio.{k}"] = float(np.linalg.norm(npy(p.grad).astype(np.float64)) / float(g["fp32.gnorm." + k]))
    rec.update(grad_cos_vs_fp32_min=min(gc.values()), grad_cos_vs_regime_min=min(gc_reg.values()),
               ref_regime_grad_cos_vs_fp32_min=min(gc_ref.values()), grad_cos_vs_fp32=gc)
    _record(streams, rec)
    # ---- gates (module docstring)
    assert rec["loss_rel_vs_fp32"] < LOSS_REL, rec
    assert rec["grad_cos_vs_fp32_min"] > GRAD_COS, rec
    assert all(0.98 < rec[f"gnorm_ratio.{k}"] < 1.02 for k in gc), rec
    assert rec["logits_rms_vs_fp32"] <= RMS_FACTOR * rec["ref_regime_logits_rms_vs_fp32_same_sample"], rec
    assert rec["logits_maxabs_vs_fp32"] <= MAXABS_FACTOR * rec["ref_regime_logits_maxabs_vs_fp32_same_sample"], rec
    assert rec["nll_rms_vs_fp32"] <= RMS_FACTOR * rec["ref_regime_nll_rms_vs_fp32"], rec
    assert rec["nll_maxabs_vs_fp32"] <= MAXABS_FACTOR * rec["ref_regime_nll_maxabs_vs_fp32"], rec
    # two independent bf16 roundings of the same fp32 function: the distance between them stays below the sum of their distances
    assert rec["logits_rms_vs_regime"] <= rec["logits_rms_vs_fp32"] + rec["ref_regime_logits_rms_vs_fp32_same_sample"], rec
    # The HEADLINE question (added after the first run of this test, with the factors above unchanged): is the default bf16-stream
    # mode also within the same factors of the RECIPE's regime -- the reference's fp32 modules under autocast?  First measurement
    # (profiles/r05_a_recipe_drift.json): logits RMS 0.0204 vs the recipe's 0.0169 (1.21x), max-abs 0.101 vs 0.091 (1.12x), NLL RMS
    # 0.0176 vs 0.0180, NLL max-abs 0.039 vs 0.040, gradient cosine 0.99958 vs 0.99965 -- i.e. as far from the fp32 function as the
    # reference's own training recipe is on the quantities training consumes, 1.2x on raw logits.
    rec_reg = {"logits_rms": rms(ref["autocast"] - ref["fp32"]), "logits_maxabs": mab(ref["autocast"] - ref["fp32"]),
               "nll_rms": float(g["autocast.nll_rms_vs_fp32"]), "nll_maxabs": float(g["autocast.nll_maxabs_vs_fp32"]),
               "grad_cos_min": min(float(g[f"autocast.gcos_vs_fp32.{k}"]) for k in gc)}
    assert rec["logits_rms_vs_fp32"] <= RMS_FACTOR * rec_reg["logits_rms"] and rec["logits_maxabs_vs_fp32"] <= MAXABS_FACTOR * rec_reg["logits_maxabs"], (rec, rec_reg)
    assert rec["nll_rms_vs_fp32"] <= RMS_FACTOR * rec_reg["nll_rms"] and rec["nll_maxabs_vs_fp32"] <= MAXABS_FACTOR * rec_reg["nll_maxabs"], (rec, rec_reg)
    assert 1.0 - rec["grad_cos_vs_fp32_min"] <= RMS_FACTOR ** 2 * (1.0 - rec_reg["grad_cos_min"]), (rec, rec_reg)     # 1 - cos ~ (relative error)^2


# ============================================================================ (b) the oracle-based full-depth tests in the fp32-stream mode
@pytest.mark.parametrize("kind", ["mlp", "moe", "lora"])
def test_full_depth_one_clip_vs_oracle_f32_streams(kind, monkeypatch):
    """tests/test_gpu_round3.py::test_full_depth_one_clip_vs_oracle with ``model_dtype="float32"`` (fp32 residual streams): same
    stated tolerances; the drift is recorded as ``<kind>_f32_streams`` beside the bf16-stream figures of the round-3 test."""
    from tests import test_gpu_round3 as T3
    real = T3.ASRConfig
    monkeypatch.setattr(T3, "ASRConfig", lambda **kw: real(model_dtype="float32", **kw))
    seen = {}
    orig_record = T3._record_drift

    def record(name, rec):
        seen["rec"] = rec
        orig_record(name + "_f32_streams", rec)
    monkeypatch.setattr(T3, "_record_drift", record)
    T3.test_full_depth_one_clip_vs_oracle(kind)
    _record(kind + "_vs_oracle_f32_streams", seen.get("rec", {}), fname="r05_full_depth_drift_both_modes.json")


# ============================================================================ (c) the boundary: position_ids, refused HF decoding arguments
def _small_asr(lm_layers=2):
    enc, lm = OW.enc_config(256, 512, 1, 4), OW.lm_config(1024, 256, 512, lm_layers, 4, 2, 128)
    cfg = ASRConfig(audio_config=enc, text_config=lm, projector_hidden_dim=128, audio_token_id=1023, pad_token_id=1000, eos_token_id=1001)
    m = ASRModel(cfg, device=DEV, init="random", seed=3)
    return m, cfg, lm


def test_position_ids_reach_the_lm(golden):
    """ASRModel.forward(position_ids=...) -> ta_lm_forward_loss(pos) (tiny_audio/asr_modeling.py:517-526): the LM of the reference
    fixture qwen3_posids_small.npz behind ASRModel.forward, text-only batch (inputs_embeds = embedding rows), left padding + a
    position gap.  Compared with the oracle on the same weights (itself pinned on that fixture) -- loss, logits, and the
    arange(L) answer must differ."""
    m, cfg, lmc = _small_asr()
    wl = OW.init_lm(R.SMALL["lm"], seed=1)
    m.language_model.load_state_dict_hf(wl)
    _, att, lab, pos = R.lm_input_leftpad()
    rng = np.random.RandomState(5)
    ids = rng.randint(0, 990, att.shape).astype(np.int64)
    lab = np.where(lab != -100, ids, -100)                               # labels = the tokens themselves where the fixture has labels
    x = wl["model.embed_tokens.weight"][ids]
    logits_o, _ = OQ.lm_forward(x, att, wl, R.SMALL["lm"], position_ids=pos)
    ce_o = float(OQ.causal_lm_loss(logits_o, lab)[0])
    logits_a, _ = OQ.lm_forward(x, att, wl, R.SMALL["lm"])
    ce_a = float(OQ.causal_lm_loss(logits_a, lab)[0])
    assert abs(ce_o - ce_a) > 1e-2
    T = torch.from_numpy
    m.eval()
    with torch.no_grad():
        out = m(input_ids=T(ids), attention_mask=T(att), labels=T(lab), position_ids=T(pos))
        out_a = m(input_ids=T(ids), attention_mask=T(att), labels=T(lab))
        out_b = m(input_ids=T(ids), attention_mask=T(att), labels=T(lab), position_ids=T(pos[1:2] * 0 + np.arange(att.shape[1])))   # [1, L] broadcast
    rows = att.astype(bool)
    assert abs(float(out.loss) - ce_o) < 5e-3 * ce_o, (float(out.loss), ce_o)
    assert abs(float(out_a.loss) - ce_a) < 5e-3 * ce_a and abs(float(out_b.loss) - float(out_a.loss)) < 1e-6
    d = np.abs(npy(out.logits)[rows] - logits_o[rows]).max() / np.abs(logits_o[rows]).max()
    assert d < 2e-2, d
    assert np.abs(npy(out.logits)[rows] - npy(out_a.logits)[rows]).max() > 10 * np.abs(npy(out.logits)[rows] - logits_o[rows]).max()
    with pytest.raises(ValueError):
        m(input_ids=T(ids), attention_mask=T(att), position_ids=T(pos[:, :-1]))


def test_position_ids_backward_matches_oracle():
    """position_ids also steer the backward (the tape's RoPE is undone with the same positions): d(loss)/d(projector) of a whole
    ASRModel step with non-default positions vs the oracle."""
    from oracle import model as OM
    enc, lmc = OW.enc_config(256, 512, 1, 4), OW.lm_config(1024, 256, 512, 2, 4, 2, 128)
    AID, PAD, EOS = 1023, 1000, 1001
    wE, wL, wP = OW.init_encoder(enc, 0), OW.init_lm(lmc, 1), OW.init_mlp_projector(256, 256, 128)
    cfg = ASRConfig(audio_config=enc, text_config=lmc, projector_hidden_dim=128, audio_token_id=AID, pad_token_id=PAD, eos_token_id=EOS)
    m = ASRModel(cfg, device=DEV, init="none")
    m.audio_tower.load_state_dict_hf(wE); m.language_model.load_state_dict_hf(wL)
    m.load_state_dict({"projector." + k: torch.from_numpy(v) for k, v in wP.items()})
    fe = LogMelFeatureExtractor(128, DEV)
    f = fe([OW.synthetic_wave(0, 16000), OW.synthetic_wave(1, 12000)], sampling_rate=16000)
    mel = f["attention_mask"].sum(-1).cpu().numpy()
    counts = (((mel - 1) // 2 + 1) - 4) // 4 + 1
    ids, att, lab, counts = OW.synthetic_tokens(2, counts.tolist(), lmc["vocab"], AID, PAD, EOS, n_text=10, n_suffix=4, ragged=True)
    L = ids.shape[1]
    pos = np.tile(np.arange(L), (2, 1)); pos[0, 8:] += 11; pos[1] = pos[1] * 2
    m.train()
    T = torch.from_numpy
    out = m(input_ids=T(ids), input_features=f["input_features"], attention_mask=T(att), labels=T(lab), audio_token_counts=T(counts),
            position_ids=T(pos), return_logits=False)
    out.loss.backward()
    batch = dict(input_ids=ids, attention_mask=att, labels=lab, input_features=npy(f["input_features"]), audio_token_counts=counts,
                 position_ids=pos)
    W = dict(encoder=wE, lm=wL, projector=wP)
    ocfg = dict(enc=enc, lm=lmc, projector_type="mlp", k=4, audio_token_id=AID)
    ref = OM.asr_forward(batch, W, ocfg, training=True)
    grads, _ = OM.asr_backward(ref, W, ocfg)
    ref0 = OM.asr_forward(dict(batch, position_ids=None), W, ocfg, training=True)
    assert abs(float(ref["loss"]) - float(ref0["loss"])) > 1e-3
    assert abs(float(out.loss) - float(ref["loss"])) < 5e-3 * float(ref["loss"])
    for k, p in m.projector.named_parameters():
        assert cosine(npy(p.grad), grads[k]) > 0.999, k


@pytest.mark.parametrize("kw", [dict(inputs_embeds=1), dict(past_key_values=1), dict(use_cache=True), dict(cache_position=1)])
def test_forward_refuses_hf_decoding_arguments(kw):
    """tiny_audio/asr_modeling.py:481-526 forwards these to the HF LM; this forward has no such protocol and says so."""
    m, cfg, _ = _small_asr(1)
    ids = torch.zeros((1, 8), dtype=torch.int64)
    if "inputs_embeds" in kw:
        kw = dict(inputs_embeds=torch.zeros((1, 8, 256)))
    with pytest.raises(NotImplementedError):
        m(input_ids=ids, **kw)
    m(input_ids=ids, use_cache=False)                                     # HF Trainer's eval loop passes use_cache=False: accepted


def test_stream_modes_follow_model_dtype_and_do_not_change_sizes():
    """ta_encoder_weights.res_f32 / ta_lm_weights.res_f32, dx_f32 <- ASRConfig.model_dtype (include/ta355.h): both storage modes run the same model in one process and
    agree to bf16-storage rounding; the tape / workspace sizes do not depend on the mode."""
    enc, lmc = OW.enc_config(256, 512, 2, 4), OW.lm_config(1024, 256, 512, 2, 4, 2, 128)
    losses, grads = {}, {}
    for dt in ("bfloat16", "float32", "bfloat16"):
        cfg = ASRConfig(audio_config=enc, text_config=lmc, projector_hidden_dim=128, audio_token_id=1023, pad_token_id=1000,
                        eos_token_id=1001, model_dtype=dt)
        torch.manual_seed(0)                                              # (the projector's nn.Linear init draws from the global generator)
        m = ASRModel(cfg, device=DEV, init="random", seed=0)
        f = LogMelFeatureExtractor(128, DEV)([OW.synthetic_wave(0, 32000)], sampling_rate=16000)
        ids, att, lab, counts = OW.synthetic_tokens(1, 25, 1024, 1023, 1000, 1001, n_text=10, n_suffix=4)
        T = torch.from_numpy
        m.train()
        out = m(input_ids=T(ids), input_features=f["input_features"], attention_mask=T(att), labels=T(lab), audio_token_counts=T(counts))
        out.loss.backward()
        want = dt == "float32"
        assert (m.audio_tower._w.res_f32, m.language_model._w.res_f32, m.language_model._w.dx_f32) == (int(want),) * 3
        losses.setdefault(dt, []).append(float(out.loss))
        grads.setdefault(dt, []).append(npy(m.projector.linear_1.weight.grad))
    # switching back restores the bf16-stream result (up to the float atomics of the loss sum / split-K weight gradients: ~1e-7)
    assert abs(losses["bfloat16"][0] - losses["bfloat16"][1]) < 1e-5 * losses["bfloat16"][0]
    assert cosine(grads["bfloat16"][0], grads["bfloat16"][1]) > 0.999999
    assert abs(losses["float32"][0] - losses["bfloat16"][0]) < 2e-2 * losses["float32"][0]
    assert cosine(grads["float32"][0], grads["bfloat16"][0]) > 0.995


def test_backward_reads_its_tape_in_the_mode_it_was_recorded_in():
    """The tape's residual rows are stored in the mode the handle carried at the forward: a backward that runs after the owner
    flipped ``res_f32`` must still read its tape as recorded (language_model.py keeps the mode per tape).  Forward in bf16 streams,
    handle switched to fp32 before the backward: same gradient as undisturbed."""
    enc, lmc = OW.enc_config(256, 512, 2, 4), OW.lm_config(1024, 256, 512, 2, 4, 2, 128)
    grads = []
    for disturb in (False, True):
        cfg = ASRConfig(audio_config=enc, text_config=lmc, projector_hidden_dim=128, audio_token_id=1023, pad_token_id=1000,
                        eos_token_id=1001, model_dtype="bfloat16", audio_token_dropout=0.0)
        torch.manual_seed(0)
        m = ASRModel(cfg, device=DEV, init="random", seed=0)
        f = LogMelFeatureExtractor(128, DEV)([OW.synthetic_wave(0, 32000)], sampling_rate=16000)
        ids, att, lab, counts = OW.synthetic_tokens(1, 25, 1024, 1023, 1000, 1001, n_text=10, n_suffix=4)
        T = torch.from_numpy
        m.train()
        out = m(input_ids=T(ids), input_features=f["input_features"], attention_mask=T(att), labels=T(lab), audio_token_counts=T(counts))
        if disturb:
            m.language_model.res_f32 = m.language_model.dx_f32 = True      # what the owner's next float32 forward would set
        out.loss.backward()
        if disturb:
            assert m.language_model.res_f32 and m.language_model._w.res_f32 == 1      # ... and is left as found
        g = npy(m.projector.linear_1.weight.grad)
        assert np.isfinite(g).all()
        grads.append(g)
    assert cosine(grads[0], grads[1]) > 0.999999
    np.testing.assert_allclose(grads[1], grads[0], rtol=0, atol=1e-5 * float(np.abs(grads[0]).max()))


def test_two_models_of_different_stream_modes_interleaved_from_two_threads():
    """a bf16-stream and an f32-stream model in ONE process, forward / backward of both interleaved from two
    Python threads (the reference's generate_streaming runs its LM on a second thread, tiny_audio/asr_modeling.py:733-734): each
    model's loss and projector gradients equal its own single-model run (to the 1e-6 of the loss sum's float atomics; the two modes
    differ from each other by ~1e-3 ... 1e-2, four orders above that).  With the process-wide mode of rounds 1-5 the two forwards
    raced on it."""
    import threading
    enc, lmc = OW.enc_config(256, 512, 2, 4), OW.lm_config(1024, 256, 512, 2, 4, 2, 128)
    T = torch.from_numpy

    def build(dt):
        cfg = ASRConfig(audio_config=enc, text_config=lmc, projector_hidden_dim=128, audio_token_id=1023, pad_token_id=1000,
                        eos_token_id=1001, model_dtype=dt, audio_token_dropout=0.0)
        torch.manual_seed(0)
        m = ASRModel(cfg, device=DEV, init="random", seed=0)
        m.train()
        return m
    f = LogMelFeatureExtractor(128, DEV)([OW.synthetic_wave(0, 32000), OW.synthetic_wave(1, 32000)], sampling_rate=16000)
    ids, att, lab, counts = OW.synthetic_tokens(2, 25, 1024, 1023, 1000, 1001, n_text=10, n_suffix=4)
    feats = f["input_features"].clone()

    def step(m):
        m.zero_grad(set_to_none=True)
        out = m(input_ids=T(ids), input_features=feats, attention_mask=T(att), labels=T(lab), audio_token_counts=T(counts),
                return_logits=False)
        out.loss.backward()
        torch.cuda.synchronize()
        return float(out.loss), {k: p.grad.clone() for k, p in m.projector.named_parameters()}
    models = {dt: build(dt) for dt in ("bfloat16", "float32")}
    alone = {dt: step(m) for dt, m in models.items()}
    assert alone["bfloat16"][0] != alone["float32"][0]                         # the two modes are different functions
    got, errs = {dt: [] for dt in models}, []
    gate = threading.Barrier(2)

    def worker(dt):
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.default_stream())
            with torch.cuda.stream(s):
                for _ in range(6):
                    gate.wait(timeout=120)                                     # both threads enter a step together, every time
                    got[dt].append(step(models[dt]))
        except Exception as e:  # noqa: BLE001
            errs.append((dt, repr(e)))
            gate.abort()
    threads = [threading.Thread(target=worker, args=(dt,)) for dt in models]
    [t.start() for t in threads]
    [t.join(timeout=600) for t in threads]
    assert not errs, errs
    for dt in models:
        assert len(got[dt]) == 6
        for loss, grads in got[dt]:
            assert abs(loss - alone[dt][0]) <= 2e-6 * abs(alone[dt][0]), (dt, loss, alone[dt][0])
            for k, g in grads.items():
                ref = alone[dt][1][k]
                assert float((g - ref).abs().max()) <= 2e-6 * float(ref.abs().max()), (dt, k)
    other = {"bfloat16": "float32", "float32": "bfloat16"}
    for dt in models:                                                          # ... and a step run in the OTHER mode would have been seen
        k = "linear_1.weight"
        d = float((alone[dt][1][k] - alone[other[dt]][1][k]).abs().max()) / float(alone[dt][1][k].abs().max())
        assert d > 1e-4, d
    assert (models["bfloat16"].language_model._w.res_f32, models["float32"].language_model._w.res_f32) == (0, 1)


# ============================================================================ (d) ADVICE r4: the MoE auxiliary shadow, numerically
def _moe_model(seed=0):
    enc, lmc = OW.enc_config(256, 512, 1, 4), OW.lm_config(1024, 256, 512, 2, 4, 2, 128)
    cfg = ASRConfig(audio_config=enc, text_config=lmc, projector_type="moe", projector_hidden_dim=128, audio_token_id=1023,
                    pad_token_id=1000, eos_token_id=1001, router_jitter_noise=0.0, audio_token_dropout=0.0, router_aux_loss_coef=0.05)
    torch.manual_seed(seed)
    m = ASRModel(cfg, device=DEV, init="random", seed=seed)
    m.train()
    f = LogMelFeatureExtractor(128, DEV)([OW.synthetic_wave(0, 32000), OW.synthetic_wave(1, 32000)], sampling_rate=16000)
    ids, att, lab, counts = OW.synthetic_tokens(2, 25, 1024, 1023, 1000, 1001, n_text=10, n_suffix=4)
    T = torch.from_numpy
    batch = dict(input_ids=T(ids), input_features=f["input_features"], attention_mask=T(att), labels=T(lab), audio_token_counts=T(counts))
    return m, batch


def test_moe_aux_shadow_values():
    """ADVICE r4 (high + medium).  (1) ta_moe_router_aux_grads == the d(norm.weight) / d(router.weight) that backward(dy = 0,
    d_aux = a) produces.  (2) An optimizer step in the SHADOW form with one micro-batch (what N > 1 ranks run: gradients written
    straight into the flat buffer, aux share into the shadow, g += (N - 1) * shadow) leaves g_ce / N + g_aux in the buffer and the
    same weights as the one-rank aux * N form (itself pinned on the reference: train3_moe_small.npz).  (3) The same with two
    accumulated micro-batches: g_ce_total / N_total + the SUM of the micro-batches' aux gradients."""
    from tiny_audio_amd.trainer import ASRTrainer, TrainingArguments
    # ---- raw pieces from autograd on an untouched model: g_ce (sum-CE) and g_aux, separately
    m0, batch = _moe_model()
    out = m0(**batch, num_items_in_batch=1.0, return_logits=False)
    N = out.n_label_tokens
    names = [n for n, _ in m0.projector.named_parameters()]
    ps = list(m0.projector.parameters())
    g_ce = dict(zip(names, torch.autograd.grad(out.loss_ce, ps, retain_graph=True, allow_unused=True)))
    g_aux = dict(zip(names, torch.autograd.grad(out.aux_loss, ps, allow_unused=True)))
    assert all(g_aux[k] is None or float(g_aux[k].abs().max()) == 0.0 for k in names if k not in ("norm.weight", "router.weight"))
    assert float(g_aux["router.weight"].abs().max()) > 0 and float(g_aux["norm.weight"].abs().max()) > 0
    w0 = {k: p.detach().clone() for k, p in m0.projector.named_parameters()}

    def run(ga, shadow):
        m, b = _moe_model()
        tr = ASRTrainer(m, TrainingArguments(learning_rate=1e-3, gradient_accumulation_steps=ga), aux_shadow=shadow)
        assert tr._aux_direct == (not shadow and ga == 1) and (m.projector._grad_direct is not None) == (ga == 1)
        for _ in range(ga):
            tr.training_step(dict(b))
        torch.cuda.synchronize()
        f = tr.flat
        cnt = float(f.count_slot)
        g = {k: (f.grad_of("projector." + k) / cnt).clone() for k in ("norm.weight", "router.weight", "shared_expert.fc1.weight")}
        sh = {k: f.shadow("projector." + k).clone() for k in ("norm.weight", "router.weight")} if f.shadow_names else None
        return g, sh, {k: p.detach().clone() for k, p in m.projector.named_parameters()}, cnt

    g_a, sh_a, w_a, cnt_a = run(1, False)              # one rank: aux * N back-propagated, no shadow
    g_b, sh_b, w_b, cnt_b = run(1, True)               # the N > 1 form on one rank: direct gradients + shadow
    g_c, sh_c, w_c, cnt_c = run(2, True)               # two micro-batches (the same batch twice)
    assert sh_a is None and cnt_a == cnt_b == N and cnt_c == 2 * N
    for k in ("norm.weight", "router.weight"):
        # (1) the shadow IS g_aux
        assert cosine(npy(sh_b[k]), npy(g_aux[k])) > 0.9999 and abs(float(sh_b[k].norm() / g_aux[k].norm()) - 1) < 2e-2, k
        assert cosine(npy(sh_c[k]), npy(g_aux[k])) > 0.9999 and abs(float(sh_c[k].norm() / (2 * g_aux[k].norm())) - 1) < 2e-2, k
        # (2) g_ce / N + g_aux in both one-micro-batch forms
        want = g_ce[k] / N + g_aux[k]
        for got in (g_a[k], g_b[k]):
            assert cosine(npy(got), npy(want)) > 0.9995 and abs(float(got.norm() / want.norm()) - 1) < 2e-2, k
        # (3) two micro-batches: 2 g_ce / 2N + 2 g_aux
        want2 = g_ce[k] / N + 2 * g_aux[k]
        assert cosine(npy(g_c[k]), npy(want2)) > 0.9995 and abs(float(g_c[k].norm() / want2.norm()) - 1) < 2e-2, k
        # aux at weight 1 / N (the bug) would be far away: the two shares are comparable in size here
        wrong = g_ce[k] / N + g_aux[k] / N
        assert float((g_b[k] - want).norm()) < 0.2 * float((wrong - want).norm()), k
    k = "shared_expert.fc1.weight"                      # a tensor aux does not reach: plain g_ce / N everywhere
    for got in (g_a[k], g_b[k], g_c[k]):
        assert cosine(npy(got), npy(g_ce[k] / N)) > 0.9995
    # the applied update: both one-micro-batch forms moved every weight the same way
    for k in w0:
        da, db = npy(w_a[k] - w0[k]), npy(w_b[k] - w0[k])
        if np.abs(da).max() > 0:
            assert cosine(da, db) > 0.999, k


def test_moe_router_aux_grads_kernel_vs_backward():
    """ta_moe_router_aux_grads (null dtopw / null dxn_sh kernel paths) against the full backward with dy = 0, d_aux = a."""
    from tiny_audio_amd import torch_ops
    m, batch = _moe_model(seed=1)
    hidden = m.audio_tower(batch["input_features"]).last_hidden_state
    y = m.projector(hidden)
    aux = m.projector.get_aux_loss()
    a = 3.0
    ps = [m.projector.norm.weight, m.projector.router.weight]
    gn, gr = torch.autograd.grad(a * aux + 0.0 * y.sum(), ps)
    # the op's saved state: run the aux-only kernel on the tape of a fresh forward
    xb = hidden.detach()
    y2, aux2, xb_out, tape = torch.ops.ta355.moe_projector(xb, None, list(m.projector._param_list()), torch_ops.register_module(m.projector), True)
    sn, sr = torch_ops.moe_router_aux_grads(torch.tensor(a, device=DEV), xb if xb_out.numel() == 0 else xb_out, None, tape,
                                            torch_ops.register_module(m.projector), True)
    for got, want, k in ((sn, gn, "norm.weight"), (sr, gr, "router.weight")):
        assert cosine(npy(got), npy(want)) > 0.99999 and abs(float(got.norm() / want.norm()) - 1) < 1e-3, k


# ============================================================================ (e) data parallel with the REAL kernels on one GPU
def _dp_one_gpu_worker(rank, world, port, q):
    """two ranks SHARING cuda:0 under the gloo backend (RCCL refuses two ranks on one device; gloo moves CUDA tensors through the host):
    everything but the transport is the N > 1 product path -- HIP kernels, flat buffer, one collective, deferred update."""
    import socket  # noqa: F401
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from tiny_audio_amd.trainer import ASRTrainer, TrainingArguments
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = R.SMALL
    cfg = ASRConfig(audio_config=S["enc"], text_config=S["lm"], projector_hidden_dim=S["proj_hidden"],
                    audio_token_id=S["audio_token_id"], audio_token_dropout=0.0)
    torch.manual_seed(0)
    m = ASRModel(cfg, device=dev, init="random", seed=0)
    m.train()
    res = {}
    for overlap in (False, True):
        m.load_state_dict({"projector." + k: torch.from_numpy(v) for k, v in
                           OW.init_mlp_projector(S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"]).items()})
        tr = ASRTrainer(m, TrainingArguments(learning_rate=1e-3), overlap_allreduce=overlap)
        batch = _dp_batch(rank)
        for _ in range(2):
            tr.training_step(batch)
        tr.flush()
        torch.cuda.synchronize()
        res[overlap] = (tr.global_step, float(tr._last[1]), tr.last_loss(), npy(m.projector.linear_2.weight).copy(),
                        npy(m.projector.linear_1.weight).copy())
    q.put((rank, res))
    dist.destroy_process_group()


def _dp_batch(rank):
    """rank's two clips (rank None: all four, as one process would see the global batch); the ranks hold different label counts"""
    S = R.SMALL
    parts = []
    for r in ((0, 1) if rank is None else (rank,)):
        ids, att, lab, counts = OW.synthetic_tokens(2, [12, 12], S["lm"]["vocab"], S["audio_token_id"], S["pad_id"], S["eos_id"],
                                                    n_text=20 if r == 0 else 11, n_suffix=4, L=40)
        x = (0.5 * np.random.RandomState(40 + r).standard_normal((2, 128, 100))).astype(np.float32)
        parts.append((ids, att, lab, counts, x))
    cat = [np.concatenate([p[i] for p in parts]) for i in range(5)]
    T = torch.from_numpy
    return dict(input_ids=T(cat[0]), attention_mask=T(cat[1]), labels=T(cat[2]), audio_token_counts=T(cat[3]), input_features=T(cat[4]))


def test_trainer_two_ranks_real_kernels_on_one_gpu():
    """BASELINE configs[2] in miniature with the REAL kernels (VERDICT r04 "missing" 2: the N > 1 path had only ever run under gloo with
    stubbed kernels, or not at all): two ranks share this box's one GPU under gloo -- different clips and label counts per rank, one
    flat [grads | count | loss] all-reduce per step, synchronous and deferred-update modes.  Replicas end bit-identical, both modes
    agree, and the result is the ONE-process step over the global batch of four clips (sum-CE / global token count)."""
    import socket
    import torch.multiprocessing as mp
    from tiny_audio_amd.trainer import ASRTrainer, TrainingArguments
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_one_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    out = dict(q.get(timeout=600) for _ in procs)
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for overlap in (False, True):
        (s0, c0, l0, w0, v0), (s1, c1, l1, w1, v1) = out[0][overlap], out[1][overlap]
        assert s0 == s1 == 2 and c0 == c1 == 2 * 21 + 2 * 12                      # label tokens of both ranks
        assert l0 == l1 and np.array_equal(w0, w1) and np.array_equal(v0, v1)     # replicas stay bit-identical
    assert np.allclose(out[0][False][3], out[0][True][3], atol=1e-6)              # deferred update == immediate update
    # ---- one process, the global batch
    S = R.SMALL
    cfg = ASRConfig(audio_config=S["enc"], text_config=S["lm"], projector_hidden_dim=S["proj_hidden"],
                    audio_token_id=S["audio_token_id"], audio_token_dropout=0.0)
    torch.manual_seed(0)
    m = ASRModel(cfg, device=DEV, init="random", seed=0)
    m.train()
    w_init = OW.init_mlp_projector(S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"])
    m.load_state_dict({"projector." + k: torch.from_numpy(v) for k, v in w_init.items()})
    tr = ASRTrainer(m, TrainingArguments(learning_rate=1e-3))
    for _ in range(2):
        tr.training_step(_dp_batch(None))
    torch.cuda.synchronize()
    assert float(tr._last[1]) == 2 * 21 + 2 * 12
    assert abs(tr.last_loss() - out[0][False][2]) < 2e-3 * abs(tr.last_loss())
    for got, name in ((out[0][False][3], "linear_2.weight"), (out[0][False][4], "linear_1.weight")):
        one = npy(getattr(m.projector, name.split(".")[0]).weight)
        d_dp, d_one = got - w_init[name], one - w_init[name]
        assert np.abs(d_one).max() > 0 and cosine(d_dp, d_one) > 0.999, name       # the same update (other tilings: bf16-level differences)


def test_bench_two_ranks_real_kernels_on_one_gpu():
    """`python bench.py --gpus 2 --dist-backend gloo --share-gpu`: the driver's N > 1 invocation with two ranks on this box's one GPU --
    full-depth model, real kernels, the flat all-reduce in both modes, the replica check.  Only the transport is not RCCL.  The line must
    say the replicas are identical (they were NOT before round 5: ta_grad_sqnorm's float atomics gave every rank its own last bits of
    the clip coefficient) and the process must exit 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--share-gpu", "--steps", "4",
                        "--warmup", "2", "--no-cpu-baseline", "--no-logits-full", "--no-roofline"], capture_output=True, text=True, timeout=600,
                       cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-800:], r.stderr[-800:])
    d = json.loads(lines[-1])
    assert d.get("error") is None and d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["config"]["global_batch"] == 64
    rep = d["replicas"]
    assert rep["replicas_identical"] is True and rep["global_step"]["min"] == rep["global_step"]["max"] > 0
    assert rep["weight_checksum"]["sum_min"] == rep["weight_checksum"]["sum_max"]
    ar = d["allreduce"]
    assert ar["bytes"] == 4 * (6293504 + 2) and ar["other_mode"]["mode"] == "synchronous"
    assert d["validation_only"] and 11.0 < d["final_loss"] < 13.0
    assert ar["collectives_per_step"] == 1.0


@pytest.mark.parametrize("flags,elements", [(["--projector", "moe"], None), (["--lora"], 5046272 + 2),
                                             (["--full-ft", "--batch", "8"], None)], ids=["moe", "lora", "fullft"])
def test_bench_two_ranks_other_configs_on_one_gpu(flags, elements):
    """the N > 1 product path of configs[3] (MoE: the auxiliary-loss shadow segment rides in the same buffer),
    configs[4] (LoRA stage 2: adapter gradients only) and the full-decoder fine-tune (2.4 GB of gradients), two ranks on this box's
    one GPU over gloo with the real kernels: replicas identical after the steps, ONE collective per optimizer step, exit code 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--share-gpu", "--steps", "3",
                        "--warmup", "2", "--no-cpu-baseline", "--no-logits-full", "--no-roofline", *flags], capture_output=True, text=True,
                       timeout=900, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-800:], r.stderr[-800:])
    d = json.loads(lines[-1])
    assert d.get("error") is None and d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["validation_only"]
    rep, ar = d["replicas"], d["allreduce"]
    assert rep["replicas_identical"] is True and rep["global_step"]["min"] == rep["global_step"]["max"] > 0
    assert rep["weight_checksum"]["sum_min"] == rep["weight_checksum"]["sum_max"]
    assert ar["collectives_per_step"] == 1.0 and ar["other_mode"]["mode"] == "synchronous"
    if elements is not None:
        assert ar["elements"] == elements
    assert ar["bytes"] == 4 * ar["elements"] and np.isfinite(d["final_loss"])
