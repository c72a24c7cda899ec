"""-m gpu, round 2: the rows VERDICT r01 marked partial / weak.

  * configs[3] / configs[4] at FULL depth (32 + 28 layers, L = 192): size-independent properties of the MoE projector
    and of LoRA stage 2 (tiny_audio/projectors.py:281-347; tiny_audio/asr_modeling.py:289-301);
  * 30 s clips: T = 3000 mel frames -> S = 1500 = the encoder's max_position_embeddings (scripts/train.py:269-272);
  * the MoE auxiliary loss through the real trainer (HF semantics: not token-normalised), pinned on a reference run;
  * the device collator end to end (scripts/train.py:240-348) and a reference-written checkpoint loaded on the GPU;
  * an exact-match greedy case (sharpened lm_head: no near ties);
  * RCCL: the real ASRTrainer on the nccl backend (skipped below 2 visible GPUs).
"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import encoder as OE
from oracle import features as OF
from oracle import weights as OW
from tests.golden import recipe as R

if torch.cuda.is_available():
    from tiny_audio_amd.asr_config import ASRConfig, EncoderConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    from tiny_audio_amd.encoder import GlmAsrEncoderMI355X
    from tiny_audio_amd.trainer import ASRTrainer, TrainingArguments

DEV = "cuda"



def _force_variant(monkeypatch, variant):
    """TA355_GEMM_VARIANT for the launches that follow (0-5, 10, 12: every tile variant the library carries; the experiment-only
    variants 6-9 / 11 of rounds 2-5 left it in round 6)."""
    monkeypatch.setenv("TA355_GEMM_VARIANT", variant)


def cosine(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    if not a.any() and not b.any():
        return 1.0                                              # two all-zero gradients (an expert without tokens) agree
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))


def npy(t):
    return t.detach().float().cpu().numpy()


def _full_batch(cfg, B=3, L=192, n_audio=125):
    V = cfg.text_config.vocab_size
    ids, att, lab, counts = OW.synthetic_tokens(B, n_audio, V, cfg.audio_token_id, cfg.pad_token_id, cfg.eos_token_id, L=L)
    return dict(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(att), labels=torch.from_numpy(lab),
                audio_token_counts=torch.from_numpy(counts))


# ============================================================================ configs[3] at full depth
def test_full_size_properties_moe():
    """Sparse MoE projector (4 routed experts, top-2, shared expert) behind the full-depth frozen models."""
    torch.manual_seed(0)
    cfg = ASRConfig(projector_type="moe", audio_token_dropout=0.0, router_jitter_noise=0.0)
    m = ASRModel(cfg, device=DEV, init="random", seed=0)
    V = cfg.text_config.vocab_size
    B = 3
    feats = torch.randn(B, 128, 1000) * 0.5
    tb = _full_batch(cfg, B)
    m.train()
    out = m(input_features=feats, **tb, return_logits=False)
    assert out.n_label_tokens == 108 and np.isfinite(float(out.loss)) and abs(float(out.loss_ce) - np.log(V)) < 3.0
    aux = float(out.aux_loss)
    assert np.isfinite(aux) and aux > 0.0 and abs(float(out.loss) - float(out.loss_ce) - aux) < 1e-4
    out.loss.backward()
    g1 = {k: p.grad.clone() for k, p in m.projector.named_parameters()}
    assert all(torch.isfinite(g).all() for g in g1.values())
    # top-2 routing: at least two experts receive tokens (with random 32-layer weights the encoder rows are strongly
    # correlated, so the router may well send every token to the same pair); an expert without tokens has no gradient at all
    routed = [e for e in range(4) if float(g1[f"experts.{e}.fc1.weight"].abs().max()) > 0]
    assert len(routed) >= 2
    for e in set(range(4)) - set(routed):
        assert all(float(g1[f"experts.{e}.{n}"].abs().max()) == 0.0 for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"))
    nll1 = out.nll.clone()
    # permuting the clips permutes the per-token NLLs, leaves loss, aux and the gradients unchanged (routing is per token)
    perm = [2, 0, 1]
    m.zero_grad()
    out2 = m(input_features=feats[perm], **{k: v[perm] for k, v in tb.items()}, return_logits=False)
    out2.loss.backward()
    assert torch.allclose(torch.cat([nll1[72:108], nll1[0:36], nll1[36:72]]), out2.nll, rtol=2e-3, atol=2e-3)
    assert abs(float(out2.aux_loss) - aux) < 1e-3 * aux + 1e-7
    for k, p in m.projector.named_parameters():
        assert cosine(npy(p.grad), npy(g1[k])) > 0.999, k
    # Every frame dropped (the injected mask of _maybe_drop_audio_tokens): all projector inputs are zero rows, the
    # router logits tie at 0 and top-2 takes experts 0 and 1 for every token -> experts 2 and 3 are unused and their
    # gradients must be EXACTLY zero (the flat all-reduce buffer relies on zero-filled unused experts)
    m.zero_grad()
    keep = torch.zeros(B * 500, device=DEV)
    out3 = m(input_features=feats, **tb, return_logits=False, frame_keep=keep)
    out3.loss.backward()
    P = dict(m.projector.named_parameters())
    for e in (2, 3):
        for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
            assert float(P[f"experts.{e}.{n}"].grad.abs().max()) == 0.0, (e, n)
    for e in (0, 1):
        assert float(P[f"experts.{e}.fc2.bias"].grad.abs().max()) > 0.0
    assert float(P["shared_expert.fc2.bias"].grad.abs().max()) > 0.0


# ============================================================================ configs[4] at full depth
def test_full_size_properties_lora():
    """Stage 2: frozen projector, rank-8 adapters on all 196 linears of the 28-layer LM."""
    torch.manual_seed(0)
    cfg = ASRConfig(use_lora=True, freeze_projector=True, audio_token_dropout=0.0)
    m = ASRModel(cfg, device=DEV, init="random", seed=0)
    V = cfg.text_config.vocab_size
    B = 3
    feats = torch.randn(B, 128, 1000) * 0.5
    tb = _full_batch(cfg, B)
    m.train()
    train = {k: p for k, p in m.named_parameters() if p.requires_grad}
    assert sum(p.numel() for p in train.values()) == 5_046_272                      # peft's count for Qwen3-0.6B, r = 8
    assert all(k.startswith("language_model.") for k in train) and not any(p.requires_grad for p in m.projector.parameters())
    # lora_B = 0 at initialisation: the adapted model IS the base model
    out = m(input_features=feats, **tb, return_logits=False)
    base = ASRModel(ASRConfig(audio_token_dropout=0.0), device=DEV, init="random", seed=0)
    base.load_state_dict({"projector." + k: v for k, v in m.projector.state_dict().items()})
    base.train()
    ref = base(input_features=feats, **tb, return_logits=False)
    assert abs(float(out.loss) - float(ref.loss)) < 2e-3 * float(ref.loss) and abs(float(out.loss) - np.log(V)) < 3.0
    del base, ref
    out.loss.backward()
    for k, p in train.items():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    assert all(p.grad is None for p in m.projector.parameters())
    g1 = {k: p.grad.clone() for k, p in train.items()}
    # with B = 0 only lora_B receives a gradient (d lora_A = s B^T ... = 0)
    assert any("lb_" in k and float(g.abs().max()) > 0 for k, g in g1.items())
    assert all(float(g.abs().max()) == 0.0 for k, g in g1.items() if "la_" in k)
    # move B off zero so that both factors carry gradient, then: permutation invariance + linearity in 1 / num_items
    with torch.no_grad():
        gen = torch.Generator(device=DEV); gen.manual_seed(5)
        for k, p in train.items():
            if "lb_" in k:
                p.copy_(torch.randn(p.shape, device=DEV, generator=gen) * 0.02)
    m.zero_grad()
    o1 = m(input_features=feats, **tb, return_logits=False)
    o1.loss.backward()
    g2 = {k: p.grad.clone() for k, p in train.items()}
    assert abs(float(o1.loss) - float(out.loss)) > 1e-4                             # the adapters now change the function
    assert all(float(g.abs().max()) > 0 for g in g2.values())
    perm = [1, 2, 0]
    m.zero_grad()
    o2 = m(input_features=feats[perm], **{k: v[perm] for k, v in tb.items()}, return_logits=False)
    o2.loss.backward()
    assert abs(float(o2.loss) - float(o1.loss)) < 1e-3 * float(o1.loss)
    for k, p in train.items():
        assert cosine(npy(p.grad), npy(g2[k])) > 0.998, k
    m.zero_grad()
    o3 = m(input_features=feats, **tb, return_logits=False, num_items_in_batch=54)   # half the 108 label tokens
    o3.loss.backward()
    for k, p in train.items():
        assert cosine(npy(p.grad), npy(g2[k])) > 0.9999 and abs(float(p.grad.norm() / g2[k].norm()) - 2.0) < 5e-3, k


# ============================================================================ 30 s clips: S = 1500 = max_position_embeddings
def test_encoder_30s_true_width_vs_oracle():
    """T = 3000 mel frames (scripts/train.py:269-272 admits clips up to 30 s) -> 1500 encoder frames: the last rotary
    position, 24 query tiles of 64 and a 28-key tail in the attention kernel, 1500-row M tails in every GEMM."""
    enc_cfg = OW.enc_config(layers=2)
    w = OW.init_encoder(enc_cfg, 0)
    enc = GlmAsrEncoderMI355X(EncoderConfig(enc_cfg), DEV).load_state_dict_hf(w)
    x = (0.6 * np.random.RandomState(13).standard_normal((1, 128, 3000))).astype(np.float32)
    ref = OE.encoder_forward(x, w, enc_cfg)
    out = enc(torch.from_numpy(x), return_f32=True).last_hidden_state
    assert out.shape == (1, 1500, 1280)
    d = np.abs(npy(out) - ref)
    assert d.max() / np.abs(ref).max() < 2e-2 and cosine(npy(out), ref) > 0.9995
    # batch of two: 30 s next to 10 s of content zero-padded to 30 s (no mask inside the encoder, as the reference)
    x2 = np.concatenate([x, np.zeros_like(x)], 0); x2[1, :, :1000] = x[0, :, :1000]
    out2 = enc(torch.from_numpy(x2), return_f32=True).last_hidden_state
    assert cosine(npy(out2[0]), ref[0]) > 0.9995
    ref2 = OE.encoder_forward(x2[1:], w, enc_cfg)
    assert cosine(npy(out2[1]), ref2[0]) > 0.9995
    with pytest.raises(Exception):
        enc(torch.from_numpy(np.zeros((1, 128, 3002), np.float32)))                  # S = 1501 > max_position_embeddings


# ============================================================================ MoE auxiliary loss through the trainer
def test_three_training_steps_moe_vs_golden(golden):
    """HF Trainer semantics (TF:trainer.py compute_loss / training_step): loss = sum(nll) / num_items + aux, the MoE
    balance / z loss at FULL weight.  Fixture: 3 AdamW + clip steps of the reference model with a 5x aux coefficient
    (DDP-style zero gradients for unused experts, see make_golden.gen_train_moe)."""
    g, g3 = golden("asr_small.npz"), golden("train3_moe_small.npz")
    S = R.SMALL
    E, D, H = S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"]
    cfg = ASRConfig(audio_config=S["enc"], text_config=S["lm"], projector_hidden_dim=H, projector_type="moe",
                    audio_token_id=S["audio_token_id"], router_jitter_noise=0.0, router_aux_loss_coef=0.05, audio_token_dropout=0.0)
    m = ASRModel(cfg, device=DEV, init="none")
    m.audio_tower.load_state_dict_hf(OW.init_encoder(S["enc"], 0))
    m.language_model.load_state_dict_hf(OW.init_lm(S["lm"], 1))
    m.load_state_dict({"projector." + k: torch.from_numpy(v) for k, v in OW.init_moe_projector(E, D, H).items()})
    ids, att, lab, counts = R.asr_tokens(g["counts"])
    batch = dict(input_ids=torch.from_numpy(ids), input_features=torch.from_numpy(g["input_features"]),
                 attention_mask=torch.from_numpy(att), labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts))
    m.train()
    tr = ASRTrainer(m, TrainingArguments(learning_rate=1e-3, weight_decay=0.0, max_grad_norm=1.0))
    losses, gnorms, auxes = [], [], []
    for _ in range(3):
        tr.training_step(batch)
        losses.append(tr.last_loss() + tr.last_aux()); gnorms.append(tr.last_grad_norm()); auxes.append(tr.last_aux())
    # the auxiliary loss is a small difference of routing statistics: a handful of bf16 near-tie routing decisions move it
    np.testing.assert_allclose(auxes, g3["aux"], rtol=0.15)
    np.testing.assert_allclose(losses, g3["losses"], rtol=5e-3)
    np.testing.assert_allclose(gnorms, g3["gnorms"], rtol=3e-2)
    P = dict(m.projector.named_parameters())
    for k in [k[2:] for k in g3.files if k.startswith("w.")]:
        d = np.abs(npy(P[k]) - g3["w." + k])
        assert d.mean() < 2e-4, (k, d.mean())                 # each Adam step moves a weight by <= lr = 1e-3
    # the router is where a mis-weighted aux shows: with aux / N (the round-1 bug) its 3-step travel differs by > 30 %
    w0 = OW.init_moe_projector(E, D, H)["router.weight"]
    assert cosine(npy(P["router.weight"]) - w0, g3["w.router.weight"] - w0) > 0.98


# ============================================================================ device collator end to end (8(f) rank 2)
def test_device_collator_end_to_end(golden):
    """DataCollator (scripts/train.py:240-348) with the log-mel on the GPU: raw waveforms -> ta_logmel_f32 -> frame mask
    -> <audio> counts -> chat rows with masked prompt; features against the fixture written by the reference's
    WhisperFeatureExtractor, counts against the reference's formulas, and the batch drives one training step."""
    from tests.test_host_logic import _StubChatTokenizer
    from tiny_audio_amd.asr_processing import LogMelFeatureExtractor
    from tiny_audio_amd.collator import DataCollator
    from tiny_audio_amd.projectors import MLPAudioProjector
    g = golden("logmel.npz")
    tok, fe = _StubChatTokenizer(), LogMelFeatureExtractor(128, DEV)
    col = DataCollator(tok, fe, 16000, system_prompt="You are a helpful assistant.", projector=MLPAudioProjector(ASRConfig()))
    waves = R.logmel_waves()
    texts = ["Hello World <comma> this is a TEST", "second [laughs] clip 50%", "third one"]
    batch = col([{"audio": {"array": w, "sampling_rate": 16000}, "text": t} for w, t in zip(waves, texts)])
    assert batch["input_features"].is_cuda and batch["input_features"].shape == g["feats"].shape
    d = np.abs(npy(batch["input_features"]) - g["feats"])
    assert d.max() < 5e-4 and d.mean() < 2e-5
    np.testing.assert_array_equal(batch["audio_attention_mask"].cpu().numpy(), g["mask"])
    want = OF.audio_token_counts(g["mask"])
    assert batch["audio_token_counts"].tolist() == want.tolist()
    ids, lab = batch["input_ids"], batch["labels"]
    for i, t in enumerate(["hello world this is a test", "second clip 50 percent", "third one"]):
        row, lr = ids[i].tolist(), lab[i].tolist()
        assert row.count(3) == int(want[i]) and all(lr[j] == -100 for j, x in enumerate(row) if x == 3)
        assert tok.decode([x for x, l in zip(row, lr) if l != -100]) == t
    # the batch is a valid training batch: one optimizer step of a small model on it
    S = R.SMALL
    cfg = ASRConfig(audio_config=S["enc"], text_config=dict(S["lm"], vocab=4096), projector_hidden_dim=S["proj_hidden"],
                    audio_token_id=3, audio_token_dropout=0.0)
    m = ASRModel(cfg, device=DEV, init="random", seed=0)
    m.train()
    tr = ASRTrainer(m, TrainingArguments(learning_rate=1e-3))
    w0 = m.projector.linear_1.weight.detach().clone()
    tr.training_step({k: v for k, v in batch.items()})
    assert np.isfinite(tr.last_loss()) and tr.global_step == 1 and not torch.equal(w0, m.projector.linear_1.weight)


# ============================================================================ reference-written checkpoint on the GPU (8(f) rank 3)
def test_reference_checkpoint_loads_on_gpu_and_matches_the_reference_run(golden):
    """tests/golden/ckpt_small was written by the REFERENCE model (its state_dict() through safetensors + its
    config.json); asr_small.npz holds the same model's loss / logits / projector gradients.  from_pretrained on the GPU +
    the frozen models' HF state dicts must reproduce them."""
    g = golden("asr_small.npz")
    S = R.SMALL
    ref_dir = os.path.join(os.path.dirname(__file__), "golden", "ckpt_small")
    class Tok:                              # the reference takes <audio>'s id from the tokenizer at load time, not from config.json
        def convert_tokens_to_ids(self, t):
            return {"<audio>": S["audio_token_id"]}.get(t)
    m = ASRModel.from_pretrained(ref_dir, device=DEV, init="none", encoder_state_dict=OW.init_encoder(S["enc"], 0),
                                 lm_state_dict=OW.init_lm(S["lm"], 1), tokenizer=Tok())
    assert m.audio_token_id == S["audio_token_id"]
    assert m.config.projector_type == "mlp" and next(m.projector.parameters()).is_cuda
    m.config.audio_token_dropout = 0.0
    ids, att, lab, counts = R.asr_tokens(g["counts"])
    m.train()
    out = m(input_ids=torch.from_numpy(ids), input_features=torch.from_numpy(g["input_features"]),
            attention_mask=torch.from_numpy(att), labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts))
    out.loss.backward()
    ref_loss = float(g["mlp.loss"])
    assert abs(float(out.loss) - ref_loss) < 5e-3 * ref_loss
    valid = att.astype(bool)
    dl = npy(out.logits)[valid].astype(np.float64) - g["mlp.logits"][valid]
    assert np.abs(dl).max() < 0.1 and np.sqrt((dl ** 2).mean()) < 0.02
    for k, prm in m.projector.named_parameters():
        assert cosine(npy(prm.grad), g["mlp.g." + k]) > 0.999, k


# ============================================================================ greedy decoding without near ties
def test_generate_exact_match_with_decisive_margins(golden):
    """Greedy parity stated exactly.  A random LM's logits are nearly flat (top-1 minus top-2 of 0.05-0.3 against a bf16
    error of ~0.1: the reason the other generate tests allow near ties), and scaling the head does not help -- margin
    and error scale together.  Here the tied embedding is scaled 30x against the blocks' outputs, so the residual stream
    is dominated by the last input token and the oracle's decisions have margins of 10-40 against a (scaled) bf16 error
    of ~3.6: every decision whose oracle margin exceeds twice that must be reproduced EXACTLY, per clip (the two clips
    end in different prompt tokens and therefore decode different tokens)."""
    from oracle import generate as OG
    S = R.SMALL
    E, D, H = S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"]
    g = golden("generate_small.npz")
    SC = 30.0
    wE, wL, wP = OW.init_encoder(S["enc"], 0), dict(R.gen_lm_weights()), OW.init_mlp_projector(E, D, H)
    wL["model.embed_tokens.weight"] = wL["model.embed_tokens.weight"] * np.float32(SC)
    cfgm = ASRConfig(audio_config=S["enc"], text_config=S["lm"], projector_hidden_dim=H, audio_token_id=S["audio_token_id"],
                     pad_token_id=S["pad_id"], eos_token_id=S["eos_id"])
    m = ASRModel(cfgm, device=DEV, init="none")
    m.audio_tower.load_state_dict_hf(wE); m.language_model.load_state_dict_hf(wL)
    m.load_state_dict({"projector." + k: torch.from_numpy(v) for k, v in wP.items()})
    ids = g["input_ids"].copy()
    ids[0, -1], ids[1, -1] = 100, 700                                             # the clips end in different tokens (oracle margins 9-65)
    kw = dict(input_ids=torch.from_numpy(ids), input_features=torch.from_numpy(g["input_features"]),
              audio_attention_mask=torch.from_numpy(g["audio_attention_mask"]),
              attention_mask=torch.ones(ids.shape, dtype=torch.int64))
    out = m.generate(**kw, max_new_tokens=5).cpu().numpy()
    W = dict(encoder=wE, lm=wL, projector=wP)
    cfg = dict(enc=S["enc"], lm=S["lm"], projector_type="mlp", k=S["k"], audio_token_id=S["audio_token_id"])
    ref, margins = OG.greedy_generate(dict(input_ids=ids, input_features=g["input_features"]), W, cfg, max_new_tokens=5,
                                      eos_ids=(S["eos_id"], S["pad_id"]), pad_id=S["pad_id"], return_margins=True)
    n = min(out.shape[1], ref.shape[1])
    decided = margins[:, :n] > 2 * 0.12 * SC
    compared = 0
    for b in range(out.shape[0]):                                                  # a row is comparable up to its first near tie
        stop = int(np.argmin(decided[b])) if not decided[b].all() else n
        assert (out[b, :stop] == ref[b, :stop]).all(), (b, out[b, :n], ref[b, :n], margins[b, :n])
        compared += stop
    assert compared == 10 and ref[0, 0] != ref[1, 0], (compared, ref, margins)


# ============================================================================ RCCL: the real trainer on the nccl backend
def _nccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", device_id=dev)
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
        tr = ASRTrainer(m, TrainingArguments(learning_rate=1e-3), overlap_allreduce=overlap, time_allreduce=True)
        n_text = 20 if rank == 0 else 11                                           # ranks hold different label counts
        ids, att, lab, counts = OW.synthetic_tokens(2, [12, 12], S["lm"]["vocab"], S["audio_token_id"], S["pad_id"], S["eos_id"],
                                                    n_text=n_text, n_suffix=4)
        x = (0.5 * np.random.RandomState(40 + rank).standard_normal((2, 128, 100))).astype(np.float32)
        batch = dict(input_ids=torch.from_numpy(ids), input_features=torch.from_numpy(x), attention_mask=torch.from_numpy(att),
                     labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts))
        for _ in range(2):
            tr.training_step(batch)
        tr.flush()
        torch.cuda.synchronize()
        res[overlap] = (tr.global_step, float(tr._last[1]), tr.last_loss(), npy(m.projector.linear_2.weight).copy(),
                        tr.allreduce_exposed_ms())
    q.put((rank, res))
    dist.destroy_process_group()


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL)")
def test_trainer_two_ranks_rccl():
    """configs[2] in miniature: two ranks, different clips and label counts, ONE flat all-reduce per step over RCCL;
    both ranks end with identical weights, the global token count, and the deferred-update mode gives the same result."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    out = dict(q.get(timeout=600) for _ in procs)
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for overlap in (False, True):
        (s0, c0, l0, w0, t0), (s1, c1, l1, w1, t1) = out[0][overlap], out[1][overlap]
        assert s0 == s1 == 2 and c0 == c1 == 2 * 21 + 2 * 12                      # label tokens of both ranks
        assert l0 == l1 and np.array_equal(w0, w1)                                # replicas stay bit-identical
        assert t0 >= 0.0 and t1 >= 0.0
    assert np.allclose(out[0][False][3], out[0][True][3], atol=1e-6)              # deferred update == immediate update


# ============================================================================ torch.library boundary + HF Trainer drop-in
def test_custom_ops_pass_opcheck():
    """torch.library.opcheck: schema (no undeclared mutation / aliasing), fake-tensor kernels agree with the real output
    metadata, and the registered autograd formulas are wired the way dispatcher-level autograd expects."""
    from tiny_audio_amd import torch_ops
    from tiny_audio_amd.asr_processing import LogMelFeatureExtractor
    from tiny_audio_amd.projectors import MLPAudioProjector
    S = R.SMALL
    cfg = ASRConfig(audio_config=S["enc"], text_config=S["lm"], projector_hidden_dim=S["proj_hidden"], audio_token_id=S["audio_token_id"])
    proj = MLPAudioProjector(cfg).to(DEV)
    x = torch.randn(2, 50, S["enc"]["hidden"], device=DEV).to(torch.bfloat16)
    args = (x, proj.linear_1.weight, proj.norm.weight, proj.linear_2.weight, proj.norm_2.weight, torch_ops.register_module(proj))
    torch.library.opcheck(torch.ops.ta355.mlp_projector, args, test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    fe = LogMelFeatureExtractor(128, DEV)
    wav = 0.1 * torch.randn(2, 16000, device=DEV)
    lens = torch.tensor([16000, 12000], device=DEV)
    torch.library.opcheck(torch.ops.ta355.logmel, (wav, lens, torch_ops.register_module(fe)), test_utils=("test_schema", "test_faketensor"))
    enc = GlmAsrEncoderMI355X(EncoderConfig(S["enc"]), DEV).load_state_dict_hf(OW.init_encoder(S["enc"], 0))
    feats = torch.randn(2, 128, 100, device=DEV)
    torch.library.opcheck(torch.ops.ta355.encoder_forward, (feats, None, torch_ops.register_module(enc), False),
                          test_utils=("test_schema", "test_faketensor"))


def test_hf_trainer_drives_the_model_unchanged(tmp_path):
    """scripts/train.py:630-643 hands the model and the collator to transformers.Trainer.  The same stock Trainer (its own
    AdamW, its own compute_loss with num_items_in_batch, its own training_step / backward) must drive ASRModel on MI355X
    unchanged: two optimizer steps move the projector, leave no gradient on frozen parts, and the loss it logs is finite."""
    transformers = pytest.importorskip("transformers")
    from tests.test_host_logic import _StubChatTokenizer
    from tiny_audio_amd.asr_processing import LogMelFeatureExtractor
    from tiny_audio_amd.collator import DataCollator
    S = R.SMALL
    cfg = ASRConfig(audio_config=S["enc"], text_config=dict(S["lm"], vocab=4096), projector_hidden_dim=S["proj_hidden"],
                    audio_token_id=3, audio_token_dropout=0.0)
    model = ASRModel(cfg, device=DEV, init="random", seed=0)
    tok, fe = _StubChatTokenizer(), LogMelFeatureExtractor(128, DEV)
    collator = DataCollator(tok, fe, 16000, system_prompt="You are a helpful assistant.", projector=model.projector)
    rng = np.random.RandomState(0)
    rows = [{"audio": {"array": (0.1 * rng.standard_normal(16000 + 800 * i)).astype(np.float32), "sampling_rate": 16000},
             "text": f"sample number {i} says hello world"} for i in range(8)]

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(rows)

        def __getitem__(self, i):
            r = rows[i]
            return {"audio": {"array": r["audio"]["array"].copy(), "sampling_rate": 16000}, "text": r["text"]}

    targs = transformers.TrainingArguments(output_dir=str(tmp_path), per_device_train_batch_size=2, max_steps=2, learning_rate=1e-3,
                                           weight_decay=0.0, max_grad_norm=1.0, logging_steps=1, save_strategy="no", report_to=[],
                                           remove_unused_columns=False, dataloader_num_workers=0, bf16=False, fp16=False,
                                           gradient_accumulation_steps=2, dataloader_pin_memory=False)
    w0 = {k: v.detach().clone() for k, v in model.projector.named_parameters()}
    trainer = transformers.Trainer(model=model, args=targs, train_dataset=DS(), data_collator=collator)
    out = trainer.train()
    assert out.global_step == 2 and np.isfinite(out.training_loss) and out.training_loss > 1.0
    for k, p in model.projector.named_parameters():
        assert not torch.equal(p.detach(), w0[k]), k
        assert torch.isfinite(p).all()
    logged = [h["loss"] for h in trainer.state.log_history if "loss" in h]
    assert len(logged) == 2 and all(np.isfinite(v) for v in logged)
    # the updated masters reach the kernels: the packed bf16 images are rebuilt from the new Parameter versions
    model.eval()
    with torch.no_grad():
        b = collator([{"audio": {"array": rows[0]["audio"]["array"].copy(), "sampling_rate": 16000}, "text": rows[0]["text"]}])
        o = model(**{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()})
    assert np.isfinite(float(o.loss))
    # Trainer.predict (prediction_loss_only = False): prediction_step concatenates every item of the output but "loss" over the
    # batches -- the output must hold tensors only (None fields / an int among the items raised TypeError there)
    pred = trainer.predict(DS())
    logits = pred.predictions[0] if isinstance(pred.predictions, tuple) else pred.predictions
    assert logits.shape[0] == len(rows) and logits.shape[-1] == 4096 and np.isfinite(pred.metrics["test_loss"])
    assert pred.label_ids.shape[0] == len(rows)


# ============================================================================ grouped GEMM (MoE experts in one launch)
@pytest.mark.parametrize("variant", ["", "0", "3", "4", "5", "10", "12"])
def test_grouped_gemm_rows_and_kslices(variant, monkeypatch):
    """ta_gemm_bf16_nt_grouped against fp32 matmuls of the same bf16 operands: ragged segments incl. an EMPTY expert and
    partial tiles, a gather list, bias + GELU; the K-slice form with an empty slice (its gradient must be exactly zero)."""
    from tiny_audio_amd import ops
    _force_variant(monkeypatch, variant)
    torch.manual_seed(1)
    E, N, K = 4, 320, 256
    counts = [300, 0, 257, 70]
    base, segs = 0, []
    for c in counts:
        segs += [base, c]; base += (c + 63) // 64 * 64
    Smax = base + 64
    T = 400
    src = torch.randn(T, K, device=DEV).to(torch.bfloat16)
    perm = torch.full((Smax,), -1, dtype=torch.int32)
    rng = np.random.RandomState(0)
    for e, c in enumerate(counts):
        perm[segs[2 * e]: segs[2 * e] + c] = torch.from_numpy(rng.randint(0, T, c).astype(np.int32))
    W = (torch.randn(E, N, K, device=DEV) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(E, N, device=DEV)
    seg = torch.tensor(segs, dtype=torch.int32, device=DEV)
    permd = perm.to(DEV)
    for out_dtype, act in ((torch.bfloat16, 1), (torch.float32, 0)):
        out = torch.full((Smax, N), 7.0, device=DEV, dtype=out_dtype)
        ops.gemm_nt_grouped(src, W, out, sum(counts), N, K, seg=seg, n_groups=E, bias=bias, act=act, a_idx=permd, w_stride=N * K)
        ref = torch.full((Smax, N), 7.0, device=DEV)
        for e, c in enumerate(counts):
            rows = permd[segs[2 * e]: segs[2 * e] + c].long()
            y = src[rows].float() @ W[e].float().t() + bias[e]
            ref[segs[2 * e]: segs[2 * e] + c] = torch.nn.functional.gelu(y) if act else y
        tol = 3e-2 if out_dtype == torch.bfloat16 else 2e-3
        assert float((out.float() - ref).abs().max()) < tol * float(ref.abs().max())     # rows outside every segment untouched (7.0)
    # K-slice form: dW[e] = Y[:, slice e] X[:, slice e]^T over 64-aligned slices of the slot axis
    Mo, No = 192, 320
    Y = torch.randn(Mo, Smax, device=DEV).to(torch.bfloat16)
    X = torch.randn(No, Smax, device=DEV).to(torch.bfloat16)
    kr = []
    for e, c in enumerate(counts):
        kr += [segs[2 * e] // 64, (segs[2 * e] + (c + 63) // 64 * 64) // 64]
    out = torch.full((E, Mo, No), 3.0, device=DEV)
    ops.gemm_nt_grouped(Y, X, out, Mo, No, Smax, krange=torch.tensor(kr, dtype=torch.int32, device=DEV), n_groups=E, c_stride=Mo * No)
    for e in range(E):
        a, b = kr[2 * e] * 64, kr[2 * e + 1] * 64
        ref = Y[:, a:b].float() @ X[:, a:b].float().t()
        assert float((out[e] - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + (0.0 if b > a else 0.0), e
    assert float(out[1].abs().max()) == 0.0                                              # the empty expert
