"""-m "not gpu": exercise the Python plumbing of the MI355X path end to end on a GPU-less box.

``_lib.DRY_RUN`` swaps every kernel-launching C entry point for a stub that only marshals its arguments through
the real ctypes prototypes (arity / type / pointer conversion) -- so a wrong argument order, a missing tensor or
a shape bug in the host code fails here instead of on the GPU box.  No numerics are checked (nothing is computed);
the numerics live in the -m gpu tests.
"""
import numpy as np
import pytest
import torch

from oracle import weights as OW
from tiny_audio_amd import _lib


class _StubTok:
    def convert_tokens_to_ids(self, t):
        return 999


@pytest.fixture()
def dry():
    _lib.DRY_RUN = True
    try:
        yield _lib.lib()
    finally:
        _lib.DRY_RUN = False
        _lib._LIB = None


def test_full_training_step_plumbing(dry):
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    from tiny_audio_amd.asr_processing import ASRProcessor, LogMelFeatureExtractor
    from tiny_audio_amd.trainer import ASRTrainer, TrainingArguments
    enc, lm = OW.enc_config(hidden=256, ffn=512, layers=2, heads=4), OW.lm_config(vocab=1000, hidden=256, ffn=512, layers=2, heads=4, kv_heads=2)
    cfg = ASRConfig(audio_config=enc, text_config=lm, projector_hidden_dim=128, audio_token_id=999, audio_token_dropout=0.1)
    m = ASRModel(cfg, device="cpu", init="none")
    m.audio_tower.load_state_dict_hf(OW.init_encoder(enc, 0))
    m.language_model.load_state_dict_hf(OW.init_lm(lm, 1))
    m.load_state_dict({"projector." + k: torch.from_numpy(v) for k, v in OW.init_mlp_projector(256, 256, 128).items()})
    assert set(m.state_dict()) == {"projector.linear_1.weight", "projector.norm.weight", "projector.linear_2.weight",
                                   "projector.norm_2.weight"}                     # asr_modeling.py:398-422
    fe = LogMelFeatureExtractor(128, "cpu")
    f = fe([OW.synthetic_wave(0, 16000), OW.synthetic_wave(1, 12000)], sampling_rate=16000)
    assert f["input_features"].shape == (2, 128, 100) and f["attention_mask"].shape == (2, 100)
    proc = ASRProcessor(fe, _StubTok(), m.projector)
    assert proc.audio_token_counts(torch.ones(2, 100, dtype=torch.int32)).tolist() == [12, 12]
    ids, att, lab, counts = OW.synthetic_tokens(2, [12, 9], 1000, 999, 990, 991, n_text=10, n_suffix=4, ragged=True)
    batch = dict(input_ids=torch.from_numpy(ids), input_features=f["input_features"], attention_mask=torch.from_numpy(att),
                 labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts))
    m.train()
    out = m(**batch, label_meta=(torch.zeros(40, dtype=torch.int32), torch.zeros(40, dtype=torch.int64), 19))
    assert out.logits.shape == (2, ids.shape[1], 1000) and out.loss.shape == ()
    out.loss.backward()
    for p in m.projector.parameters():
        assert p.grad is not None and p.grad.shape == p.shape
    tr = ASRTrainer(m, TrainingArguments(gradient_accumulation_steps=2))
    lm_meta = (torch.zeros(40, dtype=torch.int32), torch.zeros(40, dtype=torch.int64), 19)
    tr.training_step({**batch, "label_meta": lm_meta})
    assert tr.global_step == 0
    tr.training_step({**batch, "label_meta": lm_meta})
    assert tr.global_step == 1 and m.projector._pack_versions is None
    names = set(dry.calls)
    for must in ("ta_logmel_f32", "ta_encoder_forward", "ta_mlp_projector_forward", "ta_mlp_projector_backward",
                 "ta_audio_index", "ta_lm_forward_loss", "ta_lm_backward", "ta_bernoulli_keep", "ta_grad_sqnorm", "ta_adamw_step_multi"):
        assert must in names, must
    # random-init path + weight export round trip (bench.py's cpu_baseline leg)
    m2 = ASRModel(cfg, device="cpu", init="random", seed=3)
    sd = m2.audio_tower.export_state_dict_hf()
    assert set(sd) == set(OW.init_encoder(enc, 0)) and sd["conv2.weight"].shape == (256, 256, 3)
    sl = m2.language_model.export_state_dict_hf()
    assert set(sl) == set(OW.init_lm(lm, 1)) and sl["model.layers.1.mlp.up_proj.weight"].shape == (512, 256)
    enc2 = type(m2.audio_tower)(cfg.audio_config, "cpu").load_state_dict_hf(OW.init_encoder(enc, 0))
    rt = enc2.export_state_dict_hf()
    w0 = OW.init_encoder(enc, 0)
    for k in ("conv1.weight", "layers.1.self_attn.v_proj.bias", "layers.0.mlp.fc2.weight"):
        np.testing.assert_allclose(rt[k], w0[k], atol=2e-2 * np.abs(w0[k]).max())      # bf16 storage round trip


def test_moe_plumbing(dry):
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    enc, lm = OW.enc_config(hidden=256, ffn=512, layers=1, heads=4), OW.lm_config(vocab=1000, hidden=256, ffn=512, layers=1, heads=4, kv_heads=2)
    cfg = ASRConfig(audio_config=enc, text_config=lm, projector_hidden_dim=128, audio_token_id=999, projector_type="moe")
    m = ASRModel(cfg, device="cpu", init="random")
    assert set(m.projector.state_dict()) == set(OW.init_moe_projector(256, 256, 128))       # reference key names
    assert sum(p.numel() for p in ASRModel(ASRConfig(projector_type="moe"), device="cpu", init="none").projector.parameters()) == 31_493_120
    ids, att, lab, counts = OW.synthetic_tokens(2, [12, 12], 1000, 999, 990, 991, n_text=10, n_suffix=4)
    m.train()
    out = m(input_ids=torch.from_numpy(ids), input_features=torch.zeros(2, 128, 100), attention_mask=torch.from_numpy(att),
            labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts),
            label_meta=(torch.zeros(40, dtype=torch.int32), torch.zeros(40, dtype=torch.int64), 22))
    out.loss.backward()
    for n, p in m.projector.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, n
    assert "ta_moe_projector_forward" in dry.calls and "ta_moe_projector_backward_dev" in dry.calls


def test_lora_stage2_plumbing(dry):
    """Row a11: use_lora + freeze_projector -> exactly the 8 stacked adapter Parameters train; peft-named state dict
    round-trips through the stacked layout; the C structs carry per-layer pointers into the masters."""
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    from tiny_audio_amd.trainer import ASRTrainer, TrainingArguments
    enc, lm = OW.enc_config(hidden=256, ffn=512, layers=1, heads=4), OW.lm_config(vocab=1000, hidden=256, ffn=512, layers=3, heads=4, kv_heads=2)
    cfg = ASRConfig(audio_config=enc, text_config=lm, projector_hidden_dim=128, audio_token_id=999, use_lora=True,
                    freeze_projector=True)
    m = ASRModel(cfg, device="cpu", init="random")
    train = {n: p for n, p in m.named_parameters() if p.requires_grad}
    assert sorted(train) == sorted(f"language_model.lora_{ab}_{g}" for ab in ("la", "lb") for g in ("qkv", "o", "gu", "d"))
    # the true Qwen3-0.6B adapter has 5,046,272 trainable parameters ("5.05 M")
    full = ASRModel(ASRConfig(use_lora=True, freeze_projector=True), device="cpu", init="none")
    assert sum(p.numel() for p in full.language_model.lora_parameters()) == 5_046_272
    lmod = m.language_model
    assert float(lmod.lora_lb_qkv.detach().abs().max()) == 0.0 and float(lmod.lora_la_qkv.detach().abs().max()) <= 1 / 16 + 1e-6   # peft init
    lo = OW.init_lora(lm, rank=8, seed=4)
    lmod.load_lora_state_dict(lo)
    back = lmod.export_lora_state_dict(prefix="model.", suffix="")
    assert set(back) == set(lo)
    for k in lo:
        np.testing.assert_array_equal(back[k].numpy(), lo[k])
    sd = m.state_dict()
    assert "language_model.base_model.model.model.layers.2.mlp.down_proj.lora_B.weight" in sd
    m2 = ASRModel(cfg, device="cpu", init="random", seed=5)
    m2.load_state_dict(sd)
    assert torch.equal(m2.language_model.lora_lb_gu, lmod.lora_lb_gu) and torch.equal(m2.language_model.lora_la_o, lmod.lora_la_o)
    ids, att, lab, counts = OW.synthetic_tokens(2, [12, 12], 1000, 999, 990, 991, n_text=10, n_suffix=4)
    meta = (torch.zeros(40, dtype=torch.int32), torch.zeros(40, dtype=torch.int64), 22)
    batch = dict(input_ids=torch.from_numpy(ids), input_features=torch.zeros(2, 128, 100), attention_mask=torch.from_numpy(att),
                 labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts), label_meta=meta)
    m.train()
    out = m(**batch)
    out.loss.backward()
    for n, p in train.items():
        assert p.grad is not None and p.grad.shape == p.shape, n
    assert all(p.grad is None for p in m.projector.parameters())
    assert lmod._w.lora_rank == 8 and abs(lmod._w.lora_scale - 4.0) < 1e-6
    step = lmod.lora_lb_gu[0].numel() * 4
    assert lmod._layers_arr[2].lb_gu == lmod.lora_lb_gu.data_ptr() + 2 * step
    tr = ASRTrainer(m, TrainingArguments())
    assert tr.flat.n == sum(p.numel() for p in train.values())
    tr.training_step(batch)                                   # masters re-homed into the flat buffer -> pointers rebound
    assert lmod._layers_arr[1].la_d == lmod.lora_la_d.data_ptr() + lmod.lora_la_d[0].numel() * 4
    assert lmod.lora_la_d.data_ptr() >= tr.flat.flat_p.data_ptr()
    # other values of asr_config.py:72-75's knobs (round 4): rank 16 on a target subset -> peft's count, untargeted members stay zero
    # and are not exported, the C struct carries the live groups; 3 r > 64 and unknown targets are refused
    m3 = ASRModel(ASRConfig(audio_config=enc, text_config=lm, use_lora=True, lora_rank=16, lora_alpha=16,
                            lora_target_modules=["q_proj", "v_proj", "down_proj"]), device="cpu", init="random")
    l3 = m3.language_model
    assert l3.lora_rank == 16 and l3.lora_groups == 0b1001 and l3._w.lora_groups == 0b1001 and abs(l3._w.lora_scale - 1.0) < 1e-6
    D, F, hd = 256, 512, lm["head_dim"]
    assert l3.lora_param_count() == 3 * 16 * ((D + 4 * hd) + (D + 2 * hd) + (F + D))
    la3, gu3 = l3.lora_la_qkv.detach(), l3.lora_la_gu.detach()
    assert float(la3[:, 16:32].abs().max()) == 0.0 and float(la3[:, :16].abs().max()) > 0 and float(gu3.abs().max()) == 0.0
    sd3 = l3.export_lora_state_dict(prefix="model.", suffix="")
    assert len(sd3) == 3 * 3 * 2 and not any("k_proj" in k or "o_proj" in k or "gate_proj" in k for k in sd3)
    lo3 = OW.init_lora(lm, rank=16, seed=4, targets=("q_proj", "v_proj", "down_proj"))
    l3.load_lora_state_dict(lo3)
    back3 = l3.export_lora_state_dict(prefix="model.", suffix="")
    assert set(back3) == set(lo3) and all(np.array_equal(back3[k].numpy(), lo3[k]) for k in lo3)
    with pytest.raises(NotImplementedError):
        ASRModel(ASRConfig(audio_config=enc, text_config=lm, use_lora=True, lora_rank=32), device="cpu", init="none")
    with pytest.raises(ValueError):
        ASRModel(ASRConfig(audio_config=enc, text_config=lm, use_lora=True, lora_target_modules=["q_proj", "lm_head"]), device="cpu", init="none")


def test_full_finetune_plumbing(dry, tmp_path):
    """Section 8(f) rank 4: freeze_language_model=False -> the LM's fp32 masters are Parameters under ``language_model.``,
    the state dict carries them under the reference's names, the kernels' norm / embedding pointers follow the masters into
    the trainer's flat buffer, gradients land in Parameter.grad, decoder LR / no-decay groups apply."""
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    from tiny_audio_amd.trainer import ASRTrainer, TrainingArguments
    enc, lm = OW.enc_config(hidden=256, ffn=512, layers=1, heads=4), OW.lm_config(vocab=1000, hidden=256, ffn=512, layers=3, heads=4, kv_heads=2)
    cfg = ASRConfig(audio_config=enc, text_config=lm, projector_hidden_dim=128, audio_token_id=999, freeze_language_model=False)
    m = ASRModel(cfg, device="cpu", init="random")
    lmod = m.language_model
    train = {n: p for n, p in m.named_parameters() if p.requires_grad}
    ft = [n for n in train if n.startswith("language_model.")]
    assert sorted(ft) == sorted("language_model.ft_" + k for k in ("wqkv", "wo", "wgu", "wd", "ln_in_w", "ln_post_w", "qn_w", "kn_w", "norm_w", "embed"))
    wl = OW.init_lm(lm, 1)
    assert sum(train[n].numel() for n in ft) == sum(v.size for v in wl.values())
    # the true Qwen3-0.6B: 596,049,920 parameters minus the resize to V = 151,670 rows
    assert lmod._w.train_base == 1 and lmod._fp32_src is None
    sd = m.state_dict()
    assert sd["language_model.model.layers.2.self_attn.k_proj.weight"].shape == (256, 256)
    assert "language_model.model.embed_tokens.weight" in sd and "projector.linear_1.weight" in sd
    # round trip through the reference's names (the oracle's seeded weights): masters and bf16 images follow
    m.load_state_dict({**{"language_model." + k: torch.from_numpy(v) for k, v in wl.items()},
                       **{k: v for k, v in sd.items() if k.startswith("projector.")}})
    back = lmod.ft_state_dict_hf()
    for k, v in wl.items():
        np.testing.assert_array_equal(back[k].numpy(), v)
    assert torch.equal(lmod._bufs["layers.1.wd"], torch.from_numpy(wl["model.layers.1.mlp.down_proj.weight"]).to(torch.bfloat16))
    assert torch.equal(lmod._bufs["layers.1.wd_t"], lmod._bufs["layers.1.wd"].t())
    ids, att, lab, counts = OW.synthetic_tokens(2, [12, 12], 1000, 999, 990, 991, n_text=10, n_suffix=4)
    meta = (torch.zeros(40, dtype=torch.int32), torch.zeros(40, dtype=torch.int64), 22)
    batch = dict(input_ids=torch.from_numpy(ids), input_features=torch.zeros(2, 128, 100), attention_mask=torch.from_numpy(att),
                 labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts), label_meta=meta)
    m.train()
    dry.calls.clear()
    out = m(**batch)
    out.loss.backward()
    assert dry.calls.count("ta_lm_backward") == 1
    for n, p in train.items():
        assert p.grad is not None and p.grad.shape == p.shape, n
    tr = ASRTrainer(m, TrainingArguments(learning_rate=1e-3, weight_decay=0.1), decoder_learning_rate=1e-4)
    assert lmod.accumulate_into_grad and tr.flat.n >= sum(p.numel() for p in train.values())
    # a decoder_* override selects the reference's own grouping (scripts/train.py:397-405), where only nn.LayerNorm and
    # biases are exempt: Qwen3's RMSNorm scales decay there; without overrides HF's name patterns exempt them
    dec = dict(zip(tr.flat.names, tr.flat.decay))
    assert dec["language_model.ft_wqkv"] and dec["language_model.ft_embed"] and dec["language_model.ft_ln_in_w"] \
        and dec["language_model.ft_qn_w"] and dec["language_model.ft_norm_w"]
    from tiny_audio_amd.trainer import decay_flags
    dec0 = dict(zip(tr.flat.names, decay_flags(tr.flat.names, tr.flat.params, overrides=False)))
    assert dec0["language_model.ft_wqkv"] and dec0["language_model.ft_embed"] and not dec0["language_model.ft_ln_in_w"] \
        and not dec0["language_model.ft_qn_w"] and not dec0["language_model.ft_norm_w"]
    assert tr.group_hparams("language_model.ft_wd", True) == (1e-4, 0.1) and tr.group_hparams("projector.linear_1.weight", True) == (1e-3, 0.1)
    tr.training_step(batch)                         # masters re-homed into the flat buffer -> pointers rebound, images rebuilt
    assert lmod._w.embed_f32 == lmod.ft_embed.data_ptr() and lmod.ft_embed.data_ptr() >= tr.flat.flat_p.data_ptr()
    assert lmod._layers_arr[2].qn_w == lmod.ft_qn_w.data_ptr() + 2 * 128 * 4
    assert lmod._ft_versions is None                # optimizer step done: images are rebuilt by the next forward
    tr.training_step(batch)
    assert lmod._ft_versions is None or isinstance(lmod._ft_versions, tuple)
    # checkpoint: the fine-tuned LM travels in model.safetensors (asr_modeling.py:409-421)
    m.save_pretrained(tmp_path / "ck")
    m2 = ASRModel.from_pretrained(tmp_path / "ck", device="cpu")
    assert m2.language_model.train_base and torch.equal(m2.language_model.ft_wo, lmod.ft_wo)
    with pytest.raises(NotImplementedError):
        m.language_model.enable_lora()


def test_qformer_plumbing(dry):
    """Section 8(f) rank 4: the QFormer projector inside ASRModel -- reference parameter names / count, 102 audio tokens for
    500 encoder frames, one full forward + backward + optimizer step through the primitive wrappers."""
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    from tiny_audio_amd.trainer import ASRTrainer, TrainingArguments
    full = ASRModel(ASRConfig(projector_type="qformer"), device="cpu", init="none")
    assert sum(p.numel() for p in full.projector.parameters()) == 53_795_584
    assert set(full.projector.state_dict()) == set(OW.init_qformer_projector(1280, 1024))
    assert full.projector.get_output_length(500) == 102 and full.projector.get_output_length(torch.tensor([500, 15, 16])).tolist() == [102, 3, 6]
    enc, lm = OW.enc_config(hidden=256, ffn=512, layers=1, heads=4), OW.lm_config(vocab=1000, hidden=256, ffn=512, layers=1, heads=4, kv_heads=2)
    cfg = ASRConfig(audio_config=enc, text_config=lm, projector_type="qformer", qformer_num_heads=4, qformer_intermediate_size=512,
                    audio_token_id=999)
    m = ASRModel(cfg, device="cpu", init="random")
    n_tok = m.projector.get_output_length(50)                       # T = 100 mel frames -> S = 50 -> 4 windows -> 12 tokens
    assert n_tok == 12
    ids, att, lab, counts = OW.synthetic_tokens(2, [12, 12], 1000, 999, 990, 991, n_text=10, n_suffix=4)
    batch = dict(input_ids=torch.from_numpy(ids), input_features=torch.zeros(2, 128, 100), attention_mask=torch.from_numpy(att),
                 labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts),
                 label_meta=(torch.zeros(40, dtype=torch.int32), torch.zeros(40, dtype=torch.int64), 22))
    m.train()
    tr = ASRTrainer(m, TrainingArguments())
    tr.training_step(batch)
    assert tr.global_step == 1
    for must in ("ta_layernorm_res_fwd", "ta_layernorm_bwd", "ta_attn_small_fwd", "ta_attn_small_bwd", "ta_gelu_fwd", "ta_gelu_bwd",
                 "ta_colsum", "ta_bernoulli_keep"):
        assert must in dry.calls, must
    for n, p in m.projector.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, n


def test_mosa_plumbing(dry):
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    full = ASRModel(ASRConfig(projector_type="mosa"), device="cpu", init="none")
    assert sum(p.numel() for p in full.projector.parameters()) == 42_951_428
    assert set(full.projector.state_dict()) == set(OW.init_mosa_projector(1280, 1024)) and full.projector.get_output_length(500) == 125
    enc, lm = OW.enc_config(hidden=256, ffn=512, layers=1, heads=4), OW.lm_config(vocab=1000, hidden=256, ffn=512, layers=1, heads=4, kv_heads=2)
    m = ASRModel(ASRConfig(audio_config=enc, text_config=lm, projector_type="mosa", audio_token_id=999), device="cpu", init="random")
    n_tok = m.projector.get_output_length(50)
    ids, att, lab, counts = OW.synthetic_tokens(2, [n_tok, n_tok], 1000, 999, 990, 991, n_text=10, n_suffix=4)
    m.train()
    out = m(input_ids=torch.from_numpy(ids), input_features=torch.zeros(2, 128, 100), attention_mask=torch.from_numpy(att),
            labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts),
            label_meta=(torch.zeros(40, dtype=torch.int32), torch.zeros(40, dtype=torch.int64), 22))
    out.loss.backward()
    for n, p in m.projector.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, n
    for must in ("ta_mix_fwd", "ta_mix_bwd", "ta_relu_fwd", "ta_relu_bwd", "ta_gelu_bwd", "ta_colsum"):
        assert must in dry.calls, must


def test_generate_plumbing(dry):
    """Section 8(f) rank 1: the argument marshalling of prefill / decode step / greedy bookkeeping, the reference's
    error behaviour, and the trim of surplus columns (nothing is computed under DRY_RUN: every token is 0)."""
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    enc, lm = OW.enc_config(hidden=256, ffn=512, layers=1, heads=4), OW.lm_config(vocab=1000, hidden=256, ffn=512, layers=2, heads=4, kv_heads=2)
    for lora in (False, True):
        cfg = ASRConfig(audio_config=enc, text_config=lm, projector_hidden_dim=128, audio_token_id=999, pad_token_id=990,
                        eos_token_id=991, use_lora=lora)
        m = ASRModel(cfg, device="cpu", init="random")
        ids = torch.tensor([[5, 6] + [999] * 12 + [7, 8]] * 2)
        kw = dict(input_ids=ids, input_features=torch.zeros(2, 128, 100), audio_attention_mask=torch.ones(2, 100, dtype=torch.int64),
                  attention_mask=torch.ones_like(ids))
        dry.calls.clear()
        out = m.generate(**kw, max_new_tokens=5, eos_token_id=[])      # (every dry-run token is the pad id)
        assert out.shape == (2, 5) and out.dtype == torch.int64
        assert m.generate(**kw, max_new_tokens=5).shape == (2, 1)        # pad is an eos id: trimmed after step 0
        dry.calls.clear(); m.generate(**kw, max_new_tokens=5, eos_token_id=[])
        assert dry.calls.count("ta_lm_prefill") == 1 and dry.calls.count("ta_lm_decode_step") == 4
        assert dry.calls.count("ta_greedy_advance") == 5 and dry.calls.count("ta_argmax_f32") == 5
        assert dry.calls.count("ta_logits_suppress_until") == 0
        dry.calls.clear(); m.generate(**kw, max_new_tokens=5, min_new_tokens=3)          # generation_config.min_new_tokens (round 4)
        assert dry.calls.count("ta_logits_suppress_until") == dry.calls.count("ta_argmax_f32") >= 1
    with pytest.raises(ValueError, match="input_features required"):
        m.generate(input_ids=ids)
    with pytest.raises(ValueError, match="audio_attention_mask required"):
        m.generate(input_ids=ids, input_features=torch.zeros(2, 128, 100))
    with pytest.raises(ValueError, match="input_ids required"):
        m.generate(input_features=torch.zeros(2, 128, 100), audio_attention_mask=torch.ones(2, 100, dtype=torch.int64))
    with pytest.raises(NotImplementedError):
        m.generate(**kw, num_beams=3)
    dry.calls.clear(); m.generate(**kw, max_new_tokens=3, do_sample=True, top_k=5, seed=1, eos_token_id=[])     # sampling (round 4)
    assert dry.calls.count("ta_logits_warp") == dry.calls.count("ta_sample_f32") == 3 and dry.calls.count("ta_argmax_f32") == 0
    # streaming (asr_modeling.py:648-760): one clip, one token per step, same launches as generate
    one = dict(input_ids=ids[:1], input_features=torch.zeros(1, 128, 100), audio_attention_mask=torch.ones(1, 100, dtype=torch.int64))
    dry.calls.clear()
    toks = list(m.generate_streaming(**one, return_token_ids=True, max_new_tokens=5, eos_token_id=[]))
    assert toks == [990] * 5 and dry.calls.count("ta_lm_decode_step") == 4
    assert list(m.generate_streaming(**one, return_token_ids=True, max_new_tokens=5)) == [990]      # pad is an eos id
    with pytest.raises(ValueError, match="one clip at a time"):
        next(m.generate_streaming(input_ids=ids, input_features=torch.zeros(2, 128, 100),
                                  audio_attention_mask=torch.ones(2, 100, dtype=torch.int64), return_token_ids=True))
    with pytest.raises(ValueError, match="needs a tokenizer"):
        next(m.generate_streaming(**one))


def test_streaming_text_release():
    """The text side of generate_streaming: word-boundary release (as transformers' TextStreamer), special tokens
    skipped, <think> spans dropped (tiny_audio/asr_modeling.py:737-757)."""
    from tiny_audio_amd.asr_modeling import _TextPieces, _ThinkGate

    class Tok:
        vocab = {1: "hel", 2: "lo", 3: " wor", 4: "ld", 5: " ", 6: "<think>", 7: "secret", 8: "</think>", 9: "\n", 10: "\u4f60",
                 11: "<eos>"}

        def decode(self, ids, skip_special_tokens=True):
            return "".join(self.vocab[i] for i in ids if not (skip_special_tokens and i == 11))

    def run(ids):
        p, g, out = _TextPieces(Tok()), _ThinkGate(), []
        for i in ids:
            out += list(g.feed(p.push(i)))
        out += list(g.feed(p.flush()))
        tail = g.flush()
        return [o for o in out + ([tail] if tail else []) if o]

    assert run([1, 2, 3, 4, 11]) == ["hello ", "world"]             # released up to the last space, rest on flush
    assert run([1, 2, 9, 3, 4]) == ["hello\n", " ", "world"]        # a finished line is released whole
    assert run([10, 10]) == ["\u4f60", "\u4f60"]                    # CJK characters at once
    assert "".join(run([1, 5, 6, 7, 5, 8, 3, 4, 5])) == "hel  world "
    assert "secret" not in "".join(run([6, 7, 5, 7, 5]))             # unterminated think block: nothing leaks


def test_checkpoint_interchange(dry, tmp_path):
    """Section 8(f) rank 3: (i) a checkpoint written from the REFERENCE model (tests/golden/ckpt_small: its
    state_dict() via safetensors + its config.json) loads; (ii) save -> load round trip incl. the PEFT adapter layout."""
    import json, os
    from safetensors.torch import load_file
    from tiny_audio_amd.asr_modeling import ASRModel
    from tests.golden import recipe as R
    ref_dir = os.path.join(os.path.dirname(__file__), "golden", "ckpt_small")
    m = ASRModel.from_pretrained(ref_dir, device="cpu", init="none")
    S = R.SMALL
    assert m.config.projector_type == "mlp" and m.config.projector_hidden_dim == S["proj_hidden"]
    assert m.config.text_config.hidden_size == S["lm"]["hidden"] and m.config.audio_config.num_hidden_layers == S["enc"]["layers"]
    assert m.config.text_config.vocab_size == S["lm"]["vocab"] and m.config.encoder_conv_layers == [[1, 3, 1], [1, 3, 2]]
    want = OW.init_mlp_projector(S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"])
    for k, v in m.projector.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), want[k])
    # (ii) our own save: same file names / key names, then back
    cfg = type(m.config)(audio_config=S["enc"], text_config=S["lm"], projector_hidden_dim=S["proj_hidden"], audio_token_id=1023,
                         use_lora=True, freeze_projector=True)
    a = ASRModel(cfg, device="cpu", init="random", seed=3)
    a.language_model.load_lora_state_dict(OW.init_lora(S["lm"], rank=8, seed=4))
    out = tmp_path / "ck"
    a.save_pretrained(out)
    assert sorted(os.listdir(out)) == ["adapter_config.json", "adapter_model.safetensors", "config.json", "model.safetensors"]
    assert set(load_file(str(out / "model.safetensors"))) == set(load_file(os.path.join(ref_dir, "model.safetensors")))
    ad = load_file(str(out / "adapter_model.safetensors"))
    assert "base_model.model.model.layers.1.mlp.down_proj.lora_B.weight" in ad and len(ad) == 2 * 7 * S["lm"]["layers"]
    assert ad["base_model.model.model.layers.0.self_attn.k_proj.lora_A.weight"].shape == (8, S["lm"]["hidden"])
    ac = json.load(open(out / "adapter_config.json"))
    assert ac["peft_type"] == "LORA" and ac["r"] == 8 and ac["lora_alpha"] == 32 and ac["task_type"] == "CAUSAL_LM"
    b = ASRModel.from_pretrained(out, device="cpu", init="random", seed=9)
    assert b.config.use_lora and b.config.freeze_projector and b.language_model.lora_rank == 8
    for (k1, v1), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1


def test_primitive_wrappers_marshal(dry):
    from tiny_audio_amd import ops
    bf, f32 = torch.bfloat16, torch.float32
    A, W = torch.zeros(70, 128, dtype=bf), torch.zeros(256, 128, dtype=bf)
    assert ops.gemm_nt(A, W, bias=torch.zeros(256), residual=torch.zeros(70, 256), act=1, out_dtype=f32).shape == (70, 256)
    assert ops.gemm_nt(A, W, out_dtype=f32, splits=2).dtype == f32
    ops.gemm_nt(A, W, out_dtype=f32, k_ext=(torch.zeros(70, 64, dtype=bf), torch.zeros(256, 64, dtype=bf)))
    x = torch.zeros(10, 256)
    ops.layernorm(x, torch.ones(256), torch.zeros(256), out_f32=True)
    yb, yf, r = ops.rmsnorm_fwd(x, torch.ones(256), act_gelu=True, out_f32=True)
    ops.rmsnorm_bwd(x, x, r, torch.ones(256), dres=x, want_dw=True)
    B, Hq, Hkv, L = 2, 4, 2, 70
    qkv0 = torch.zeros(B * L, (Hq + 2 * Hkv) * 128, dtype=bf)
    cs = torch.zeros(256, 64)
    Q, K, V, QT, KT, VT, rq, rk = ops.lm_qkv_post_fwd(qkv0, torch.ones(128), torch.ones(128), cs, cs, B, Hq, Hkv, L)
    O, lse = ops.attention_fwd(Q, K, VT, L, True, 0.1, kmask=torch.ones(B, L, dtype=torch.int32))
    delta, dOT = ops.attn_bwd_prep(O, O, B, Hq, L)
    dQ, dK, dV = ops.attention_bwd(Q, QT, K, KT, V, O, dOT, lse, delta, L, True, 0.1)
    ops.lm_qkv_post_bwd(dQ, dK, dV, qkv0, rq, rk, torch.ones(128), torch.ones(128), cs, cs, B, Hq, Hkv, L)
    ops.enc_qkv_post(torch.zeros(B * 77, 3 * 4 * 64, dtype=bf), torch.zeros(128, 16), torch.zeros(128, 16), B, 4, 77)
    gu = torch.zeros(9, 64, dtype=bf)
    ops.swiglu_bwd(ops.swiglu_fwd(gu, 32), gu, 32)
    ops.transpose_to_bf16(ops.cast_bf16(torch.zeros(8, 12)), ld_out=64)
    ops.audio_index(torch.zeros(2, 8, dtype=torch.int64), torch.zeros(2, dtype=torch.int64), 3, 5)
    ops.label_rows(torch.zeros(2, 8, dtype=torch.int64))
    ops.cross_entropy(torch.zeros(4, 128), torch.zeros(4, dtype=torch.int64), 100, 0.25)
    ops.bernoulli_keep(10, 0.9, 1, "cpu")
    ops.adamw_step(torch.zeros(8), torch.zeros(8), torch.zeros(8), torch.zeros(8), 1e-3, 0.9, 0.999, 1e-8, 0.0, 1,
                   sqnorm=torch.zeros(1), max_norm=1.0, denom=torch.ones(1))


def test_hub_sized_embedding_is_cut_to_the_tokenizer_vocab(dry):
    """Hub Qwen3 checkpoints carry 151 936 embedding rows; the reference shrinks them to len(tokenizer) after adding
    <audio> (resize_token_embeddings keeps the first rows; tiny_audio/asr_modeling.py:160-171)."""
    from tiny_audio_amd.asr_config import LMConfig
    from tiny_audio_amd.language_model import Qwen3MI355X
    lm = OW.lm_config(vocab=1000, hidden=256, ffn=512, layers=1, heads=4, kv_heads=2)
    w = OW.init_lm(lm, 1)
    big = dict(w)
    extra = np.random.RandomState(0).standard_normal((266, 256)).astype(np.float32)
    big["model.embed_tokens.weight"] = np.concatenate([w["model.embed_tokens.weight"], extra], 0)
    m = Qwen3MI355X(LMConfig(lm), device="cpu").load_state_dict_hf(big)
    assert m._bufs["embed_f32"].shape == (1000, 256)
    np.testing.assert_array_equal(m._bufs["embed_f32"].numpy(), w["model.embed_tokens.weight"])
    assert m._bufs["embed_bf16"].shape[0] == m.vocab_pad and float(m._bufs["embed_bf16"][1000:].abs().sum()) == 0.0
    small = dict(w)
    small["model.embed_tokens.weight"] = w["model.embed_tokens.weight"][:900]
    with pytest.raises(ValueError):
        Qwen3MI355X(LMConfig(lm), device="cpu").load_state_dict_hf(small)
