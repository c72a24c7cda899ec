"""-m "not gpu": host-side logic of the MI355X path -- the C ABI loads and exports every symbol the header
declares, config / length formulas / schedules, feature tables, and the data-parallel reduction (gloo, world 2)."""
import ctypes as C
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tiny_audio_amd import _lib
from tiny_audio_amd.asr_config import ASRConfig, compute_encoder_output_length
from tiny_audio_amd.trainer import FlatTrainable, TrainingArguments, allreduce_flat, lr_multiplier

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------------------- C ABI
def test_library_builds_and_exports_every_declared_symbol():
    _lib.build()
    protos = _lib.parse_header()
    assert len(protos) >= 40
    handle = _lib.lib()                      # binds every prototype; AttributeError if one is not exported
    for name in protos:
        assert hasattr(handle, name), name
    assert handle.ta_version() == 4
    # every extern "C" ta_* symbol the library exports is declared in the header (no undocumented entry points)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.SO_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T ta_" in l}
    assert exported == set(protos), exported ^ set(protos)


def test_header_cites_the_reference_interfaces():
    text = open(_lib.HEADER).read()
    for cite in ("tiny_audio/asr_modeling.py:448-450", "tiny_audio/projectors.py:57-71", "scripts/train.py:327-333",
                 "tiny_audio/asr_modeling.py:27-44", "TF:loss/loss_utils.py:33-71"):
        assert cite in text, cite


def test_host_only_queries_run_without_a_gpu():
    L = _lib.lib()
    mw = _lib.MlpWeights(enc_dim=1280, k=4, hidden=1024, llm_dim=1024, eps=1e-6)
    tape = L.ta_mlp_tape_bytes(C.byref(mw), 32, 500)
    assert tape >= 32 * 125 * (1024 * 4 * 2 + 1024 * 2)       # h1 + h2 (f32) + a1 (bf16)
    ew = _lib.EncoderWeights(hidden=1280, ffn=5120, n_layers=32, heads=20, n_mels=128, max_pos=1500, ln_eps=1e-5)
    ws = L.ta_encoder_workspace_bytes(C.byref(ew), 32, 1000)
    assert 0.3e9 < ws < 2e9
    lw = _lib.LmWeights(vocab=151670, vocab_pad=151680, hidden=1024, ffn=3072, n_layers=28, heads=16, kv_heads=8,
                        head_dim=128, max_pos=4096, eps=1e-6)
    assert 5e9 < L.ta_lm_tape_bytes(C.byref(lw), 32, 192, 1152) < 20e9
    assert L.ta_gemm_splitk_ws_bytes(100, 64, 4) == 100 * 64 * 4 * 4 and L.ta_gemm_splitk_ws_bytes(100, 64, 1) == 0


def test_product_path_has_no_cpu_fallback():
    from tiny_audio_amd.projectors import MLPAudioProjector
    p = MLPAudioProjector(ASRConfig())
    with pytest.raises(_lib.Ta355Error):
        p(torch.zeros(1, 8, 1280))
    src = "".join(open(os.path.join(ROOT, "tiny_audio_amd", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "tiny_audio_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src      # the product never touches the checker


# ----------------------------------------------------------------------------- config / formulas (reference known answers)
def test_config_and_length_formulas():
    c = ASRConfig()
    assert (c.encoder_dim, c.llm_dim, c.projector_pool_stride) == (1280, 1024, 4)
    assert c.audio_config.num_hidden_layers == 32 and c.text_config.num_key_value_heads == 8
    # reference tests/test_asr_config.py:150-175
    assert [compute_encoder_output_length(x) for x in (100, 1, 3000)] == [50, 1, 1500]
    assert compute_encoder_output_length(torch.tensor([100, 1000])).tolist() == [50, 500]
    from tiny_audio_amd.projectors import MLPAudioProjector, PROJECTOR_CLASSES
    p = MLPAudioProjector(c)
    assert [p.get_output_length(x) for x in (100, 104, 4)] == [25, 26, 1]      # tests/test_projectors.py:65-69
    assert set(p.state_dict()) == {"linear_1.weight", "norm.weight", "linear_2.weight", "norm_2.weight"}
    assert sum(v.numel() for v in p.parameters()) == 6_293_504
    assert "mlp" in PROJECTOR_CLASSES
    with pytest.raises(ValueError):
        ASRConfig(audio_config=dict(hidden=1280, heads=10))                      # head_dim 128 encoder: unsupported


def test_feature_tables(golden):
    from tiny_audio_amd.asr_processing import dft_tables, slaney_mel_filters
    g = golden("logmel.npz")
    np.testing.assert_allclose(slaney_mel_filters(128), g["mel_filters"], rtol=1e-6, atol=1e-9)
    dft, win = dft_tables()
    assert dft.shape == (400, 402) and win.shape == (400,)
    x = np.random.RandomState(0).standard_normal(400)
    spec = np.fft.rfft(x * win)
    np.testing.assert_allclose((x * win) @ dft[:, :201].astype(np.float64), spec.real, atol=2e-5)
    np.testing.assert_allclose(-((x * win) @ dft[:, 201:].astype(np.float64)), spec.imag, atol=2e-5)
    np.testing.assert_array_equal(win, torch.hann_window(400).numpy())    # feature_extraction_whisper.py:141


def test_lr_schedules():
    a = TrainingArguments(learning_rate=1e-3, warmup_steps=10, max_steps=110, lr_scheduler_type="cosine")
    assert lr_multiplier(0, a) == 0.0 and lr_multiplier(5, a) == 0.5 and lr_multiplier(10, a) == 1.0
    assert abs(lr_multiplier(60, a) - 0.5) < 1e-12 and lr_multiplier(110, a) < 1e-12
    p = TrainingArguments(learning_rate=1e-3, warmup_steps=500, max_steps=10500, lr_scheduler_type="polynomial",
                          lr_scheduler_kwargs={"power": 0.5})                    # transcription.yaml:22-25
    assert abs(lr_multiplier(500, p) - 1.0) < 1e-9
    assert abs(lr_multiplier(8000, p) - ((1e-3 - 1e-7) * math.sqrt(0.25) + 1e-7) / 1e-3) < 1e-9
    assert lr_multiplier(20000, p) == 1e-7 / 1e-3


def test_flat_trainable_views_and_decay_groups():
    from tiny_audio_amd.projectors import MLPAudioProjector
    m = torch.nn.Module()
    m.projector = MLPAudioProjector(ASRConfig(audio_config=dict(hidden=128, heads=2), text_config=dict(hidden=64, heads=2, kv_heads=1),
                                              projector_hidden_dim=64))
    w0 = m.projector.linear_1.weight.detach().clone()
    ft = FlatTrainable(list(m.named_parameters()), device="cpu")
    assert ft.n == sum(ft.sizes) and ft.flat_g.numel() == ft.n + 2
    assert torch.equal(m.projector.linear_1.weight, w0)
    assert m.projector.linear_1.weight.data_ptr() == ft.flat_p[ft.offsets[0]:].data_ptr()
    # what transformers' get_decay_parameter_names returns for the reference's MLPAudioProjector (checked against the
    # import): ".norm." matches HF's no-decay name patterns, "norm_2" matches none of them and decays
    assert dict(zip(ft.names, ft.decay)) == {"projector.linear_1.weight": True, "projector.norm.weight": False,
                                             "projector.linear_2.weight": True, "projector.norm_2.weight": True}
    from tiny_audio_amd.trainer import decay_flags
    # scripts/train.py:397-405 (any decoder_* / projector_weight_decay override): only nn.LayerNorm and biases are exempt
    assert decay_flags(ft.names, ft.params, overrides=True) == [True, True, True, True]
    (m.projector.linear_2.weight.sum() * 2).backward()                            # autograd accumulates into the flat views
    o = ft.offsets[2]
    assert torch.all(ft.flat_g[o:o + ft.sizes[2]] == 2.0) and float(ft.flat_g[:ft.offsets[2]].abs().sum()) == 0.0
    ft.zero_grad()
    assert float(ft.flat_g.abs().sum()) == 0.0


# ----------------------------------------------------------------------------- data parallel: gloo, world_size 2
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    lin = torch.nn.Linear(8, 4, bias=False)                    # same weights on every rank
    ft = FlatTrainable([("w", lin.weight)], device="cpu")
    # rank r holds (r+1)*3 "label tokens"; the per-rank objective is the SUM of per-token losses
    n_tok = (rank + 1) * 3
    x = torch.arange(n_tok * 8, dtype=torch.float32).reshape(n_tok, 8) / 10 + rank
    loss_sum = (lin(x) ** 2).sum()
    ft.zero_grad()
    loss_sum.backward()
    ft.count_slot.add_(float(n_tok)); ft.loss_slot.add_(loss_sum.detach().reshape(1))
    allreduce_flat(ft.flat_g)
    q.put((rank, ft.grads.clone().numpy(), float(ft.count_slot), float(ft.loss_slot)))
    dist.destroy_process_group()


def test_dp_allreduce_matches_single_process_global_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in procs])
    [p.join(60) for p in procs]
    torch.manual_seed(0)
    lin = torch.nn.Linear(8, 4, bias=False)
    xs = [torch.arange((r + 1) * 3 * 8, dtype=torch.float32).reshape((r + 1) * 3, 8) / 10 + r for r in range(world)]
    total = sum((lin(x) ** 2).sum() for x in xs)
    total.backward()
    for rank, g, cnt, ls in res:
        np.testing.assert_allclose(g, lin.weight.grad.reshape(-1).numpy(), rtol=1e-5)    # SUM over ranks
        assert cnt == 9.0 and abs(ls - float(total)) < 1e-3 * float(total)
    # => grads / count == gradient of the global MEAN over all 9 tokens: HF Trainer's sum-CE / num_items semantics


# ----------------------------------------------------------------------------- text post-processing + WER (8(f) rank 1)
def test_text_postprocessing_matches_reference_outputs():
    import json, os
    from tiny_audio_amd import eval_text as ET
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "text_post.json")))
    for src, want in g["truncate"]:
        assert ET.truncate_repetitions(src) == want, src
    for src, want in g["truncate_min2"]:
        assert ET.truncate_repetitions(src, 2) == want, src
    for src, want in g["think"]:
        assert ET.strip_think(src) == want, src
    vocab = {1: "<think>", 2: "</think>", 3: "hello", 4: "there", 9: "<eos>"}
    text = ET.postprocess_tokens([1, 2, 3, 4, 4, 4, 9, 9], [9], lambda ids: " ".join(vocab[i] for i in ids))
    assert text == "hello there"


def test_word_error_rate_known_answers():
    from tiny_audio_amd.eval_text import word_error_rate as wer
    assert wer("the cat sat", "the cat sat") == 0.0
    assert wer("the cat sat", "the cat") == pytest.approx(1 / 3)               # one deletion
    assert wer("the cat sat", "the big cat sat") == pytest.approx(1 / 3)       # one insertion
    assert wer("the cat sat", "a cat sat") == pytest.approx(1 / 3)             # one substitution
    assert wer("a b c d", "") == 1.0
    assert wer("a b", "x y z w") == 2.0                                        # WER can exceed 1
    # corpus level: edits are pooled over the corpus, not averaged per sentence (jiwer.wer on lists)
    assert wer(["a b c d", "e f"], ["a b c d", "x y"]) == pytest.approx(2 / 6)
    assert wer("Hello, World", "hello world", normalize=lambda s: s.lower().replace(",", "")) == 0.0
    with pytest.raises(ValueError):
        wer(["a"], ["a", "b"])


# ----------------------------------------------------------------------------- collator (8(f) rank 2)
class _StubChatTokenizer:
    """Whitespace tokenizer with a ChatML-shaped template: <|im_start|>role\n content <|im_end|>\n per message."""
    pad_token_id, eos_token_id = 0, 2

    def __init__(self):
        self.vocab = {"<pad>": 0, "<|im_start|>": 1, "<|im_end|>": 2, "<audio>": 3, "\n": 4}

    def _id(self, w):
        return self.vocab.setdefault(w, len(self.vocab))

    def convert_tokens_to_ids(self, t):
        return self.vocab.get(t)

    def _words(self, text):
        return [w for w in text.replace("<audio>", " <audio> ").split() if w]

    def apply_chat_template(self, messages, tokenize=True, add_generation_prompt=False, **_):
        ids = []
        for m in messages:
            ids += [1, self._id(m["role"]), 4] + [self._id(w) for w in self._words(m["content"])] + [2, 4]
        if add_generation_prompt:
            ids += [1, self._id("assistant"), 4]
        return ids

    def decode(self, ids, skip_special_tokens=True):
        inv = {v: k for k, v in self.vocab.items()}
        return " ".join(inv[i] for i in ids if not (skip_special_tokens and i in (0, 1, 2, 4)))


def test_label_normaliser_matches_reference_outputs():
    import json, os
    from tiny_audio_amd.collator import normalize_label
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "text_post.json")))
    assert len(g["normalize_label"]) >= 20
    for src, want in g["normalize_label"]:
        assert normalize_label(src) == want, src


def test_collator_contract(dry_lib_for_features):
    """What the reference's tests/test_data_collator.py pins: assistant text + <|im_end|> unmasked, system / user prompt and
    every <audio> placeholder masked, one placeholder per projector output frame, row filters, batch keys."""
    from tiny_audio_amd.collator import DataCollator, MultiTaskDataCollator
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.projectors import MLPAudioProjector
    tok, fe = _StubChatTokenizer(), dry_lib_for_features
    proj = MLPAudioProjector(ASRConfig())
    col = DataCollator(tok, fe, 16000, system_prompt="You are a helpful assistant.", projector=proj)
    rng = np.random.RandomState(0)
    mk = lambda text, sec: {"audio": {"array": (0.1 * rng.standard_normal(int(sec * 16000))).astype(np.float32), "sampling_rate": 16000}, "text": text}
    batch = col([mk("Hello World this is a TEST <comma>", 1.0), mk("second clip", 2.0)])
    assert set(batch) == {"input_ids", "attention_mask", "labels", "prompts", "prompt_attention_mask", "input_features",
                          "audio_attention_mask", "audio_token_counts"}          # trl's collation also emits the two prompt tensors
    assert batch["audio_token_counts"].tolist() == [12, 25]                       # 100 / 200 mel frames -> 50 / 100 -> 12 / 25
    ids, lab, att = batch["input_ids"], batch["labels"], batch["attention_mask"]
    for i, n_audio in enumerate([12, 25]):
        row, lr = ids[i].tolist(), lab[i].tolist()
        assert row.count(3) == n_audio and all(lr[j] == -100 for j, t in enumerate(row) if t == 3)       # <audio> masked
        unmasked = [t for t, l in zip(row, lr) if l != -100]
        assert unmasked[-2:] == [2, 4] or unmasked[-1] == 2 or 2 in unmasked      # the stop token is learned
        text = tok.decode(unmasked)
        assert text == ("hello world this is a test" if i == 0 else "second clip")
        assert all(l == -100 for l, a in zip(lr, att[i].tolist()) if a == 0)      # padding masked
        prompt_part = tok.decode([t for t, l in zip(row, lr) if l == -100 and t != 0])
        assert "helpful" in prompt_part and "Transcribe" in prompt_part and "hello" not in prompt_part
    assert att[0].tolist()[0] == 0 and att[1].tolist()[0] == 1                    # left padding of the shorter row
    # row filters (scripts/train.py:274-311)
    bad = [mk("<noise>", 1.0), mk("", 1.0), mk("too long", 30.5), {"audio": {"array": np.array([np.nan, 0.1], np.float32)}, "text": "nan"},
           {"audio": {"array": np.zeros(0, np.float32)}, "text": "empty"}]
    kept = col(bad + [mk("fine", 30.0)])
    assert kept["input_ids"].shape[0] == 1
    with pytest.raises(ValueError, match="No valid audio"):
        col([mk("<unk>", 1.0)])
    mt = MultiTaskDataCollator(tok, fe, 16000, projector=proj)
    b2 = mt([dict(mk("ignored", 1.0), task="sift", sift_response="A man speaks calmly")])
    un = tok.decode([t for t, l in zip(b2["input_ids"][0].tolist(), b2["labels"][0].tolist()) if l != -100])
    assert un == "A man speaks calmly" and "Describe" in tok.decode(b2["input_ids"][0].tolist())


class _StubFeatureExtractor:
    """Stands in for the device log-mel (numerics are covered by the -m gpu tests): shapes + the frame mask contract
    (one frame per 160 samples, TF:models/whisper/feature_extraction_whisper.py:330-339)."""

    def __call__(self, arrays, sampling_rate=16000, padding="longest", return_attention_mask=True, return_tensors="pt"):
        lens = [len(a) for a in arrays]
        T = max(lens) // 160
        mask = torch.zeros((len(arrays), T), dtype=torch.int32)
        for i, n in enumerate(lens):
            mask[i, : n // 160] = 1
        return {"input_features": torch.zeros((len(arrays), 128, T)), "attention_mask": mask}


@pytest.fixture()
def dry_lib_for_features():
    return _StubFeatureExtractor()


def test_split_parameter_groups():
    """scripts/train.py:384-437: language_model.* parameters take decoder_learning_rate / decoder_weight_decay, the rest the
    base LR / projector_weight_decay; norm scales and biases never decay."""
    from tiny_audio_amd.trainer import ASRTrainer, TrainingArguments

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.projector = torch.nn.Linear(4, 4)
            self.projector.norm = torch.nn.LayerNorm(4)
            self.language_model = torch.nn.Module()
            self.language_model.lora_la_qkv = torch.nn.Parameter(torch.zeros(2, 8, 4))
    tr = ASRTrainer(M(), TrainingArguments(learning_rate=1e-3, weight_decay=0.1), decoder_learning_rate=1e-4,
                    decoder_weight_decay=0.01, projector_weight_decay=0.05)
    hp = {n: tr.group_hparams(n, d) for n, d in zip(tr.flat.names, tr.flat.decay)}
    assert hp["projector.weight"] == (1e-3, 0.05) and hp["projector.bias"] == (1e-3, 0.0)
    assert hp["projector.norm.weight"] == (1e-3, 0.0) and hp["projector.norm.bias"] == (1e-3, 0.0)     # an nn.LayerNorm
    assert hp["language_model.lora_la_qkv"] == (1e-4, 0.01)
    tr2 = ASRTrainer(M(), TrainingArguments(learning_rate=1e-3, weight_decay=0.1))
    assert tr2.group_hparams("language_model.lora_la_qkv", True) == (1e-3, 0.1) and tr2.group_hparams("projector.weight", True) == (1e-3, 0.1)


# ----------------------------------------------------------------------------- trainer under world_size 2 (gloo, dry-run kernels)
def _trainer_worker(rank, world, port, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import weights as OW
    from tiny_audio_amd import _lib
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    from tiny_audio_amd.trainer import ASRTrainer, TrainingArguments
    _lib.DRY_RUN = True                      # every kernel launch is a marshalling-only stub: this exercises the host path
    enc = OW.enc_config(hidden=256, ffn=512, layers=1, heads=4)
    lm = OW.lm_config(vocab=1000, hidden=256, ffn=512, layers=2, heads=4, kv_heads=2)
    kw = dict(use_lora=True, freeze_projector=True) if mode == "lora" else (dict(freeze_language_model=False) if mode == "fullft" else {})
    cfg = ASRConfig(audio_config=enc, text_config=lm, projector_hidden_dim=128, audio_token_id=999, **kw)
    torch.manual_seed(0)
    m = ASRModel(cfg, device="cpu", init="random")
    tr = ASRTrainer(m, TrainingArguments(gradient_accumulation_steps=2), decoder_learning_rate=1e-4)
    ids, att, lab, counts = OW.synthetic_tokens(2, [12, 12], 1000, 999, 990, 991, n_text=10, n_suffix=4)
    meta = (torch.zeros(40, dtype=torch.int32), torch.zeros(40, dtype=torch.int64), 22 + rank)      # ranks hold different token counts
    batch = dict(input_ids=torch.from_numpy(ids), input_features=torch.zeros(2, 128, 100), attention_mask=torch.from_numpy(att),
                 labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts), label_meta=meta)
    m.train()
    if mode == "mlp-overlap":
        # deferred update: the collective of step n is launched asynchronously, its clip + AdamW run inside step n+1 right
        # after the frozen encoder's forward (or in flush()); the gradient buffer is cleared only after that update
        tr = ASRTrainer(m, TrainingArguments(gradient_accumulation_steps=1), overlap_allreduce=True)
        tr.training_step(batch)
        s1, pend1 = tr.global_step, tr._pending is not None
        tr.training_step(batch)              # applies step 1 (after the encoder), launches step 2's collective
        s2, c2 = tr.global_step, float(tr._last[1])
        tr.flush()
        s3, c3, pend3 = tr.global_step, float(tr._last[1]), tr._pending is not None
        q.put((rank, s1, int(pend1), s2, c2, s3, c3 + int(pend3)))
        dist.destroy_process_group()
        return
    tr.training_step(batch)                  # micro-step 1: no collective yet
    assert tr.global_step == 0
    before = float(tr.flat.count_slot)
    tr.training_step(batch)                  # micro-step 2: all-reduce of [grads | count | loss], optimizer step
    step, cnt = tr.global_step, float(tr.flat.count_slot)
    # the gradient values of a dry run are uninitialised memory: check the collective itself on stand-in values
    from tiny_audio_amd.trainer import allreduce_flat
    tr.flat.flat_g.zero_(); tr.flat.grads.fill_(float(rank + 1))
    allreduce_flat(tr.flat.flat_g)
    q.put((rank, step, tr.flat.n, cnt, before, float(tr.flat.grads[0]), float(tr.flat.grads[-1])))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["mlp", "lora", "fullft"])
def test_trainer_two_ranks_flat_allreduce(mode):
    """(e) data parallel through the real ASRTrainer: gradient accumulation defers the collective to the last micro-step,
    every trainable tensor (projector / LoRA adapters / the whole LM) travels in ONE flat buffer with the token count."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q, mode)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=300) for _ in procs])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, s0, n0, c0, b0, g0a, g0b), (r1, s1, n1, c1, b1, g1a, g1b) = res
    assert s0 == s1 == 1 and n0 == n1 and c0 == c1
    assert c0 == 2 * 22 + 2 * 23                     # both micro-steps of both ranks: the global label-token count
    assert b0 == 22.0 and b1 == 23.0                 # ... which was still local before the collective
    assert g0a == g1a == g0b == g1b == 3.0           # SUM over ranks of the stand-in gradients (1 + 2), first and last element


def _moe_trainer_ga1_worker(rank, world, port, q):
    """Two ranks, ONE micro-batch per optimizer step: the projector writes its gradients straight into the flat buffer
    (_grad_direct) and must STILL fill the auxiliary shadow."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import weights as OW
    from tiny_audio_amd import _lib
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    from tiny_audio_amd.trainer import ASRTrainer, TrainingArguments
    _lib.DRY_RUN = True
    enc = OW.enc_config(hidden=256, ffn=512, layers=1, heads=4)
    lm = OW.lm_config(vocab=1000, hidden=256, ffn=512, layers=2, heads=4, kv_heads=2)
    cfg = ASRConfig(audio_config=enc, text_config=lm, projector_type="moe", projector_hidden_dim=128, audio_token_id=999)
    torch.manual_seed(0)
    m = ASRModel(cfg, device="cpu", init="random")
    tr = ASRTrainer(m, TrainingArguments(gradient_accumulation_steps=1))
    f = tr.flat
    ids, att, lab, counts = OW.synthetic_tokens(2, [12, 12], 1000, 999, 990, 991, n_text=10, n_suffix=4)
    meta = (torch.zeros(40, dtype=torch.int32), torch.zeros(40, dtype=torch.int64), 22 + rank)
    batch = dict(input_ids=torch.from_numpy(ids), input_features=torch.zeros(2, 128, 100), attention_mask=torch.from_numpy(att),
                 labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts), label_meta=meta)
    m.train()
    _lib.lib().calls.clear()
    tr.training_step(batch)
    calls = list(_lib.lib().calls)
    q.put((rank, tr._aux_direct, m.projector._grad_direct is not None, list(f.shadow_names),
           calls.count("ta_moe_projector_backward_dev"), calls.count("ta_moe_router_aux_grads")))
    dist.destroy_process_group()


def test_moe_trainer_two_ranks_one_micro_batch_fills_the_shadow():
    """world > 1 with gradient_accumulation_steps = 1 -- the direct-gradient branch of the MoE backward returned
    before the shadow fill, so `g += (N - 1) * shadow` added zeros and the optimizer's division by N left aux at weight 1 / N.
    The branch now calls ta_moe_router_aux_grads into the shadow segment (the values are checked on the GPU:
    tests/test_gpu_round5.py::test_moe_aux_shadow_values)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_moe_trainer_ga1_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=300) for _ in procs])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, aux_direct, grad_direct, shadow_names, n_bwd, n_shadow in res:
        assert aux_direct is False and grad_direct is True
        assert shadow_names == ["projector.norm.weight", "projector.router.weight"]
        assert n_bwd == 1 and n_shadow == 1


def _moe_trainer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import weights as OW
    from tiny_audio_amd import _lib
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    from tiny_audio_amd.trainer import ASRTrainer, TrainingArguments
    _lib.DRY_RUN = True
    enc = OW.enc_config(hidden=256, ffn=512, layers=1, heads=4)
    lm = OW.lm_config(vocab=1000, hidden=256, ffn=512, layers=2, heads=4, kv_heads=2)
    cfg = ASRConfig(audio_config=enc, text_config=lm, projector_type="moe", projector_hidden_dim=128, audio_token_id=999)
    torch.manual_seed(0)
    m = ASRModel(cfg, device="cpu", init="random")
    calls = []
    real = dist.all_reduce
    dist.all_reduce = lambda t, *a, **k: (calls.append(t.numel()), real(t, *a, **k))[1]
    tr = ASRTrainer(m, TrainingArguments(gradient_accumulation_steps=2))
    f = tr.flat
    ids, att, lab, counts = OW.synthetic_tokens(2, [12, 12], 1000, 999, 990, 991, n_text=10, n_suffix=4)
    meta = (torch.zeros(40, dtype=torch.int32), torch.zeros(40, dtype=torch.int64), 22 + rank)
    batch = dict(input_ids=torch.from_numpy(ids), input_features=torch.zeros(2, 128, 100), attention_mask=torch.from_numpy(att),
                 labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts), label_meta=meta)
    m.train()
    tr.training_step(batch)                  # micro-step 1 (gradient accumulation with an auxiliary loss: no num_items_in_batch needed)
    n_before = len(calls)
    # stand-in values for the second micro-step's result: the arithmetic of the fix-up is what is checked (dry-run kernels compute nothing)
    tr.training_step(batch)                  # micro-step 2: ONE collective over [grads | shadows | count | loss], then the update
    shadow_names = list(f.shadow_names)
    f.flat_g.zero_()
    f.grad_of("projector.router.weight").fill_(1.0); f.shadow("projector.router.weight").fill_(2.0)
    f.count_slot.fill_(5.0)
    tr._apply_update()
    q.put((rank, n_before, len(calls), calls[-1] if calls else 0, f.flat_g.numel(), shadow_names,
           float(f.grad_of("projector.router.weight").flatten()[0]), float(f.grad_of("projector.norm.weight").flatten()[0]),
           int(m.projector._aux_shadow["router.weight"].data_ptr() == f.shadow("projector.router.weight").data_ptr())))
    dist.destroy_process_group()


def test_moe_trainer_two_ranks_one_collective():
    """Round 4: the MoE projector's auxiliary losses no longer cost a second, blocking all-reduce of the token
    count ahead of the backward -- the auxiliary share of d(norm.weight) / d(router.weight) travels as a SHADOW segment of the flat
    buffer and g += (N - 1) * shadow restores its full weight in front of the update; gradient accumulation works without
    num_items_in_batch."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_moe_trainer_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=300) for _ in procs])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, n_before, n_calls, last_numel, flat_numel, shadow_names, g_router, g_norm, aliased in res:
        assert n_before == 0 and n_calls == 1            # no collective in micro-step 1, exactly ONE for the optimizer step
        assert last_numel == flat_numel                  # ... and it is the flat buffer (gradients + shadows + the two slots)
        assert shadow_names == ["projector.norm.weight", "projector.router.weight"] and aliased == 1
        assert g_router == 1.0 + 2.0 * (5.0 - 1.0) and g_norm == 0.0


def test_trainer_two_ranks_deferred_update():
    """overlap_allreduce: same collective, same update order, applied one encoder-forward later (hidden under the next
    step's frozen-encoder pass on the GPU)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q, "mlp-overlap")) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=300) for _ in procs])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, s1, pend1, s2, c2, s3, c3 in res:
        assert (s1, pend1) == (0, 1)         # nothing applied yet, one collective in flight
        assert (s2, c2) == (1, 45.0)         # step 1 applied inside step 2 with the GLOBAL token count 22 + 23
        assert (s3, c3) == (2, 45.0)         # flush() applied step 2; nothing pending


def test_bench_gpus_flag_starts_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher environment re-launches itself under torch.distributed.run with N
    ranks on 127.0.0.1 (the driver's own command line); with too few GPUs it refuses instead of printing an n_gpus: 1 line."""
    import argparse
    import subprocess
    import bench
    seen = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "call", lambda cmd: seen.setdefault("cmd", cmd) and 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    with pytest.raises(SystemExit) as e:
        bench.relaunch_if_needed(argparse.Namespace(gpus=8))
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.relaunch_if_needed(argparse.Namespace(gpus=8))
    assert "only 1 GPU" in str(e.value.code)
    monkeypatch.setenv("WORLD_SIZE", "8")                     # under the driver's launcher: nothing to do
    assert bench.relaunch_if_needed(argparse.Namespace(gpus=8)) is None
    assert bench.relaunch_if_needed(argparse.Namespace(gpus=1)) is None


def test_synthetic_batch_contract():
    """bench.py's batch: the collator's layout (scripts/train.py:324-348) -- prompt | <audio>*n | prompt | transcript |
    <|im_end|> | pad; only transcript + <|im_end|> carry labels; the count is known on the host."""
    from tiny_audio_amd.synthetic import token_batch
    from oracle import weights as OW
    ids, att, lab, counts, n = token_batch(3, 125, 151670, 151669, 151643, 151645, L=192)
    o_ids, o_att, o_lab, o_counts = OW.synthetic_tokens(3, 125, 151670, 151669, 151643, 151645, L=192)
    assert np.array_equal(ids, o_ids) and np.array_equal(att, o_att) and np.array_equal(lab, o_lab) and np.array_equal(counts, o_counts)
    assert n == 3 * 36 and (ids == 151669).sum() == 3 * 125 and (lab[att == 0] == -100).all()
    with pytest.raises(ValueError):
        token_batch(1, 125, 151670, 151669, 151643, 151645, L=100)


def test_torch_library_operators_are_registered():
    """north_star: the composites are PyTorch custom ops -- torch.ops.ta355.* with schemas, fake (meta) kernels and autograd."""
    from tiny_audio_amd import torch_ops
    for name in torch_ops.OPERATORS:
        op = getattr(torch.ops.ta355, name)
        schema = str(op.default._schema)
        assert schema.startswith(f"ta355::{name}("), schema
    s = str(torch.ops.ta355.lm_forward_loss.default._schema)
    assert "Tensor audio" in s and "Tensor[] trainable" in s and "SymInt handle" in s and "Tensor? src_row" in s
    assert "-> (Tensor, Tensor, Tensor, Tensor, Tensor)" in s
    s = str(torch.ops.ta355.mlp_projector.default._schema)
    assert "Tensor x, Tensor w1, Tensor g1, Tensor w2, Tensor g2, SymInt handle" in s
    # shapes without a GPU: the fake kernels
    from torch._subclasses.fake_tensor import FakeTensorMode
    from tiny_audio_amd.asr_config import ASRConfig as Cfg
    from tiny_audio_amd.projectors import MLPAudioProjector
    proj = MLPAudioProjector(Cfg())
    h = torch_ops.register_module(proj)
    assert torch_ops.register_module(proj) == h and torch_ops.module_of(h) is proj
    with FakeTensorMode():
        x = torch.empty(2, 500, 1280, dtype=torch.bfloat16)
        y, xb, tape = torch.ops.ta355.mlp_projector(x, torch.empty(1024, 5120), torch.empty(1024), torch.empty(1024, 1024),
                                                    torch.empty(1024), h)
        # a contiguous bf16 input IS the image the kernels read: the operator returns an empty placeholder for it (no copy)
        assert y.shape == (2, 125, 1024) and y.dtype == torch.float32 and xb.shape == (0,)
        y32, xb32, _ = torch.ops.ta355.mlp_projector(x.float(), torch.empty(1024, 5120), torch.empty(1024), torch.empty(1024, 1024),
                                                     torch.empty(1024), h)
        assert xb32.shape == x.shape and xb32.dtype == torch.bfloat16
    with pytest.raises(Exception):
        torch_ops.module_of(10 ** 9)


def test_model_output_is_trainer_compatible():
    from tiny_audio_amd.asr_modeling import CausalLMOutput
    o = CausalLMOutput(loss=torch.tensor(1.5), logits=None, n_label_tokens=3)
    assert isinstance(o, dict) and o["loss"] is o.loss is o[0] and o.n_label_tokens == 3 and o.logits is None
    with pytest.raises(AttributeError):
        o.nope


def test_hub_snapshot_loaders(tmp_path):
    """The frozen models' weights from local hub snapshots: sharded safetensors, the audio_tower sub-tree of the GLM-ASR
    checkpoint (tiny_audio/asr_modeling.py:221-231), Qwen3 with its tied lm_head dropped and an oversized embedding."""
    import json
    from safetensors.torch import save_file
    from oracle import weights as OW
    from tiny_audio_amd import hub_weights
    enc = OW.enc_config(hidden=128, ffn=256, layers=1, heads=2)
    w = {k: torch.from_numpy(v) for k, v in OW.init_encoder(enc, 0).items()}
    d = tmp_path / "glm"; d.mkdir()
    names = sorted(w)
    a = {"audio_tower." + k: w[k] for k in names[: len(names) // 2]}
    b = {"audio_tower." + k: w[k] for k in names[len(names) // 2:]}
    b["language_model.model.embed_tokens.weight"] = torch.zeros(4, 4)            # the GLM decoder: must be ignored
    b["multi_modal_projector.linear_1.weight"] = torch.zeros(4, 4)
    save_file(a, str(d / "model-00001-of-00002.safetensors")); save_file(b, str(d / "model-00002-of-00002.safetensors"))
    json.dump({"weight_map": {**{k: "model-00001-of-00002.safetensors" for k in a}, **{k: "model-00002-of-00002.safetensors" for k in b}}},
              open(d / "model.safetensors.index.json", "w"))
    sd = hub_weights.encoder_state_dict(str(d))
    assert set(sd) == set(w) and all(torch.equal(sd[k], w[k]) for k in w)
    lm = OW.lm_config(vocab=300, hidden=64, ffn=128, layers=1, heads=2, kv_heads=1)
    wl = {k: torch.from_numpy(v) for k, v in OW.init_lm(lm, 1).items()}
    wl["model.embed_tokens.weight"] = torch.cat([wl["model.embed_tokens.weight"], torch.ones(20, 64)], 0)      # hub padding rows
    wl["lm_head.weight"] = wl["model.embed_tokens.weight"].clone()
    q = tmp_path / "qwen"; q.mkdir()
    save_file(wl, str(q / "model.safetensors"))
    sl = hub_weights.lm_state_dict(str(q))
    assert "lm_head.weight" not in sl and sl["model.embed_tokens.weight"].shape == (320, 64)
    with pytest.raises(KeyError):
        hub_weights.encoder_state_dict(str(q))
    # an UNTIED head (tie_word_embeddings = false, the larger Qwen3 models) must be refused, not loaded as if it were tied
    wl["lm_head.weight"] = wl["lm_head.weight"] + 0.01
    u = tmp_path / "untied"; u.mkdir()
    save_file(wl, str(u / "model.safetensors"))
    with pytest.raises(ValueError, match="not tied"):
        hub_weights.lm_state_dict(str(u))
    wl.pop("lm_head.weight")
    c = tmp_path / "untied_cfg"; c.mkdir()
    save_file(wl, str(c / "model.safetensors")); json.dump({"tie_word_embeddings": False}, open(c / "config.json", "w"))
    with pytest.raises(ValueError, match="not tied"):
        hub_weights.lm_state_dict(str(c))


def test_causal_lm_output_is_model_output_shaped():
    """No None / Python scalars among the items (HF Trainer.prediction_step concatenates every item but "loss"), index and
    slice access, extras as attributes."""
    from tiny_audio_amd.asr_modeling import CausalLMOutput
    o = CausalLMOutput(loss=torch.tensor(1.5), logits=None, nll=torch.ones(3), n_label_tokens=3, aux_loss=None, loss_ce=torch.tensor(1.5))
    assert list(o.keys()) == ["loss"] and o.logits is None and o["loss"] is o[0] is o.loss
    assert o.n_label_tokens == 3 and o.aux_loss is None and float(o.loss_ce) == 1.5 and o.nll.shape == (3,)
    o2 = CausalLMOutput(loss=torch.tensor(2.0), logits=torch.zeros(2, 4, 8), n_label_tokens=5)
    assert tuple(v for k, v in o2.items() if k != "loss")[0] is o2.logits and o2[1:] == (o2.logits,) and o2.to_tuple()[1] is o2.logits
    o3 = CausalLMOutput(logits=torch.zeros(1))
    assert o3.loss is None and o3[0] is o3.logits
    with pytest.raises(AttributeError):
        o3.hidden_states


def test_causal_lm_output_rebuilds_from_a_mapping_as_accelerate_does():
    """under bf16=True accelerate wraps model.forward in convert_to_fp32, whose recursively_apply rebuilds every Mapping
    as type(data)({k: f(v)}) -- ModelOutput.__init__ accepts that; so must this class (it used to put the dict under 'loss')."""
    import copy
    import pickle
    from accelerate.utils import convert_to_fp32, send_to_device
    from tiny_audio_amd.asr_modeling import CausalLMOutput
    o = CausalLMOutput(loss=torch.tensor(1.5), logits=torch.zeros(2, 3, 8, dtype=torch.bfloat16), nll=torch.ones(3), n_label_tokens=3,
                       loss_ce=torch.tensor(1.5))
    r = convert_to_fp32(o)
    assert type(r) is CausalLMOutput and list(r.keys()) == ["loss", "logits"]
    assert torch.is_tensor(r["loss"]) and float(r["loss"]) == 1.5 and r.logits.dtype == torch.float32 and r[0] is r.loss
    assert send_to_device(o, "cpu").logits.shape == (2, 3, 8)
    same = type(o)(o)                                    # a rebuild from the output itself keeps the attribute extras
    assert same.n_label_tokens == 3 and same.nll is o.nll and float(same.loss_ce) == 1.5
    plain = type(o)(dict(o))                             # from a plain dict they cannot travel: None, not an error
    assert plain.loss is o.loss and plain.n_label_tokens is None
    for c in (copy.copy(o), copy.deepcopy(o), pickle.loads(pickle.dumps(o))):
        assert type(c) is CausalLMOutput and c.n_label_tokens == 3 and float(c.loss) == 1.5 and c.logits.shape == (2, 3, 8)
    with pytest.raises(KeyError):
        CausalLMOutput({"loss": torch.tensor(0.0), "hidden_states": 1})


def test_hf_label_names_are_just_labels():
    """HF Trainer derives label_names from the forward signature (every parameter whose name contains "label"); a second such
    parameter made Trainer.predict treat the batches as unlabelled (round 3: label_meta now travels through **kwargs)."""
    from transformers.utils.generic import find_labels
    from tiny_audio_amd.asr_modeling import ASRModel
    assert find_labels(ASRModel) == ["labels"]


def test_register_module_handles_survive_deepcopy():
    """A deep copy carries the original's handle attribute; it must get its own handle instead of rebinding the original's."""
    import copy
    from tiny_audio_amd import torch_ops

    class M:
        pass
    a = M()
    ha = torch_ops.register_module(a)
    b = copy.deepcopy(a)
    assert b.__dict__["_ta355_handle"] == ha
    hb = torch_ops.register_module(b)
    assert hb != ha and torch_ops.module_of(ha) is a and torch_ops.module_of(hb) is b
    assert torch_ops.register_module(a) == ha and torch_ops.register_module(b) == hb          # idempotent
    del a
    import gc; gc.collect()
    with pytest.raises(Exception, match="stale"):
        torch_ops.module_of(ha)


# ----------------------------------------------------------------------------- bench.py: the N > 1 line validates itself (round 3)
def _replica_worker(rank, world, port, q, diverge):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    torch.manual_seed(0)
    p = torch.randn(1000)
    if diverge and rank == 1:
        p[17] += 1e-6                                   # one replica drifts by one ulp-scale step of one weight
    q.put((rank, bench.replica_report(p, 7 + (rank if diverge else 0), 40.0 + rank)))
    dist.destroy_process_group()


@pytest.mark.parametrize("diverge", [False, True])
def test_bench_replica_report_two_ranks(diverge):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_replica_worker, args=(r, world, port, q, diverge)) for r in range(world)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=120) for _ in procs)
    [p.join(60) for p in procs]
    assert res[0] == res[1]                             # every rank holds the same report (it came out of one all-reduce)
    r = res[0]
    assert r["replicas_identical"] is (not diverge)
    assert r["ms_per_step_by_rank"] == {"min": 40.0, "max": 41.0}
    assert r["global_step"] == ({"min": 7, "max": 8} if diverge else {"min": 7, "max": 7})
    if diverge:
        assert r["weight_checksum"]["sum_min"] < r["weight_checksum"]["sum_max"]


def test_bench_parse_rccl_log_and_error_line(capsys):
    import io
    import json
    import bench
    log = """host:1:1 [0] NCCL INFO RCCL version 2.22.3+hip6.4 HEAD:abc
host:1:1 [0] NCCL INFO === System : maxBw 48.0 totalBw 336.0 ===  GPU/0 -XGMI-> GPU/1
host:1:9 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC
host:1:9 [0] NCCL INFO Channel 01/0 : 0[0] -> 1[1] via P2P/IPC
host:1:9 [0] NCCL INFO Connected all rings
host:1:9 [0] NCCL INFO 16 coll channels, 16 collnet channels, 0 nvls channels, 16 p2p channels
"""
    r = bench.parse_rccl_log(log)
    assert r["via_p2p"] == 2 and r["via_shm"] == 0 and r["channels"] == 16 and r["transport"] == "p2p (xGMI)"
    assert "RCCL version 2.22.3" in r["version_line"]
    assert bench.parse_rccl_log("x via SHM/direct\ny via P2P/IPC\n")["transport"] == "p2p+shm"
    assert bench.parse_rccl_log("")["transport"].startswith("unknown")
    # any rank's exception -> one JSON line with "error" + a non-zero exit code
    buf = io.StringIO()
    with pytest.raises(SystemExit) as e:
        bench.guarded(lambda: (_ for _ in ()).throw(RuntimeError("boom on this rank")), rank=3, world=8, out=buf)
    assert e.value.code == 1
    rec = json.loads(buf.getvalue())
    assert rec["error"] == "RuntimeError: boom on this rank" and rec["rank"] == 3 and rec["n_gpus"] == 8 and rec["value"] is None
    assert bench.guarded(lambda: 5) == 5


def test_bench_flop_model_counts_moe_and_lora():
    import bench
    mlp = bench.algorithmic_gflop_per_clip(192, 151670, 36, False)
    moe = bench.algorithmic_gflop_per_clip(192, 151670, 36, False, projector="moe")
    lora = bench.algorithmic_gflop_per_clip(192, 151670, 36, False, lora=True)
    assert abs(mlp - 1048.5) < 0.1
    assert 3 * 1.57 - 1.57 < (moe - mlp) < 20                 # 3x the adapter forward (+ its backward) on top of one adapter
    assert abs((lora - mlp) - (3 * 1.94 - 1.84)) < 0.05        # + 5.8 GF of adapter work, - the frozen projector's dW


def test_bench_dry_run_two_ranks_end_to_end():
    """`bench.py --gpus 2 --dry-run`: the whole N > 1 control flow of the script (launcher re-exec on 127.0.0.1, process group,
    rank counting, deferred all-reduce, the synchronous A/B leg, replica report, the JSON line) on CPU ranks under gloo with
    stubbed kernels.  The driver is the first to run this path on real GPUs: it must at least be free of host-side errors."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["dry_run"] is True and rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["config"]["global_batch"] == 4
    assert rec["replicas"]["replicas_identical"] is True and rec["replicas"]["global_step"]["min"] == rec["replicas"]["global_step"]["max"] > 0
    ar = rec["allreduce"]
    assert ar["bytes"] == 4 * ar["elements"] and ar["other_mode"]["mode"] == "synchronous" and ar["other_mode"]["steps"] >= 2
    assert "error" not in rec
