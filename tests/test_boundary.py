"""-m "not gpu": the drop-in seams around the hot path that are host code -- ASRProcessor / ASRModel.get_processor (north_star:
"ASRModel/ASRProcessor ... stay drop-in"), the chat-ML text collation, and INTEGRATION.md's ctypes
stub checked against the header and the built library so that the document cannot drift from the ABI again.

The ASRProcessor tests restate the assertions of the reference's own tests/test_asr_processing.py:76-233 (mocked feature
extractor / tokenizer / projector); the fixture tests compare with what transformers itself produced for a real fast tokenizer
with a Qwen3-shaped ChatML template (tests/golden/make_chatml_fixture.py).
"""
import ctypes as C
import json
import os
import re
from unittest.mock import MagicMock

import pytest
import torch

from tiny_audio_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


# ----------------------------------------------------------------------------- ASRProcessor: the reference's own assertions
def _mock_parts(mask=None, chat_ids=((1, 2, 3, 4, 5),), out_len=50):
    fe = MagicMock()
    fe.sampling_rate = 16000
    fe.return_value = {"input_features": torch.randn(1, 80, 100), "attention_mask": torch.ones(1, 100) if mask is None else mask}
    tok = MagicMock()
    tok.convert_tokens_to_ids.return_value = 12345
    tok.apply_chat_template.return_value = torch.tensor(chat_ids)
    proj = MagicMock()
    if callable(out_len):
        proj.get_output_length.side_effect = out_len
    else:
        proj.get_output_length.return_value = out_len
    return fe, tok, proj


def test_processor_constants_and_init():
    from tiny_audio_amd.asr_config import DEFAULT_ENCODER_CONV_LAYERS
    from tiny_audio_amd.asr_processing import ASRProcessor
    assert ASRProcessor.AUDIO_TOKEN == "<audio>" and ASRProcessor.TRANSCRIBE_PROMPT == "Transcribe the speech to text"
    assert DEFAULT_ENCODER_CONV_LAYERS == [(1, 3, 1), (1, 3, 2)]
    fe, tok, proj = _mock_parts()
    p = ASRProcessor(fe, tok, proj)
    assert p.feature_extractor is fe and p.tokenizer is tok and p.projector is proj and p.audio_token_id == 12345
    assert p.encoder_conv_layers == DEFAULT_ENCODER_CONV_LAYERS
    custom = [(0, 3, 2), (0, 3, 2)]
    assert ASRProcessor(fe, tok, proj, encoder_conv_layers=custom).encoder_conv_layers == custom
    assert ASRProcessor(feature_extractor=fe, tokenizer=tok).projector is None                  # projector is optional, as in the reference
    assert p._compute_encoder_output_length(100) == 50 and p._compute_encoder_output_length(1) == 1
    assert p._compute_encoder_output_length(3000) == 1500


def test_processor_call_contract():
    """tests/test_asr_processing.py:120-190 of the reference: keys, the user turn, the system turn, text-only, the assistant turn."""
    from tiny_audio_amd.asr_processing import ASRProcessor
    p = ASRProcessor(*_mock_parts())
    r = p(audio=torch.randn(16000))
    assert set(r) == {"input_features", "audio_attention_mask", "input_ids", "attention_mask"}
    assert r["input_ids"].tolist() == [[1, 2, 3, 4, 5]] and r["attention_mask"].tolist() == [[1] * 5]
    args, kw = p.tokenizer.apply_chat_template.call_args
    assert kw["add_generation_prompt"] is True and kw["enable_thinking"] is False and kw["tokenize"] is True
    assert kw["return_tensors"] == "pt"
    user = [m for m in args[0] if m["role"] == "user"][0]
    assert user["content"] == "<audio>" * 50 + " Transcribe the speech to text"
    fe_args, fe_kw = p.feature_extractor.call_args
    assert fe_kw["sampling_rate"] == 16000 and fe_kw["return_attention_mask"] is True and fe_kw["return_tensors"] == "pt"

    p(audio=torch.randn(16000), system_prompt="You are helpful.")
    msgs = p.tokenizer.apply_chat_template.call_args[0][0]
    assert [m["role"] for m in msgs] == ["system", "user"] and msgs[0]["content"] == "You are helpful."

    r = p(text="hello")                                                        # no audio: no features, no placeholders
    assert "input_ids" in r and "input_features" not in r and "audio_attention_mask" not in r
    args, kw = p.tokenizer.apply_chat_template.call_args
    assert kw["add_generation_prompt"] is False
    assert args[0] == [{"role": "user", "content": "Transcribe the speech to text"}, {"role": "assistant", "content": "hello"}]

    p(audio=torch.randn(16000), text="hello world")
    msgs = p.tokenizer.apply_chat_template.call_args[0][0]
    assert [m for m in msgs if m["role"] == "assistant"] == [{"role": "assistant", "content": "hello world"}]


def test_processor_counts_placeholders_from_the_attention_mask():
    """tests/test_asr_processing.py:193-233: 80 valid of 100 frames -> conv 80 -> 40 -> projector 40 // 4 = 10 placeholders."""
    from tiny_audio_amd.asr_processing import ASRProcessor
    mask = torch.cat([torch.ones(1, 80), torch.zeros(1, 20)], dim=1)
    p = ASRProcessor(*_mock_parts(mask=mask, chat_ids=((1, 2, 3),), out_len=lambda x: x // 4))
    p(audio=torch.randn(16000))
    user = [m for m in p.tokenizer.apply_chat_template.call_args[0][0] if m["role"] == "user"][0]
    assert user["content"].count("<audio>") == 10
    # tokenizers that hand back a 1-D tensor or a BatchEncoding-like mapping (asr_processing.py:114-123)
    p.tokenizer.apply_chat_template.return_value = torch.tensor([7, 8, 9])
    assert p(audio=torch.randn(16000))["input_ids"].shape == (1, 3)
    p.tokenizer.apply_chat_template.return_value = {"input_ids": torch.tensor([[7, 8]]), "attention_mask": torch.ones(1, 2)}
    assert p(audio=torch.randn(16000))["input_ids"].tolist() == [[7, 8]]


# ----------------------------------------------------------------------------- against transformers' own outputs (fixture)
@pytest.fixture(scope="module")
def chatml():
    pytest.importorskip("transformers")
    import sys
    sys.path.insert(0, GOLDEN)
    try:
        from make_chatml_fixture import load_tokenizer
    finally:
        sys.path.remove(GOLDEN)
    return load_tokenizer(), json.load(open(os.path.join(GOLDEN, "chatml_collation.json")))


def test_processor_matches_transformers_chat_template(chatml):
    """ASRProcessor.__call__ over a REAL fast tokenizer whose template has Qwen3's enable_thinking switch: ids identical to what
    transformers' apply_chat_template gave for the same request (generation prompt + empty think block iff no target text)."""
    from tiny_audio_amd.asr_processing import ASRProcessor
    tok, g = chatml
    assert tok.convert_tokens_to_ids("<audio>") == g["audio_token_id"]
    for case in g["processor"]:
        n = case["num_audio_tokens"]
        fe = MagicMock(); fe.sampling_rate = 16000
        fe.return_value = {"input_features": torch.zeros(1, 128, 100), "attention_mask": torch.ones(1, 100, dtype=torch.int32)}
        proj = MagicMock(); proj.get_output_length.return_value = n
        p = ASRProcessor(fe, tok, proj)
        r = p(audio=[torch.zeros(16000).numpy()] if n else None, text=case["text"], system_prompt=case["system_prompt"])
        assert r["input_ids"].tolist() == [case["input_ids"]], case
        assert r["attention_mask"].tolist() == [[1] * len(case["input_ids"])]
        assert (r["input_ids"] == p.audio_token_id).sum().item() == n
        think = tok.decode(r["input_ids"][0]).endswith("<think>\n\n</think>\n\n")
        assert think == (case["text"] is None)                                   # the generation prompt carries the closed think block


def test_chatml_collation_matches_transformers_outputs(chatml):
    """Row a12: input_ids / attention_mask / labels / prompts / prompt_attention_mask equal to the batch built from transformers'
    renderings by trl's rule (prompt-token-count split, left padding, truncation to max_length)."""
    from tiny_audio_amd.collator import ChatMLTextCollator
    tok, g = chatml
    assert tok.padding_side == "right"                                           # ignored: the collation left-pads by itself
    for name, c in g["collation"].items():
        out = ChatMLTextCollator(tok, max_length=c["max_length"])([{"messages": m} for m in c["messages"]])
        assert set(out) == {"input_ids", "attention_mask", "labels", "prompts", "prompt_attention_mask"}, name
        for k in out:
            assert out[k].dtype == torch.int64 and out[k].tolist() == c[k], (name, k)
        ids, lab, att = out["input_ids"], out["labels"], out["attention_mask"]
        assert ((lab == -100) | (lab == ids)).all() and (lab[att == 0] == -100).all()
        assert (lab[ids == g["audio_token_id"]] == -100).all()                   # placeholders are never targets
    c = g["collation"]["two_rows_system"]
    first = [t for t, l in zip(c["input_ids"][0], c["labels"][0]) if l != -100]
    assert tok.decode(first) == "hello world this is a test<|im_end|>\n"          # the answer, its stop token and the template's newline


def test_data_collator_emits_trl_keys(chatml):
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.collator import DataCollator
    from tiny_audio_amd.projectors import MLPAudioProjector
    import numpy as np
    tok, g = chatml

    class FE:
        def __call__(self, arrays, sampling_rate=16000, **_):
            T = max(len(a) for a in arrays) // 160
            m = torch.zeros((len(arrays), T), dtype=torch.int32)
            for i, a in enumerate(arrays):
                m[i, : len(a) // 160] = 1
            return {"input_features": torch.zeros((len(arrays), 128, T)), "attention_mask": m}
    col = DataCollator(tok, FE(), 16000, system_prompt="You are a helpful assistant.", projector=MLPAudioProjector(ASRConfig()))
    rng = np.random.RandomState(0)
    mk = lambda text, sec: {"audio": {"array": (0.1 * rng.standard_normal(int(sec * 16000))).astype(np.float32)}, "text": text}
    b = col([mk("Hello World this is a TEST <comma>", 1.0), mk("second clip", 2.0)])
    assert set(b) == {"input_ids", "attention_mask", "labels", "prompts", "prompt_attention_mask", "input_features",
                      "audio_attention_mask", "audio_token_counts"}                # scripts/train.py:344-348 + trl's two extra keys
    c = g["collation"]["two_rows_system"]                                         # the same two rows, 12 / 25 placeholders
    assert b["audio_token_counts"].tolist() == [12, 25]
    for k in ("input_ids", "attention_mask", "labels", "prompts", "prompt_attention_mask"):
        assert b[k].tolist() == c[k], k


# ----------------------------------------------------------------------------- ASRModel.get_processor and the HF surface
def test_get_processor_and_hf_surface(chatml):
    from oracle import weights as OW
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    from tiny_audio_amd.asr_processing import ASRProcessor, LogMelFeatureExtractor
    tok, g = chatml
    enc, lm = OW.enc_config(hidden=256, ffn=512, layers=1, heads=4), OW.lm_config(vocab=len(tok), hidden=256, ffn=512, layers=1, heads=4, kv_heads=2)
    cfg = ASRConfig(audio_config=enc, text_config=lm, projector_hidden_dim=128)
    m = ASRModel(cfg, device="cpu", init="none")
    with pytest.raises(ValueError, match="tokenizer"):
        m.get_processor()
    m = ASRModel(cfg, device="cpu", init="none", tokenizer=tok)
    assert m.audio_token_id == g["audio_token_id"]                               # read from the tokenizer (asr_modeling.py:160-171)
    assert isinstance(m.feature_extractor, LogMelFeatureExtractor) and m.feature_extractor.feature_size == 128
    p = m.get_processor()
    assert isinstance(p, ASRProcessor) and p.feature_extractor is m.feature_extractor and p.tokenizer is tok
    assert p.projector is m.projector and p.encoder_conv_layers == cfg.encoder_conv_layers
    assert p.audio_token_counts(torch.ones(2, 100, dtype=torch.int32)).tolist() == [12, 12]
    # the PreTrainedModel surface of tiny_audio/asr_modeling.py:359-382
    m.language_model.load_state_dict_hf(OW.init_lm(lm, 1))
    emb, head = m.get_input_embeddings(), m.get_output_embeddings()
    assert emb.weight.shape == (len(tok), 256) and head.weight.data_ptr() == emb.weight.data_ptr() and not emb.weight.requires_grad
    assert torch.equal(emb(torch.tensor([3])), m.language_model.get_input_embeddings_weight()[3:4])
    new = torch.nn.Embedding(len(tok), 256)
    m.set_input_embeddings(new)
    assert torch.equal(m.get_input_embeddings().weight, new.weight.detach())
    assert m.language_model._w.embed_f32 == m.language_model.get_input_embeddings_weight().data_ptr()
    m._set_gradient_checkpointing(True); m.gradient_checkpointing_disable()
    assert m.gradient_checkpointing is False
    with pytest.raises(NotImplementedError):
        m.prepare_inputs_for_generation(torch.zeros(1, 3, dtype=torch.long))


def test_stream_mode_is_per_model_not_process_wide():
    """ABI 4: res_f32 / dx_f32 are fields of each model's weights handle; nothing named *stream_modes* is exported any more."""
    from oracle import weights as OW
    from tiny_audio_amd import ops
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    protos = _lib.parse_header()
    assert "ta_set_stream_modes" not in protos and "ta_get_stream_modes" not in protos
    assert not hasattr(ops, "set_stream_modes") and not hasattr(ops, "get_stream_modes")
    enc, lm = OW.enc_config(hidden=256, ffn=512, layers=1, heads=4), OW.lm_config(vocab=1000, hidden=256, ffn=512, layers=1, heads=4, kv_heads=2)
    models = {}
    for dt in ("bfloat16", "float32"):
        m = ASRModel(ASRConfig(audio_config=enc, text_config=lm, projector_hidden_dim=128, model_dtype=dt), device="cpu", init="none")
        m.audio_tower.load_state_dict_hf(OW.init_encoder(enc, 0))
        m.language_model.load_state_dict_hf(OW.init_lm(lm, 1))
        m._apply_stream_modes()
        models[dt] = m
    a, b = models["bfloat16"], models["float32"]
    assert (a.audio_tower._w.res_f32, a.language_model._w.res_f32, a.language_model._w.dx_f32) == (0, 0, 0)
    assert (b.audio_tower._w.res_f32, b.language_model._w.res_f32, b.language_model._w.dx_f32) == (1, 1, 1)
    b.config.model_dtype = "bfloat16"; b._apply_stream_modes()                   # follows the config at the next forward ...
    assert (b.audio_tower._w.res_f32, b.language_model._w.res_f32) == (0, 0)
    assert (a.audio_tower._w.res_f32, a.language_model._w.res_f32) == (0, 0)     # ... of THAT model only


# ----------------------------------------------------------------------------- INTEGRATION.md section B is executable
def _integration_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## B. Bind the C ABI directly"):]
    return re.findall(r"```python\n(.*?)```", sec, flags=re.S)


# op -> the size queries a stub that CALLS it must use for its scratch / workspace / tape arguments
_SIZE_QUERIES = {"ta_logmel_f32": ["ta_logmel_scratch_floats"], "ta_encoder_forward": ["ta_encoder_workspace_bytes"],
                 "ta_mlp_projector_forward": ["ta_mlp_tape_bytes"], "ta_mlp_projector_backward": ["ta_mlp_bwd_workspace_bytes"],
                 "ta_moe_projector_forward": ["ta_moe_tape_bytes"], "ta_moe_projector_backward": ["ta_moe_bwd_workspace_bytes"],
                 "ta_lm_forward_loss": ["ta_lm_tape_bytes", "ta_lm_workspace_bytes"], "ta_lm_backward": ["ta_lm_workspace_bytes"],
                 "ta_lm_prefill": ["ta_lm_prefill_workspace_bytes"], "ta_lm_decode_step": ["ta_lm_decode_workspace_bytes"]}


def test_integration_md_stub_matches_the_header_and_the_library():
    blocks = _integration_blocks()
    assert blocks, "INTEGRATION.md section B lost its python block"
    protos = _lib.parse_header()
    assert set(_SIZE_QUERIES) <= set(protos) and all(q in protos for qs in _SIZE_QUERIES.values() for q in qs)
    _lib.build()
    ns = {}
    for src in blocks:
        src = src.replace('C.CDLL("libta355.so")', f"C.CDLL({_lib.SO_PATH!r})")
        exec(compile(src, "INTEGRATION.md", "exec"), ns)           # runs the version assertion against the built library
        # every prototype the document binds has the header's arity and return width
        bound = re.findall(r"_lib\.(ta_\w+)\.argtypes\s*=", src)
        assert len(bound) >= 10
        for name in bound:
            assert name in protos, f"INTEGRATION.md binds {name}, which include/ta355.h does not declare"
            ret, argtypes = protos[name]
            fn = getattr(ns["_lib"], name)
            assert len(fn.argtypes) == len(argtypes), f"{name}: INTEGRATION.md lists {len(fn.argtypes)} arguments, the header {len(argtypes)}"
            for i, (doc_t, hdr_t) in enumerate(zip(fn.argtypes, argtypes)):
                assert C.sizeof(doc_t) == C.sizeof(hdr_t) and (doc_t is C.c_float) == (hdr_t is C.c_float), (name, i, doc_t, hdr_t)
            assert fn.restype is not None and C.sizeof(fn.restype) == C.sizeof(ret), name
        # a stub that calls an op sizes that op's buffers with the library's query, never by formula
        code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
        for op, queries in _SIZE_QUERIES.items():
            if re.search(rf"_lib\.{op}\(", code):
                for q in queries:
                    assert re.search(rf"_lib\.{q}\(", code), f"INTEGRATION.md calls {op} without sizing its buffer through {q}()"
        assert re.search(r"ta_version\(\)\s*==\s*(\d+)", code).group(1) == str(_lib.lib().ta_version())
    # and the two stubs that are real code run end to end through the argument marshalling (no GPU: nothing is launched)
    assert callable(ns["logmel"]) and callable(ns["encoder_forward"])


def test_integration_md_stubs_marshal(monkeypatch):
    """Drive the document's logmel() / encoder_forward() with CPU tensors against a library whose kernel-launching entry points are
    replaced by argument-checking no-ops (the _DryLib of the plumbing tests): a wrong argument order or a missing buffer fails here."""
    from oracle import weights as OW
    from tiny_audio_amd.asr_config import EncoderConfig
    from tiny_audio_amd.encoder import GlmAsrEncoderMI355X
    ns = {}
    src = _integration_blocks()[0].replace('C.CDLL("libta355.so")', f"C.CDLL({_lib.SO_PATH!r})")
    exec(compile(src, "INTEGRATION.md", "exec"), ns)
    real = ns["_lib"]
    dry = _lib._DryLib(real)
    ns["_lib"] = dry                                                              # functions look _lib up in their globals
    tables = type("T", (), {})()
    tables.mel_ranges, tables.mel, tables.dft, tables.win = None, torch.zeros(201, 128), torch.zeros(400, 402), torch.zeros(400)
    wav, lens = torch.zeros(2, 16000), torch.tensor([16000, 12000])
    ns["logmel"](wav, lens, tables, torch.empty(2, 128, 100), torch.empty(2, 100, dtype=torch.int32))
    assert dry.calls == ["ta_logmel_mel_ranges", "ta_logmel_f32"] and tables.mel_ranges.numel() == 256
    enc_cfg = OW.enc_config(hidden=256, ffn=512, layers=1, heads=4)
    enc = GlmAsrEncoderMI355X(EncoderConfig(**enc_cfg) if not isinstance(enc_cfg, EncoderConfig) else enc_cfg, device="cpu")
    enc.load_state_dict_hf(OW.init_encoder(enc_cfg, 0))
    ns["encoder_forward"](enc._w, torch.zeros(2, 128, 100), torch.empty(2, 50, 256, dtype=torch.bfloat16))
    assert dry.calls[-1] == "ta_encoder_forward"
