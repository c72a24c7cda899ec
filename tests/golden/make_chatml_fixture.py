"""Writes tests/golden/chatml_collation.json + chatml_tokenizer.json: what transformers itself produces for the chat-ML text
side of the batch contract (VERDICT r05 "Next round" 7).

trl (``DataCollatorForChatML``, scripts/train.py:265,344) is not installable in the build container, but everything it
delegates to IS here: ``PreTrainedTokenizerFast.apply_chat_template`` (Jinja rendering, generation prompt, extra template
variables such as Qwen3's ``enable_thinking``) and the fast tokenizer's ``__call__`` (``add_special_tokens=False``, truncation).
This script builds a small byte-level BPE tokenizer with the ChatML special tokens and a Qwen3-shaped template (generation prompt
``<|im_start|>assistant\\n``, plus the empty ``<think>`` block when ``enable_thinking`` is false), and records

* ``collation``: for several message lists, the token ids transformers gives for the two renderings trl makes (prompt WITH the
  generation prompt, full message without), from which the expected batch follows by trl's published rule (labels masked over
  the prompt's token count, everything left-padded) -- written here with plain list arithmetic, independent of
  ``tiny_audio_amd/collator.py``;
* ``processor``: ``apply_chat_template(..., tokenize=True, enable_thinking=False)`` outputs for the message lists
  ``ASRProcessor.__call__`` builds (tiny_audio/asr_processing.py:89-112), with and without target text / system prompt.

Run in the build container:  python tests/golden/make_chatml_fixture.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

TEMPLATE = (
    "{%- for message in messages %}"
    "{{- '<|im_start|>' + message.role + '\\n' + message.content + '<|im_end|>' + '\\n' }}"
    "{%- endfor %}"
    "{%- if add_generation_prompt %}"
    "{{- '<|im_start|>assistant\\n' }}"
    "{%- if enable_thinking is defined and enable_thinking is false %}"
    "{{- '<think>\\n\\n</think>\\n\\n' }}"
    "{%- endif %}"
    "{%- endif %}"
)
SPECIALS = ["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<audio>"]
CORPUS = [
    "Transcribe the speech to text", "You are a helpful assistant.", "hello world this is a test", "second clip",
    "the quick brown fox jumps over the lazy dog", "system user assistant", "Describe all the information you can hear",
    "a man speaks calmly while a door closes", "forty two percent of the people", "<think>\n\n</think>\n\n",
    "numbers 0 1 2 3 4 5 6 7 8 9 and punctuation , . ! ? ' - :",
]


def build_tokenizer():
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    from transformers import PreTrainedTokenizerFast
    tk = Tokenizer(models.BPE(unk_token=None))
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tk.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=420, special_tokens=SPECIALS, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                                  show_progress=False)
    tk.train_from_iterator(CORPUS * 4, trainer)
    path = os.path.join(HERE, "chatml_tokenizer.json")
    tk.save(path)
    return load_tokenizer(path)


def load_tokenizer(path=None):
    from transformers import PreTrainedTokenizerFast
    tok = PreTrainedTokenizerFast(tokenizer_file=path or os.path.join(HERE, "chatml_tokenizer.json"), pad_token="<|endoftext|>",
                                  eos_token="<|im_end|>", additional_special_tokens=["<|im_start|>", "<audio>"])
    tok.chat_template = TEMPLATE
    tok.padding_side = "right"          # as tiny_audio/asr_modeling.py:335 sets it: the collation must left-pad anyway
    return tok


def messages(n_audio, text, system=None, prompt="Transcribe the speech to text"):
    m = [{"role": "system", "content": system}] if system else []
    m.append({"role": "user", "content": "<audio>" * n_audio + " " + prompt})
    if text is not None:
        m.append({"role": "assistant", "content": text})
    return m


def left_pad(rows, value):
    L = max(len(r) for r in rows)
    return [[value] * (L - len(r)) + list(r) for r in rows]


def main():
    tok = build_tokenizer()
    pad = tok.pad_token_id
    batches = {
        "two_rows_system": ([messages(12, "hello world this is a test", "You are a helpful assistant."),
                             messages(25, "second clip", "You are a helpful assistant.")], 2048),
        "no_system_three_rows": ([messages(3, "the quick brown fox"), messages(1, "a"), messages(7, "forty two percent of the people")], 2048),
        "truncated": ([messages(40, "the quick brown fox jumps over the lazy dog"), messages(2, "hello")], 48),
    }
    collation = {}
    for name, (rows, max_length) in batches.items():
        ids, att, lab, p_ids, p_att = [], [], [], [], []
        for msgs in rows:
            prompt_text = tok.apply_chat_template(msgs[:-1], tokenize=False, add_generation_prompt=True)
            full_text = tok.apply_chat_template(msgs, tokenize=False, add_generation_prompt=False)
            full = tok(full_text, truncation=True, max_length=max_length, padding=False, return_tensors=None, add_special_tokens=False)
            prm = tok(prompt_text, truncation=True, max_length=len(full["input_ids"]), padding=False, return_tensors=None,
                      add_special_tokens=False)
            n = len(prm["input_ids"])
            ids.append(full["input_ids"]); att.append(full["attention_mask"])
            p_ids.append(prm["input_ids"]); p_att.append(prm["attention_mask"])
            lab.append([-100] * n + full["input_ids"][n:])
        collation[name] = {"messages": rows, "max_length": max_length,
                           "input_ids": left_pad(ids, pad), "attention_mask": left_pad(att, 0), "labels": left_pad(lab, -100),
                           "prompts": left_pad(p_ids, pad), "prompt_attention_mask": left_pad(p_att, 0)}
    processor = []
    for n_audio, text, system in ((12, None, None), (12, None, "You are a helpful assistant."), (5, "hello world", None),
                                  (0, "hello", None), (0, None, None)):
        user = ("<audio>" * n_audio + " Transcribe the speech to text") if n_audio else "Transcribe the speech to text"
        m = [{"role": "system", "content": system}] if system else []
        m.append({"role": "user", "content": user})
        if text is not None:
            m.append({"role": "assistant", "content": text})
        out = tok.apply_chat_template(m, tokenize=True, add_generation_prompt=text is None, return_tensors="pt", enable_thinking=False)
        out = out["input_ids"] if not hasattr(out, "tolist") else out
        processor.append({"num_audio_tokens": n_audio, "text": text, "system_prompt": system, "input_ids": out.reshape(-1).tolist()})
    rec = {"template": TEMPLATE, "pad_token_id": pad, "audio_token_id": tok.convert_tokens_to_ids("<audio>"),
           "collation": collation, "processor": processor}
    with open(os.path.join(HERE, "chatml_collation.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("wrote chatml_collation.json:", {k: len(v["input_ids"][0]) for k, v in collation.items()}, "vocab", len(tok))


if __name__ == "__main__":
    main()
