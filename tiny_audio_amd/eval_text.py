"""Text-side helpers of the generate -> WER loop.  Host-only string work; the GPU
never sees it.

* ``postprocess_tokens``   ASRPipeline.postprocess, tiny_audio/asr_pipeline.py:232-268: drop eos ids, decode, strip
                           ``<think>...</think>`` blocks, truncate trailing repetitions.
* ``truncate_repetitions`` behaviour of ``_truncate_repetitions`` (tiny_audio/asr_pipeline.py:271-330) restated as
                           explicit scans instead of backtracking regular expressions; pinned on outputs of the
                           reference function (tests/golden/text_post.json).
* ``word_error_rate``      what ``jiwer.wer(refs, hyps)`` returns for the evaluator (scripts/eval/evaluators/base.py:
                           100-150): total word-level edit distance / total reference words.  The Whisper English
                           normaliser the reference applies first needs tokenizer assets that are not available
                           offline, so it is an injectable ``normalize`` callable here.
"""
from __future__ import annotations

import re
from typing import Callable, Iterable, Optional, Sequence

_WORD = re.compile(r"\w+")


def strip_think(text: str) -> str:
    """Remove every ``<think> ... </think>`` block (and the whitespace after it); Qwen3 emits them even with
    thinking disabled (asr_pipeline.py:263-265)."""
    out, i = [], 0
    while True:
        a = text.find("<think>", i)
        b = text.find("</think>", a + 7) if a >= 0 else -1
        if a < 0 or b < 0:
            out.append(text[i:])
            break
        out.append(text[i:a])
        i = b + len("</think>")
        while i < len(text) and text[i].isspace():
            i += 1
    return "".join(out).strip()


def _collapse_trailing_char_run(text: str, n: int) -> str:
    body = text[:-1] if text.endswith("\n") else text          # '$' also matches just before one final newline
    if not body or body[-1] == "\n":
        return text
    c, k = body[-1], 1
    while k < len(body) and body[-1 - k] == c:
        k += 1
    return text if k < n else body[: len(body) - k + 1] + text[len(body):]


def _collapse_trailing_word_run(text: str, n: int) -> Optional[str]:
    """One application of: a \\w+ word repeated >= n times (whitespace separated, case-insensitive) at the very end."""
    tokens = [(m.start(), m.end()) for m in re.finditer(r"\S+", text)]
    if not tokens:
        return None
    s, e = tokens[-1]
    w = text[s:e]
    if not _WORD.fullmatch(w):
        return None
    lw = w.lower()
    k = 1
    while k < len(tokens) and text[tokens[-1 - k][0]:tokens[-1 - k][1]].lower() == lw:
        k += 1
    first = tokens[-k][0]
    if k < len(tokens):                                          # the run may begin INSIDE the previous token ("x-the the the")
        ps, pe = tokens[-1 - k]
        prev = text[ps:pe]
        if len(prev) > len(w) and prev.lower().endswith(lw) and not _WORD.fullmatch(prev[-len(w) - 1]):
            k, first = k + 1, pe - len(w)
    if k < n:
        return None
    return text[:first] + text[first:first + len(w)]


def _collapse_trailing_phrase(text: str, words: Sequence[str], phrase_len: int, n: int) -> Optional[str]:
    phrase = " ".join(words[-phrase_len:]).lower()
    end = len(text.rstrip())
    low = text.lower()
    starts = []                                                  # start offsets of consecutive occurrences, last first
    pos = end
    while pos - len(phrase) >= 0 and low[pos - len(phrase):pos] == phrase:
        starts.append(pos - len(phrase))
        q = pos - len(phrase)
        gap = q
        while gap > 0 and text[gap - 1].isspace():
            gap -= 1
        if gap == q:                                             # no whitespace before this occurrence: chain ends here
            break
        pos = gap
    # the chain may only start at the beginning of the text or right after whitespace, on the same line as it
    while starts and not (starts[-1] == 0 or text[starts[-1] - 1].isspace()):
        starts.pop()
    if len(starts) < n:
        return None
    first = starts[-1]
    if "\n" in text[:max(first - 1, 0)]:                         # the reference's lazy '.' prefix cannot cross a line break
        return None
    return (text[:first] + text[first:first + len(phrase)]).strip()


def truncate_repetitions(text: str, min_repeats: int = 3) -> str:
    """Trailing repetitions removed: characters ("444444" -> "4"), words ("the the the" -> "the"), phrases of 2..20
    words ("i am sorry i am sorry i am sorry" -> "i am sorry")."""
    if not text:
        return text
    n = int(min_repeats)
    text = _collapse_trailing_char_run(text, n)
    while True:
        nxt = _collapse_trailing_word_run(text, n)
        if nxt is None:
            break
        text = nxt
    words = text.split()
    if len(words) < 2 * n:
        return text
    tail = words[-2 * n:]
    if len(set(tail)) == len(tail):                              # all distinct: no phrase can repeat n times
        return text
    for phrase_len in range(2, min(21, len(words) // n + 1)):
        out = _collapse_trailing_phrase(text, words, phrase_len, n)
        if out is not None:
            return out
    return text


def postprocess_tokens(tokens: Iterable[int], eos_ids: Iterable[int], decode: Callable[[list], str]) -> str:
    eos = set(int(e) for e in eos_ids)
    kept = [int(t) for t in tokens if int(t) not in eos]
    text = decode(kept).strip()
    if "<think>" in text:
        text = strip_think(text)
    return truncate_repetitions(text)


def _edit_distance(ref: Sequence[str], hyp: Sequence[str]) -> int:
    prev = list(range(len(hyp) + 1))
    for i, r in enumerate(ref, 1):
        cur = [i] + [0] * len(hyp)
        for j, h in enumerate(hyp, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (r != h))
        prev = cur
    return prev[-1]


def word_error_rate(references, hypotheses, normalize: Optional[Callable[[str], str]] = None) -> float:
    """Corpus WER: sum of word edit distances / sum of reference lengths (strings or lists of strings)."""
    if isinstance(references, str):
        references, hypotheses = [references], [hypotheses]
    if len(references) != len(hypotheses):
        raise ValueError("references and hypotheses differ in length")
    norm = normalize or (lambda s: s)
    edits = total = 0
    for r, h in zip(references, hypotheses):
        rw, hw = norm(r).split(), norm(h).split()
        edits += _edit_distance(rw, hw)
        total += len(rw)
    if total == 0:
        raise ValueError("empty reference")
    return edits / total
