"""CPU ORACLE (test infrastructure): restatement of internal/text/prepare.go and chunk.go.

PrepareText (prepare.go:66-100), splitSentences (chunk.go:49-73), PrepareChunks
(prepare.go:105-184), EstimateMaxFrames (:38-48), FramesAfterEOS (:53-59).
Go's unicode.IsSpace / IsLetter / IsDigit / ToUpper map onto str.isspace / isalpha /
isdigit / upper for the inputs the reference tests use.
"""
from __future__ import annotations

import math

DEFAULT_MIMI_FRAME_RATE = 12.5


def estimate_max_frames(token_count: int, frame_rate: float = DEFAULT_MIMI_FRAME_RATE) -> int:
    if token_count < 0:
        token_count = 0
    if frame_rate <= 0 or math.isnan(frame_rate) or math.isinf(frame_rate):
        frame_rate = DEFAULT_MIMI_FRAME_RATE
    return int(math.ceil((token_count / 3.0 + 2.0) * frame_rate))


def frames_after_eos(num_words: int) -> int:
    return 5 if num_words <= 4 else 3


def split_words(s: str) -> list[str]:
    return s.split()


def prepare_text(inp: str) -> str:
    s = inp.replace("\r\n", " ").replace("\r", " ").replace("\n", " ")
    while "  " in s:
        s = s.replace("  ", " ")
    s = s.strip()
    if s:
        up = s[0].upper()   # unicode.ToUpper is the SIMPLE case mapping (one rune): 'ß' has none, str.upper() would give 'SS'
        s = (up if len(up) == 1 else s[0]) + s[1:]
    if s:
        last = s[-1]
        if last.isalpha() or last.isdigit():
            s += "."
    if len(split_words(s)) < 5:
        s = "        " + s
    return s


def split_sentences(text: str) -> list[str]:
    out, start = [], 0
    for i, r in enumerate(text):
        if r in ".!?":
            s = text[start:i + 1].strip()
            if s:
                out.append(s)
            start = i + 1
    if start < len(text):
        s = text[start:].strip()
        if s:
            out.append(s)
    return out


def prepare_chunks(inp: str, encode, max_tokens: int = 50) -> list[dict]:
    if not inp.strip():
        raise ValueError("input text is empty")
    sentences = split_sentences(inp) or [inp]
    chunks: list[dict] = []
    pending: list[str] = []

    def flush():
        if not pending:
            return
        joined = " ".join(pending)
        prepared = prepare_text(joined)
        ids = list(encode(prepared))
        chunks.append({"text": prepared, "token_ids": ids, "num_tokens": len(ids),
                       "num_words": len(split_words(joined))})
        pending.clear()

    for sent in sentences:
        ids = list(encode(prepare_text(sent)))
        if pending:
            pending_tokens = len(list(encode(prepare_text(" ".join(pending + [sent])))))
        else:
            pending_tokens = len(ids)
        if pending and pending_tokens > max_tokens:
            flush()
        pending.append(sent)
    flush()
    return chunks
