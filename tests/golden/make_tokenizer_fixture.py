"""Generates tests/golden/tiny_unigram.model and tests/golden/tokenizer_ids.json.

The reference's tokenizer tests need the real `tokenizer.model` (internal/tokenizer/tokenizer_test.go:10-38: skipped without it),
which is not available offline.  This script trains a small unigram model with the SentencePiece library itself (Python
sentencepiece, the upstream implementation the reference's pinned id vectors come from) and records what THAT library encodes
for a list of probe strings.  Settings mirror what the reference's pinned vectors imply about the real model: a dummy prefix,
whitespace kept as it is ("        hello" -> eight U+2581 pieces, tokenizer_test.go:122-141), plain NFKC.
    python tests/golden/make_tokenizer_fixture.py"""
import io
import json
import os

import sentencepiece as spm

HERE = os.path.dirname(os.path.abspath(__file__))
CORPUS = [
    "Hello world.", "Test sentence.", "The quick brown fox jumps over the lazy dog.", "She sells sea shells by the sea shore.",
    "How much wood would a woodchuck chuck if a woodchuck could chuck wood?", "Peter Piper picked a peck of pickled peppers.",
    "It was the best of times, it was the worst of times.", "To be, or not to be, that is the question.",
    "All happy families are alike; each unhappy family is unhappy in its own way.", "Call me Ishmael.",
    "In the beginning the Universe was created.", "This has made a lot of people very angry and been widely regarded as a bad move.",
    "A text to speech model turns written words into natural sounding audio.", "Numbers like 1, 2, 3 and 42 appear too.",
    "Prices: $9.99, 50% off!", "café naïve résumé crème brûlée", "hello", "world", "speech synthesis", "unhappiness",
] * 40
PROBES = [
    "hello", "Hello world.", "        hello", "Test sentence.", "        Hello world.", "", " ", "  two  spaces  ", "tab\there", "line\nbreak",
    "The quick brown fox.", "unhappiness is widely regarded", "café", "café", "Ｈｅｌｌｏ", "ﬁne", "x² + y²",
    "non breaking", "zero​width", "emoji \U0001f600 here", "中文 text", "▁already", "42 apples, 3.14 pies", "MiXeD CaSe WoRdS",
    "trailing space ", "question? answer! ok...", "a", "zzzzqqqq", "한국어", "ệ",
]


def main():
    model = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(CORPUS), model_writer=model, vocab_size=260, model_type="unigram",
                                   character_coverage=1.0, normalization_rule_name="nfkc", add_dummy_prefix=True,
                                   remove_extra_whitespaces=False, split_by_whitespace=True, hard_vocab_limit=False,
                                   unk_id=0, bos_id=1, eos_id=2, pad_id=-1, num_threads=1, minloglevel=2)
    blob = model.getvalue()
    with open(os.path.join(HERE, "tiny_unigram.model"), "wb") as f:
        f.write(blob)
    sp = spm.SentencePieceProcessor(model_proto=blob)
    out = {"note": "ids produced by Python sentencepiece %s (the upstream library) on tests/golden/tiny_unigram.model" % spm.__version__,
           "vocab_size": sp.get_piece_size(),
           "cases": [{"text": t, "ids": sp.encode(t)} for t in PROBES]}
    with open(os.path.join(HERE, "tokenizer_ids.json"), "w") as f:
        json.dump(out, f, indent=0, ensure_ascii=True)
    print(f"vocab {sp.get_piece_size()}, model {len(blob)} bytes, {len(PROBES)} probes")


if __name__ == "__main__":
    main()
