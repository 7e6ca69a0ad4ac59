"""the SentencePiece unigram encoder (internal/tokenizer/sentencepiece.go:19-40; algorithm
internal/tokenizer/sentencepiece_bytes_wasm.go = go-sentencepiece-encoder v1.1.1).  No GPU needed.

Three legs:  product (C++ in libptts_hip.so)  ==  oracle (Python restatement of the reference file)  ==  the ids Python
sentencepiece -- the upstream library the reference's own pinned vectors come from -- produces on a small unigram model
trained in the build container (tests/golden/make_tokenizer_fixture.py).  The reference's four pinned id vectors
(tokenizer_test.go:82-160) need the real tokenizer.model and run when PTTS_TOKENIZER_MODEL points at it."""
import json
import os
import sys
import unicodedata

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.sentencepiece_unigram import Unigram  # noqa: E402
from oracle import text_prepare as TP  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def toks(pkg):
    blob = open(os.path.join(GOLD, "tiny_unigram.model"), "rb").read()
    return pkg.runtime.Tokenizer(blob), Unigram(blob), json.load(open(os.path.join(GOLD, "tokenizer_ids.json")))


# Where the reference's encoder deliberately differs from upstream SentencePiece's "nfkc" normaliser (the text the service hands it
# has been through PrepareText, which leaves none of these): it maps EVERY White_Space to a separator (upstream keeps tab / newline
# as unknown characters), drops the control set of sentencepiece_bytes_wasm.go:222-260 (upstream keeps U+200B as unknown), and does
# not prepend a separator to a text that already starts with one (spToRunes, :279-292).  For these the reference restatement rules.
DIVERGES_FROM_UPSTREAM = {"tab\there": "tab -> separator", "line\nbreak": "newline -> separator", "zero\u200bwidth": "U+200B dropped",
                          "\u2581already": "no second separator"}


def test_fixture_ids_from_upstream_sentencepiece(toks):
    prod, orc, gold = toks
    assert prod.vocab_size == len(orc.pieces) == gold["vocab_size"]
    agree = 0
    for case in gold["cases"]:
        t = case["text"]
        assert prod.encode(t) == orc.encode(t), ("product vs oracle", t)
        if t in DIVERGES_FROM_UPSTREAM:
            assert orc.encode(t) != case["ids"], ("documented divergence vanished", t)
        else:
            assert orc.encode(t) == case["ids"], ("oracle vs upstream sentencepiece", t)
            agree += 1
    assert agree == len(gold["cases"]) - len(DIVERGES_FROM_UPSTREAM) == 26


def test_product_equals_oracle_on_a_generated_corpus(toks):
    prod, orc, _ = toks
    rng = np.random.default_rng(0)
    alphabet = list("abcdefghijklmnopqrstuvwxyzABCDEFGHIJ .,!?'-0123456789\t\n") + ["é", "ß", "ﬁ", "²", "　", " ", "​", "中", "한", "▁", "́", "\U0001f600", "\x01", "ǆ", "Ǆ", "ﷺ"]
    for _ in range(400):
        n = int(rng.integers(0, 40))
        t = "".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), n))
        assert prod.encode(t) == orc.encode(t), repr(t)
    words = ["hello", "world", "the", "quick", "brown", "fox", "unhappiness", "woodchuck", "speech", "synthesis", "times,", "question."]
    for _ in range(200):
        t = " ".join(words[int(i)] for i in rng.integers(0, len(words), int(rng.integers(1, 30))))
        assert prod.encode(t) == orc.encode(t), t


def test_nfkc_tables_against_unicodedata(pkg):
    """The generated tables reproduce unicodedata's NFKC (same Unicode version by construction): every BMP code point alone,
    combining sequences that need reordering and composition, Hangul."""
    R = pkg.runtime
    for lo in range(0, 0x10000, 0x400):
        s = "".join(chr(c) for c in range(lo, lo + 0x400) if not 0xD800 <= c <= 0xDFFF and c != 0)
        # one code point at a time (a separator keeps neighbours from composing with each other)
        joined = "|".join(s)
        assert R.nfkc(joined) == "|".join(unicodedata.normalize("NFKC", c) for c in s), hex(lo)
    for t in ["é", "ẹ́", "ẹ́", "ǟ", "각", "각", "Ḍ̇", "ḍ̇",
              "Å", "ﬁ", "ẛ̣", "q̣̇", "\U0001d400\U0001d7ce", "㎒", "ｶﾞ", "क़", "ཱི", "̈́"]:
        assert R.nfkc(t) == unicodedata.normalize("NFKC", t), [hex(ord(c)) for c in t]


def test_whitespace_is_kept_piece_by_piece(toks):
    """tokenizer_test.go:122-141: eight leading spaces give eight U+2581 pieces in front of the word's own pieces."""
    prod, orc, _ = toks
    plain, padded = prod.encode("hello"), prod.encode("        hello")
    sep = [i for i, (p, _, _) in enumerate(orc.pieces) if p == "▁"]
    assert len(sep) == 1 and padded[:8] == sep * 8 and padded[8:] == plain
    assert prod.encode("") == [] and orc.encode("") == []


def test_prepare_chunks_with_the_builtin_encoder(pkg, toks):
    """PrepareChunks (prepare.go:105-184) driven by the library's own encoder (no callback) == the oracle's PrepareChunks over the
    oracle's encoder."""
    prod, orc, _ = toks
    text = ("The quick brown fox jumps over the lazy dog. She sells sea shells by the sea shore! How much wood would a woodchuck chuck? "
            "It was the best of times, it was the worst of times. To be, or not to be, that is the question.")
    got = pkg.runtime.prepare_chunks(text, prod, max_tokens=50)
    want = TP.prepare_chunks(text, orc.encode, 50)
    assert len(got) == len(want) > 1
    for g, w in zip(got, want):
        assert (g.text, g.token_ids, g.num_words) == (w["text"], w["token_ids"], w["num_words"])
        assert g.max_frames == TP.estimate_max_frames(len(w["token_ids"])) and g.frames_after_eos == TP.frames_after_eos(w["num_words"])


def test_errors_are_worded_like_the_reference(pkg, tmp_path):
    R = pkg.runtime
    with pytest.raises(R.PttsError) as e:
        R.Tokenizer("")
    assert "tokenizer model path must not be empty" in str(e.value)      # sentencepiece.go:20-22
    with pytest.raises(R.PttsError) as e:
        R.Tokenizer(str(tmp_path / "missing.model"))
    assert "load sentencepiece model" in str(e.value)                     # sentencepiece.go:27-29
    with pytest.raises(R.PttsError) as e:
        R.Tokenizer(b"\xff\xff\xff")
    assert "unmarshal sentencepiece model" in str(e.value)                # sentencepiece_bytes_wasm.go:35-37


REAL = os.environ.get("PTTS_TOKENIZER_MODEL", "")


@pytest.mark.skipif(not os.path.exists(REAL), reason="needs the real tokenizer.model (PTTS_TOKENIZER_MODEL): not available offline")
@pytest.mark.parametrize("text,ids", [("hello", [1876, 393]), ("Hello world.", [2994, 578, 263]),
                                      ("        hello", [260] * 8 + [1876, 393]), ("Test sentence.", [602, 552, 1472, 599, 263])])
def test_reference_pinned_id_vectors(pkg, text, ids):
    """internal/tokenizer/tokenizer_test.go:82-160 (ids recorded there from the upstream Python tokenizer)."""
    blob = open(REAL, "rb").read()
    assert pkg.runtime.Tokenizer(blob).encode(text) == ids
    assert Unigram(blob).encode(text) == ids
