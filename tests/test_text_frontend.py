"""the text front end in the product (csrc/text.cpp through the C ABI) against the reference's table tests
(transcribed as data in tests/golden/reference_kat.json) and against the oracle restatement (oracle/text_prepare.py).  CPU only."""
import json
import os
import random

import pytest

from oracle import text_prepare as TP

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_kat.json")) as f:
    KAT = {c["name"]: c for c in json.load(f)["cases"]}


def stub_encode(text):   # prepare_test.go:9-21: one token per whitespace-separated word
    return list(range(1, len(text.split()) + 1))


def impls(pkg):
    R = pkg.runtime
    return [("product", R.prepare_text, lambda t, mt: [(c.text, c.token_ids, c.num_words, c.max_frames, c.frames_after_eos) for c in R.prepare_chunks(t, stub_encode, mt)]),
            ("oracle", TP.prepare_text, lambda t, mt: [(c["text"], c["token_ids"], c["num_words"], TP.estimate_max_frames(c["num_tokens"]), TP.frames_after_eos(c["num_words"]))
                                                        for c in TP.prepare_chunks(t, stub_encode, mt)])]


def test_prepare_text_table(pkg):
    for who, prep, _ in impls(pkg):
        for c in KAT["text_prepare_table"]["cases"]:
            got = prep(c["in"])
            inner = got.lstrip(" ")
            assert ("starts" not in c or got.startswith(c["starts"])) and ("not_starts" not in c or not got.startswith(c["not_starts"])), (who, c, got)
            assert "ends" not in c or got.endswith(c["ends"]), (who, c, got)
            assert "lstrip_starts" not in c or inner.startswith(c["lstrip_starts"]), (who, c, got)
            assert "no" not in c or c["no"] not in got, (who, c, got)
            assert "lstrip_no" not in c or c["lstrip_no"] not in inner, (who, c, got)


def test_prepare_chunks_table(pkg):
    for who, _, chunks in impls(pkg):
        for c in KAT["text_chunks_first"]["cases"]:
            assert chunks(c["in"], 50)[0][0] == c["first"], (who, c)
        for c in KAT["text_chunks_props"]["cases"]:
            if "error" in c:
                with pytest.raises(Exception, match=c["error"]):
                    chunks(c["in"], c["max_tokens"])
                continue
            got = chunks(c["in"], c["max_tokens"])
            assert "n_chunks" not in c or len(got) == c["n_chunks"], (who, c, got)
            assert "min_chunks" not in c or len(got) >= c["min_chunks"], (who, c, got)
            assert "num_words" not in c or got[0][2] == c["num_words"], (who, c, got)
            assert "frames_after_eos" not in c or got[0][4] == c["frames_after_eos"], (who, c, got)
            for text, ids, _, mf, _ in got:
                assert ids == stub_encode(text) and mf == TP.estimate_max_frames(len(ids)) > 0   # NumTokens is of the final chunk text


def test_split_sentences_table():
    for c in KAT["text_split_sentences"]["cases"]:
        got = TP.split_sentences(c["in"])
        assert "want" not in c or got == c["want"]
        assert "min_parts" not in c or len(got) >= c["min_parts"]
        assert "first_contains" not in c or c["first_contains"] in got[0]
        assert all(s.strip() for s in got)


def test_estimates_match_reference_table(pkg):
    for n, want in KAT["estimate_max_frames"]["cases"] if "cases" in KAT.get("estimate_max_frames", {}) else [(3, 38), (4, 42), (9, 63), (10, 67), (14, 84), (50, 234)]:
        assert pkg.runtime.estimate_max_frames(n) == want == TP.estimate_max_frames(n)
    assert pkg.runtime.estimate_max_frames(-5) == pkg.runtime.estimate_max_frames(0) == 25
    assert pkg.runtime.estimate_max_frames(9, float("nan")) == 63 and pkg.runtime.estimate_max_frames(9, 0.0) == 63
    assert [pkg.runtime.frames_after_eos(w) for w in (0, 1, 4, 5, 50)] == [5, 5, 5, 3, 3]


def test_product_equals_oracle_on_a_corpus(pkg):
    """Same outputs from the two independent restatements on generated text: ASCII, Latin-1, Latin Extended-A, Greek and
    Cyrillic words, the white-space set of unicode.IsSpace, sentence punctuation, multi-sentence packing at several budgets."""
    rng = random.Random(5)
    alphabets = ["abcdefghijklmnopqrstuvwxyzABCXYZ0123456789", "àéîõüçñøåæþßÿÀÉÎ", "āăąćĉċčďđēěĝğġħĩīĭįıĵķĺļľłńňōőœŕřśşšţťŧũūůűųŵŷźżž", "αβγδεζηθικλμνξοπρστυφχψωςάέήίόύώ", "абвгдежзийклмнопрстуфхцчшщъыьэюяёђѓєѕіїјљњћќўџ"]
    spaces = [" ", "  ", "\n", "\r\n", "\t", " ", " ", "　", " \n "]
    texts = []
    for _ in range(300):
        words = []
        for _ in range(rng.randint(1, 14)):
            a = rng.choice(alphabets)
            w = "".join(rng.choice(a) for _ in range(rng.randint(1, 7)))
            words.append(w + rng.choice(["", "", "", ".", "!", "?", ",", "...", "?!"]))
        texts.append(rng.choice(["", " ", "\n"]) + "".join(w + rng.choice(spaces) for w in words))
    R = pkg.runtime
    for t in texts:
        assert R.prepare_text(t) == TP.prepare_text(t), repr(t)
        for mt in (3, 8, 50):
            a = [(c.text, c.token_ids, c.num_words, c.max_frames, c.frames_after_eos) for c in R.prepare_chunks(t, stub_encode, mt)]
            b = [(c["text"], c["token_ids"], c["num_words"], TP.estimate_max_frames(c["num_tokens"]), TP.frames_after_eos(c["num_words"])) for c in TP.prepare_chunks(t, stub_encode, mt)]
            assert a == b, (repr(t), mt)
