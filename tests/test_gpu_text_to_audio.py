"""Text in, audio out, with nothing of the host's in the loop: tts.Service.Synthesize (service.go:107-153) over the library's own
PrepareChunks + SentencePiece encoder + GenerateAudio, against the same pipeline on the oracle side (oracle PrepareChunks over the
oracle's unigram encoder, one oracle GenerateAudio per chunk, concatenated)."""
import dataclasses
import os

import numpy as np
import pytest

from oracle import oracle as O
from oracle import text_prepare as TP
from oracle.sentencepiece_unigram import Unigram
from _parity import parity

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_text_to_audio_with_the_builtin_tokenizer(pkg, tmp_path):
    blob = open(os.path.join(ROOT, "tests", "golden", "tiny_unigram.model"), "rb").read()
    tok, otok = pkg.runtime.Tokenizer(blob), Unigram(blob)
    synth = pkg.synth
    cfg = dataclasses.replace(synth.SynthConfig.tiny(), n_bins=tok.vocab_size + 5)   # every piece id has an embedding row
    path = str(tmp_path / "tiny_vocab.safetensors")
    synth.write_safetensors(path, synth.make_checkpoint(cfg, seed=99))
    om = O.OracleModel.from_file(path)
    gm = pkg.Model.open(path, device=0, max_batch=16)
    text = ("The quick brown fox jumps over the lazy dog. She sells sea shells by the sea shore! How much wood would a woodchuck chuck? "
            "It was the best of times, it was the worst of times. To be, or not to be, that is the question.")
    svc = pkg.Service(gm, tok, pkg.TTSConfig(eos_threshold=float("inf"), max_steps=3))   # 3 steps per chunk keep the oracle run short
    pairs = svc.synthesize_chunks(text)
    want_chunks = TP.prepare_chunks(text, otok.encode, 50)
    assert len(pairs) == len(want_chunks) >= 2
    want = []
    for (c, r), w in zip(pairs, want_chunks):
        assert c.text == w["text"] and c.token_ids == w["token_ids"] and c.num_words == w["num_words"]
        assert c.frames_after_eos == TP.frames_after_eos(w["num_words"]) and c.max_frames == TP.estimate_max_frames(len(w["token_ids"]))
        assert r.n_frames == 3
        want.append(om.generate(w["token_ids"], max_steps=3, eos_threshold=1e30, frames_after_eos=c.frames_after_eos)["pcm"])
    got = svc.synthesize(text)
    parity("text -> audio (built-in tokenizer)", got, np.concatenate(want), (3e-4, 1e-1))
    # the estimate-driven step budget of the default configuration (service.go:271-278)
    assert pkg.Service(gm, tok).generate_config(pairs[0][0]).max_steps == pairs[0][0].max_frames
    # the optional post-processing of the CLI on the result (synth.go:361-390)
    post = pkg.runtime.dsp_apply(got, normalize=True, fade_in_ms=5, fade_out_ms=5)
    assert abs(float(np.abs(post).max()) - 1.0) < 1e-6 and post[0] == 0.0 and post[-1] == 0.0
    gm.close()
    om.close()
