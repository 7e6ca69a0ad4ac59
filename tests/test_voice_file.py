"""Voice FILES through the C ABI (ptts_voice_file_*: internal/safetensors/reader.go:69-155,219-308 and the consumer-side checks of
internal/native/flow_transformer.go:451-590), against the reference's own cases held as data in tests/golden/voice_file_cases.json
-- and the oracle's restatement against the same cases, so the two are pinned by the same vectors.  CPU only: nothing here needs
a GPU (the upload, ptts_voice_open, is covered by tests/test_gpu_model.py and the C host)."""
import json
import os
import struct

import numpy as np
import pytest

import ptts_amd
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "voice_file_cases.json")) as f:
    CASES = {c["name"]: c for c in json.load(f)["cases"]}


@pytest.fixture(scope="module")
def pkg():
    p = ptts_amd.load()
    p.runtime.build()
    return p


def tensor_values(t) -> np.ndarray:
    if "gen" in t:
        n, scale = t["gen"]
        return (np.arange(n).astype(np.float32) * np.float32(scale)).astype(np.float32)   # float32(i) * scale, as the Go tests build them
    return np.array(t.get("values", []), np.float64)


def tensor_bytes(t) -> bytes:
    if "raw_bytes" in t:
        return bytes(t["raw_bytes"])
    v = tensor_values(t)
    dt = t["dtype"]
    if dt == "F32":
        return v.astype("<f4").tobytes()
    if dt == "I64":
        return v.astype("<i8").tobytes()
    if dt == "BF16":
        return (v.astype("<f4").view("<u4") >> 16).astype("<u2").tobytes()   # the case's values are exact in bf16
    raise AssertionError(dt)


def decoded(t) -> np.ndarray:
    """what store.go:339-395 decodes the tensor to"""
    return tensor_values(t).astype(np.float32).reshape(t["shape"])


def build_file(case) -> bytes:
    """buildSafetensors of reader_test.go:23-77: sorted names, contiguous data"""
    if "raw_hex" in case:
        return bytes.fromhex(case["raw_hex"])
    if "raw_header" in case:
        h = case["raw_header"].encode()
        return struct.pack("<Q", len(h)) + h
    header, blobs, off = {}, [], 0
    if "metadata" in case:
        header["__metadata__"] = case["metadata"]
    for name in sorted(case["tensors"]):
        t = case["tensors"][name]
        raw = tensor_bytes(t)
        n = len(raw)
        if t.get("lie_about_length"):
            n = int(np.prod(t["shape"])) * 4
        header[name] = {"dtype": t["dtype"], "shape": t["shape"], "data_offsets": [off, off + n]}
        blobs.append(raw)
        off += len(raw)
    h = json.dumps(header).encode()
    return struct.pack("<Q", len(h)) + h + b"".join(blobs)


class Product:
    """the library's answers"""
    err = None

    def __init__(self, pkg, case, tmp_path, by_path):
        self.pkg = pkg
        if "path" in case:
            self.vf = pkg.VoiceFile(case["path"])
            return
        blob = build_file(case)
        if by_path:
            p = tmp_path / (case["name"] + ".safetensors")
            p.write_bytes(blob)
            self.vf = pkg.VoiceFile(str(p))
        else:
            self.vf = pkg.VoiceFile(blob)

    kind = property(lambda self: self.vf.kind)

    def embedding(self):
        e = self.vf.embedding()
        return e.data, list(e.shape)

    def modules(self):
        return self.vf.model_state().modules

    def state(self, n, heads, hd):
        ptrs, steps, offs = self.vf.state_arrays(n, heads, hd)
        caches = []
        for i in range(n):
            sz = 2 * int(steps[i]) * heads * hd
            caches.append(np.ctypeslib.as_array(ptrs[i], shape=(sz,)).copy().reshape(2, 1, int(steps[i]), heads, hd))
        return caches, [int(x) for x in steps], [int(x) for x in offs]


class Oracle:
    """the restatement's answers (oracle/oracle.py)"""

    def __init__(self, case):
        if "path" in case:
            self.store = O.Store.open(case["path"])
        else:
            self.store = O.Store(build_file(case))

    kind = property(lambda self: O.classify_voice(self.store.names))

    def embedding(self):
        e = O.load_voice_embedding(self.store)
        return e, list(e.shape)

    def modules(self):
        return O.load_voice_model_state(self.store)

    def state(self, n, heads, hd):
        caches, steps, offs = O.voice_state_layers(self.modules(), n, heads, hd)
        return caches, list(steps), list(offs)


def make(side, pkg, case, tmp_path):
    if side == "oracle":
        return Oracle(case)
    return Product(pkg, case, tmp_path, by_path=(side == "product-path"))


ERRORS = (Exception,)


@pytest.mark.parametrize("side", ["product-bytes", "product-path", "oracle"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_voice_file_case(pkg, tmp_path, name, side):
    case = CASES[name]
    want = case["want"]
    if "path" in case and side == "product-bytes":
        pytest.skip("a missing file has no bytes")
    if want.get("open_error"):
        with pytest.raises(ERRORS):
            make(side, pkg, case, tmp_path)
        return
    v = make(side, pkg, case, tmp_path)
    assert v.kind == want["kind"]

    if "embedding_shape" in want:
        data, shape = v.embedding()
        assert shape == want["embedding_shape"]
        ref = decoded(case["tensors"][want["embedding_equals"]])
        assert np.array_equal(np.asarray(data, np.float32).ravel().view(np.uint32), ref.ravel().view(np.uint32))   # bit for bit (reader_test.go:503-539)
    if "embedding_error" in want:
        with pytest.raises(ERRORS, match=want["embedding_error"]):
            v.embedding()
    if "modules" in want:
        mods = v.modules()
        assert sorted(mods) == sorted(want["modules"])
        for mname, tensors in want["modules"].items():
            for key, t in tensors.items():
                got = np.asarray(mods[mname][key], np.float32)
                assert list(got.shape) == t["shape"], (mname, key, got.shape)
                assert got.ravel().tolist() == [float(x) for x in t["data"]]
    if "modules_error" in want:
        with pytest.raises(ERRORS, match=want["modules_error"]):
            v.modules()
    if "state_open_error" in want:
        with pytest.raises(ERRORS, match=want["state_open_error"]):
            v.modules()
    lay = case.get("layers")
    if "state" in want:
        caches, steps, offs = v.state(lay["n"], lay["heads"], lay["head_dim"])
        assert steps == want["state"]["steps"] and offs == want["state"]["offsets"]
        # the re-layout the device kernel applies to these arrays, on the host: [2,B,T,H,D] -> [B,H,T,D] (flow_transformer.go:592-627)
        k0 = np.transpose(caches[0][0], (0, 2, 1, 3)).ravel().tolist()
        v0 = np.transpose(caches[0][1], (0, 2, 1, 3)).ravel().tolist()
        assert k0 == want["state"]["k0"] and v0 == want["state"]["v0"]
    if "state_error" in want:
        with pytest.raises(ERRORS, match=want["state_error"].replace("[", r"\[").replace("]", r"\]")):
            v.state(lay["n"], lay["heads"], lay["head_dim"])


def test_load_voice_conditioning_routes_by_kind(pkg, tmp_path):
    """tts.loadVoiceConditioning (service.go:216-246): blank path -> nothing; model state -> VoiceModelState; else an embedding"""
    assert pkg.load_voice_conditioning("   ") == {}
    p1 = tmp_path / "state.safetensors"
    p1.write_bytes(build_file(CASES["inspect_model_state"]))
    got = pkg.load_voice_conditioning(str(p1))
    assert list(got) == ["voice_model_state"] and "transformer.layers.0.self_attn" in got["voice_model_state"].modules
    p2 = tmp_path / "emb.safetensors"
    p2.write_bytes(build_file(CASES["embedding_values_preserved"]))
    got = pkg.load_voice_conditioning(str(p2))
    assert list(got) == ["voice_embedding"] and list(got["voice_embedding"].shape) == [1, 2, 4]
    with pytest.raises(pkg.PttsError, match="inspect voice safetensors"):
        pkg.load_voice_conditioning(str(tmp_path / "missing.safetensors"))
    p3 = tmp_path / "bad.safetensors"
    p3.write_bytes(build_file(CASES["embedding_1d_returns_error"]))
    with pytest.raises(pkg.PttsError, match="load voice embedding: safetensors: voice embedding has 1D shape"):
        pkg.load_voice_conditioning(str(p3))


def test_synthetic_voice_files_round_trip(pkg, tmp_path):
    """the files the GPU tests use as voices (synth.make_voice_state / make_voice_embedding, stock-voice layout incl. the legacy
    current_end form and NaN padding past the offset) read back through the library as the arrays they were written from"""
    cfg = pkg.synth.SynthConfig.tiny()
    for legacy in (False, True):
        tens = pkg.synth.make_voice_state(cfg, offset=9, capacity=(9 if legacy else 16), legacy_current_end=legacy)
        p = tmp_path / f"voice_{int(legacy)}.safetensors"
        pkg.synth.write_safetensors(str(p), tens)
        vf = pkg.VoiceFile(str(p))
        assert vf.kind == "model_state"
        n_layers = len([k for k in tens if k.endswith("/cache")])
        heads, hd = tens["transformer.layers.0.self_attn/cache"].shape[3:]
        ptrs, steps, offs = vf.state_arrays(n_layers, int(heads), int(hd))
        assert [int(x) for x in offs] == [9] * n_layers
        for i in range(n_layers):
            want = tens[f"transformer.layers.{i}.self_attn/cache"].astype(np.float32)
            got = np.ctypeslib.as_array(ptrs[i], shape=(want.size,)).reshape(want.shape)
            assert int(steps[i]) == want.shape[2]
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))   # NaN padding included, bit for bit
    emb = pkg.synth.make_voice_embedding(cfg, frames=7)
    p = tmp_path / "emb.safetensors"
    pkg.synth.write_safetensors(str(p), emb)
    vf = pkg.VoiceFile(str(p))
    assert vf.kind == "embedding"
    e = vf.embedding()
    (name, arr), = emb.items()
    assert list(e.shape) == [1] + list(arr.shape[-2:]) and np.array_equal(np.asarray(e.data).ravel(), arr.astype(np.float32).ravel())
