"""CPU-only checks of the drop-in boundary: the built library exports every symbol include/ptts.h declares, the
safetensors reader / arena planner behave like internal/safetensors/store.go on good and corrupt files, and compute
entry points fail loudly (no CPU fallback) when no HIP device is present."""
import ctypes
import json
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "ptts.h")).read()
    declared = set(re.findall(r"\b(ptts_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ptts_step_callback"}
    lib = ctypes.CDLL(pkg.runtime.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(pkg.runtime.ABI_SYMBOLS) <= declared | {"ptts_version"}
    assert b"gfx950" in ctypes.cast(lib.ptts_version, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_product_library_exports_no_test_hooks(pkg):
    """The library a host of the reference links (include/ptts.h) carries no fault injection, micro-benchmark, launch census or staged
    decoder hook: those are declared in include/ptts_debug.h and exported by libptts_hooks.so alone (the reference's seam has nothing of the kind:
    internal/tts/runtime.go:42-45)."""
    dbg = open(os.path.join(ROOT, "include", "ptts_debug.h")).read()
    hooks = set(re.findall(r"\b(ptts_[a-z0-9_]+)\s*\(", dbg))
    assert hooks == set(pkg.runtime.HOOK_SYMBOLS), hooks ^ set(pkg.runtime.HOOK_SYMBOLS)
    prod = set(re.findall(r"\b(ptts_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "ptts.h")).read()))
    assert not (hooks & prod) and not any(s.startswith("ptts_debug") for s in prod), hooks & prod
    import subprocess
    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("ptts_")}
    ex_prod, ex_hooks = exported(pkg.runtime.LIB_PATH), exported(pkg.runtime.HOOKS_PATH)
    assert not (ex_prod & hooks), ex_prod & hooks
    assert not any("debug" in s for s in ex_prod), [s for s in ex_prod if "debug" in s]
    assert ex_hooks == hooks, ex_hooks ^ hooks


def test_plan_matches_checkpoint_shapes(pkg, tmp_path):
    cfg = pkg.synth.SynthConfig.tiny()
    path = str(tmp_path / "tiny.safetensors")
    pkg.synth.write_safetensors(path, pkg.synth.make_checkpoint(cfg))
    for w in (pkg.WEIGHTS_F32, pkg.WEIGHTS_BF16):
        plan, nbytes = pkg.Model.plan(path, weights=w)
        img = pkg.Model.plan_fill_host(plan, nbytes)
        assert img.shape == (nbytes,) and img.any()
        pkg.Model.plan_free(plan)
    p32 = pkg.Model.plan(path, weights=pkg.WEIGHTS_F32)
    p16 = pkg.Model.plan(path, weights=pkg.WEIGHTS_BF16)
    assert p16[1] < p32[1]
    pkg.Model.plan_free(p32[0]); pkg.Model.plan_free(p16[0])


def _blob(entries, raw, header_extra=None):
    h = dict(entries)
    if header_extra:
        h.update(header_extra)
    hj = json.dumps(h).encode()
    return struct.pack("<Q", len(hj)) + hj + raw


@pytest.mark.parametrize("blob,msg", [
    (b"\x01\x02", "file too short"),                                              # store_test.go:166-191
    (struct.pack("<Q", 1000) + b"{}", "exceeds file size"),
    (struct.pack("<Q", 2) + b"{}" , "no tensors found"),
    (_blob({"a": {"dtype": "F64", "shape": [1], "data_offsets": [0, 8]}}, b"\0" * 8), "unsupported dtype"),
    (_blob({"a": {"dtype": "F32", "shape": [4], "data_offsets": [0, 8]}}, b"\0" * 8), "needs 16 bytes"),
    (_blob({"a": {"dtype": "F32", "shape": [2], "data_offsets": [0, 64]}}, b"\0" * 8), "exceeds file size"),
    (struct.pack("<Q", 5) + b"{nope", "parse header"),
])
def test_corrupt_files_are_rejected_like_the_reference(pkg, blob, msg):
    buf = (ctypes.c_char * len(blob)).from_buffer_copy(blob)
    p = ctypes.c_void_p()
    rc = pkg.runtime.lib().ptts_plan_create_bytes(buf, len(blob), None, ctypes.byref(p))
    assert rc != 0
    assert msg in pkg.runtime.lib().ptts_last_error().decode()


def test_missing_tensor_is_reported_by_name(pkg):
    raw = np.zeros(4, np.float32).tobytes()
    blob = _blob({"flow_lm.other": {"dtype": "F32", "shape": [4], "data_offsets": [0, 16]}}, raw, {"__metadata__": {"format": "pt"}})
    buf = (ctypes.c_char * len(blob)).from_buffer_copy(blob)
    p = ctypes.c_void_p()
    assert pkg.runtime.lib().ptts_plan_create_bytes(buf, len(blob), None, ctypes.byref(p)) == pkg.runtime.PTTS_EFORMAT
    assert 'tensor "flow_lm.conditioner.embed.weight" not found' in pkg.runtime.lib().ptts_last_error().decode()


def test_no_cpu_fallback(pkg, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the product path runs for real in the -m gpu tests")
    path = str(tmp_path / "tiny.safetensors")
    pkg.synth.write_safetensors(path, pkg.synth.make_checkpoint(pkg.synth.SynthConfig.tiny()))
    with pytest.raises(pkg.PttsError, match="no HIP device"):
        pkg.Model.open(path)
    with pytest.raises(pkg.PttsError, match="no HIP device"):
        pkg.runtime.op_linear(np.ones((1, 4), np.float32), np.ones((2, 4), np.float32))


def test_synthetic_checkpoint_roundtrips_through_both_readers(pkg, tmp_path):
    from oracle import oracle as O
    cfg = pkg.synth.SynthConfig.tiny()
    t = pkg.synth.make_checkpoint(cfg)
    for dt in ("F32", "BF16", "F16"):
        path = str(tmp_path / f"t_{dt}.safetensors")
        pkg.synth.write_safetensors(path, t, dtype=dt)
        got = O.Store.open(path).read_all()
        want = pkg.synth.quantize_like_file(t, dt)
        assert set(got) == set(want)
        for k in want:
            assert np.array_equal(got[k], want[k].astype(np.float32)), k


def test_header_is_plain_c99(tmp_path):
    """The drop-in boundary is a C ABI: include/ptts.h must compile as C (cgo compiles it as C), warnings as errors."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include "ptts.h"\nint main(void) { ptts_request r; ptts_result s; ptts_dispatch_opts d; ptts_chunk_info c; (void)r; (void)s; (void)d; (void)c;'
                   ' return sizeof(ptts_result) == 56 && sizeof(ptts_opts) == 64 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0
