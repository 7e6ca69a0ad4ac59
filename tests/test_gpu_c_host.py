"""The C ABI driven from a C host: tests/host_c/smoke.c is compiled with plain `gcc -std=c99` against include/ptts.h, linked to
libptts_hip.so and run as a child process with neither Python nor PyTorch in it -- the position of the reference's Go service behind
cgo (internal/tts/service.go:39-98, runtime_native_safetensors.go:36-38).  Its HIP runtime is therefore the
system's /opt/rocm libamdhip64 (the library's DT_NEEDED / RUNPATH), not the copy PyTorch bundles, which every ctypes test shares.
One request alone (n_reqs = 1: the reference's GenerateAudio), then a batch of 8 (two of them on a voice model state passed as arrays,
two on the same voice read from its FILE by the library: ptts_voice_file_open / ptts_voice_open); PCM, latents,
frame counts against the oracle at the smoke tolerances of tests/test_gpu_model.py."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from _parity import parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MULTI_LAT_TOL = (2.5e-4, 5e-2)
MULTI_PCM_TOL = (3e-4, 1e-1)


def _modules(tensors):
    mods = {}
    for name, t in tensors.items():
        mod, key = name.rsplit("/", 1)
        mods.setdefault(mod, {})[key] = np.asarray(t, np.float32 if key == "cache" else np.int64)
    return mods


def build_host(tmp_path) -> str:
    """gcc only: the header and the program are C99; the library is found through -L / -rpath like any C dependency."""
    exe = str(tmp_path / "smoke")
    libdir = os.path.join(ROOT, "go-pocket-tts_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host_c", "smoke.c"), "-o", exe, "-L", libdir, "-lptts_hip", "-Wl,-rpath," + libdir])
    return exe


@pytest.mark.skipif(not shutil.which("gcc"), reason="no gcc")
def test_c_host_compiles_and_links_against_the_library(pkg, tmp_path):
    """CPU side of the same check: the program builds against the header and resolves every symbol it uses from libptts_hip.so."""
    exe = build_host(tmp_path)
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libptts_hip.so" in out and "not found" not in out.split("libptts_hip.so")[1].splitlines()[0], out
    assert "libtorch" not in out and "python" not in out.lower(), out


@pytest.mark.gpu
@pytest.mark.skipif(not shutil.which("gcc"), reason="no gcc")
def test_c_host_generates_audio_without_python_or_torch(pkg, tmp_path):
    synth = pkg.synth
    cfg = synth.SynthConfig.tiny()
    path = str(tmp_path / "tiny.safetensors")
    synth.write_safetensors(path, synth.make_checkpoint(cfg, seed=1234))
    voice = synth.make_voice_state(cfg, offset=6, capacity=8)
    mods = _modules(voice)
    names = [f"transformer.layers.{i}.self_attn" for i in range(cfg.n_layers)]
    T, H, D = mods[names[0]]["cache"].shape[2:]
    rng = np.random.default_rng(5)
    reqs = [(np.array([10, 20, 30], np.int64), 3, 0)]
    for i in range(8):
        reqs.append((rng.integers(0, cfg.n_bins, size=3 + i, dtype=np.int64), 2 + i % 3, 1 if i in (2, 5) else 2 if i in (3, 7) else 0))
    vpath = str(tmp_path / "voice.safetensors")          # the same voice as a FILE: requests 3 and 7 let the library read and upload it
    synth.write_safetensors(vpath, voice)
    with open(tmp_path / "case.bin", "wb") as f:
        f.write(struct.pack("<4i", cfg.n_layers, T, H, D))
        f.write(np.array([int(mods[n]["offset"].reshape(-1)[0]) for n in names], np.int64).tobytes())
        for n in names:
            f.write(np.ascontiguousarray(mods[n]["cache"], np.float32).tobytes())
        f.write(struct.pack("<i", len(reqs)))
        for toks, steps, use_voice in reqs:
            f.write(struct.pack("<3i", len(toks), steps, use_voice))
            f.write(toks.tobytes())
    exe = build_host(tmp_path)
    # a clean child: no PYTHONPATH tricks, no preloaded HIP runtime; LD_LIBRARY_PATH is left as the box has it
    env = {k: v for k, v in os.environ.items() if not k.startswith("PTTS_")}
    r = subprocess.run([exe, path, str(tmp_path / "case.bin"), str(tmp_path / "out.bin"), vpath], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    maps = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "torch" not in maps, maps
    om = O.OracleModel.from_file(path)
    raw = open(tmp_path / "out.bin", "rb").read()
    pos = 0
    for i, (toks, steps, use_voice) in enumerate(reqs):
        status, n_frames, eos_step, ldim = struct.unpack_from("<4i", raw, pos); pos += 16
        (n_samples,) = struct.unpack_from("<q", raw, pos); pos += 8
        assert status == 0, (i, status)
        pcm = np.frombuffer(raw, np.float32, n_samples, pos); pos += 4 * n_samples
        lat = np.frombuffer(raw, np.float32, n_frames * ldim, pos).reshape(n_frames, ldim); pos += 4 * n_frames * ldim
        ref = om.generate(toks, max_steps=steps, eos_threshold=1e30, frames_after_eos=3, voice_state=mods if use_voice else None)
        assert n_frames == ref["n_frames"] == steps and eos_step == -1 and n_samples == steps * 1920, (i, n_frames, eos_step, n_samples)
        parity(f"C host request {i} latents", lat, ref["latents"], MULTI_LAT_TOL)
        parity(f"C host request {i} pcm", pcm, ref["pcm"], MULTI_PCM_TOL)
    assert pos == len(raw)
    om.close()
