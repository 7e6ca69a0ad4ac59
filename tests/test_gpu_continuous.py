"""Continuous batching in the dispatcher (go-pocket-tts_amd/csrc/continuous.cpp): one long-lived batch per model, slots
refilled between groups of AR steps.  The reference's counterpart is the worker semaphore (internal/server/server.go:398-421) in front
of one-chunk-at-a-time GenerateAudio calls whose lengths differ (EOS, runtime_native_safetensors.go:178-190; per-chunk budgets,
internal/tts/service.go:138-153).  What must hold: every caller's audio equals what the same request returns on its own."""
import threading

import numpy as np
import pytest

from oracle import oracle as O
from _parity import parity

pytestmark = pytest.mark.gpu


def _modules(tensors):
    mods = {}
    for name, t in tensors.items():
        mod, key = name.rsplit("/", 1)
        mods.setdefault(mod, {})[key] = np.asarray(t, np.float32 if key == "cache" else np.int64)
    return mods


@pytest.fixture(scope="module")
def tiny(pkg, tmp_path_factory):
    synth = pkg.synth
    cfg = synth.SynthConfig.tiny()
    path = str(tmp_path_factory.mktemp("ckpt") / "tiny.safetensors")
    synth.write_safetensors(path, synth.make_checkpoint(cfg, seed=1234))
    om = O.OracleModel.from_file(path)
    gm = pkg.Model.open(path, device=0, max_batch=8)
    yield cfg, path, om, gm
    gm.close()
    om.close()


def run_clients(d, prompts, cfgs, stagger_s=0.0):
    import time
    got, errs = [None] * len(prompts), [None] * len(prompts)

    def client(i):
        try:
            if stagger_s:
                time.sleep(stagger_s * i)
            got[i] = d.generate(prompts[i], cfgs[i])
        except Exception as e:  # noqa: BLE001
            errs[i] = e

    ts = [threading.Thread(target=client, args=(i,)) for i in range(len(prompts))]
    [t.start() for t in ts]
    [t.join(180) for t in ts]
    return got, errs


@pytest.mark.parametrize("graph", [False, True])
def test_every_caller_gets_its_stand_alone_audio_with_slots_refilled_on_the_way(pkg, tiny, graph):
    """24 requests of mixed lengths (2..17 frames), 4 slots, groups of 3 steps: slots are refilled many times while others are still
    generating.  Mixed in: a device voice, a host voice state, a voice embedding, temperature 0.7 with a named seed, PCM16 egress.
    Each result against the same request run alone through ptts_generate (kernel-selection rounding only), a few against the oracle."""
    cfg, _, om, gm = tiny
    gm.set_use_graph(graph)
    rng = np.random.default_rng(31)
    n = 24
    prompts = [rng.integers(1, cfg.n_bins, size=int(rng.integers(3, 9))).astype(np.int64) for _ in range(n)]
    steps = [int(rng.integers(2, 18)) for _ in range(n)]
    voice_mods = _modules(pkg.synth.make_voice_state(cfg, offset=7, seed=5))
    dv = gm.upload_voice(pkg.VoiceModelState(voice_mods))
    ve = pkg.synth.make_voice_embedding(cfg, frames=5)["audio_prompt"]
    cfgs = []
    for i in range(n):
        kw = dict(max_steps=steps[i], eos_threshold=1e30, want_latents=True)
        if i % 6 == 1:
            kw["device_voice"] = dv
        elif i % 6 == 2:
            kw["voice_model_state"] = pkg.VoiceModelState(voice_mods)
        elif i % 6 == 3:
            kw["voice_embedding"] = pkg.VoiceEmbedding(ve, list(ve.shape))
        elif i % 6 == 4:
            kw.update(temperature=0.7, noise_seed=1000 + i)
        elif i % 6 == 5:
            kw["pcm16"] = True
        cfgs.append(pkg.RuntimeGenerateConfig(**kw))
    want = [gm.generate_batch([prompts[i]], [cfgs[i]])[0] for i in range(n)]
    d = pkg.Dispatcher([gm], max_batch=4, window_us=2000, continuous=True, cont_kv_capacity=64, cont_max_steps=32, cont_steps_per_group=3)
    try:
        got, errs = run_clients(d, prompts, cfgs)
        assert not any(errs), errs
        st = d.stats()
        assert st["requests"] == n and st["batches"] >= n // 4, st      # "batches" counts admissions here: 24 requests cannot fit 4 slots in fewer than 6
        for i in range(n):
            assert got[i].n_frames == want[i].n_frames == steps[i] and got[i].eos_step == -1
            parity(f"continuous latents[{i}]", got[i].latents, want[i].latents, (1e-4, 5e-3))
            if cfgs[i].pcm16:
                assert got[i].pcm.dtype == np.int16 and np.abs(got[i].pcm.astype(np.int32) - want[i].pcm.astype(np.int32)).max() <= 2
            else:
                parity(f"continuous pcm[{i}]", got[i].pcm, want[i].pcm, (1e-4, 5e-2))
        for i in (0, 1, 2, 3):   # and against the reference's arithmetic: no voice, device voice, host voice state, voice embedding
            kw = {}
            if i % 6 in (1, 2):
                kw["voice_state"] = voice_mods
            if i % 6 == 3:
                kw["voice_emb"] = ve[0]
            ref = om.generate(prompts[i], max_steps=steps[i], eos_threshold=1e30, frames_after_eos=3, **kw)
            parity(f"continuous latents[{i}] vs oracle", got[i].latents, ref["latents"], (2.5e-4, 5e-2))
            parity(f"continuous pcm[{i}] vs oracle", got[i].pcm, ref["pcm"], (3e-4, 1e-1))
    finally:
        d.close()
    dv.close()
    gm.set_use_graph(False)


def test_eos_frees_a_slot_and_the_tail_frames_are_kept(pkg, tiny):
    """Utterances that end by EOS (threshold picked from the oracle's own logits, then the frames_after_eos tail) leave their slot
    early; frame counts and EOS steps are exact, the audio is the stand-alone audio."""
    cfg, _, om, gm = tiny
    rng = np.random.default_rng(7)
    n = 10
    prompts = [rng.integers(1, cfg.n_bins, size=int(rng.integers(3, 8))).astype(np.int64) for _ in range(n)]
    cfgs, want = [], []
    for i in range(n):
        ref = om.generate(prompts[i], max_steps=14, eos_threshold=1e30, frames_after_eos=2)
        # EOS fires at the first step whose logit exceeds the threshold: a step whose logit is a new running maximum can be made that
        # step by a threshold halfway between it and the previous maximum (margin = half the difference, far above the GPU's logit error)
        cands, run = [], -np.inf
        for t, l in enumerate(ref["eos_logits"]):
            if l > run:
                if t > 0 and l - run > 2e-3:
                    cands.append((abs(t - (2 + i % 7)), t, float((l + run) / 2)))
                run = l
        thr = min(cands)[2] if cands else 1e30
        c = pkg.RuntimeGenerateConfig(max_steps=14, eos_threshold=thr, frames_after_eos=2, want_latents=True)
        w = gm.generate_batch([prompts[i]], [c])[0]
        r2 = om.generate(prompts[i], max_steps=14, eos_threshold=thr, frames_after_eos=2)
        assert (w.n_frames, w.eos_step) == (r2["n_frames"], r2["eos_step"]), i
        cfgs.append(c)
        want.append(w)
    d = pkg.Dispatcher([gm], max_batch=3, window_us=1000, continuous=True, cont_kv_capacity=64, cont_max_steps=16, cont_steps_per_group=2)
    try:
        got, errs = run_clients(d, prompts, cfgs, stagger_s=0.002)
        assert not any(errs), errs
        assert len({w.n_frames for w in want}) > 1                # the lengths do differ
        for i in range(n):
            assert (got[i].n_frames, got[i].eos_step) == (want[i].n_frames, want[i].eos_step), i
            parity(f"continuous (EOS) latents[{i}]", got[i].latents, want[i].latents, (1e-4, 5e-3))
            parity(f"continuous (EOS) pcm[{i}]", got[i].pcm, want[i].pcm, (1e-4, 5e-2))
    finally:
        d.close()


def test_cancellation_callbacks_and_oversized_requests(pkg, tiny):
    """A request cancelled while it generates is answered ctx.Err()-style and its slot is reused; a request with a step callback and
    one with two Euler steps per frame run batch-at-a-time and still get their stand-alone audio."""
    cfg, _, om, gm = tiny
    rng = np.random.default_rng(9)
    prompts = [rng.integers(1, cfg.n_bins, size=5).astype(np.int64) for _ in range(6)]
    flag = np.zeros(1, np.int32)
    seen = []

    def on_step(step, mx):
        seen.append((step, mx))

    cfgs = [pkg.RuntimeGenerateConfig(max_steps=500, eos_threshold=1e30, cancel=flag),                    # cancelled on the way (500 steps take > 100 ms)
            pkg.RuntimeGenerateConfig(max_steps=6, eos_threshold=1e30, want_latents=True),
            pkg.RuntimeGenerateConfig(max_steps=4, eos_threshold=1e30, want_latents=True, step_callback=on_step),   # batch-at-a-time: callback
            pkg.RuntimeGenerateConfig(max_steps=40, eos_threshold=1e30, want_latents=True),                # batch-at-a-time: budget > cont_max_steps
            pkg.RuntimeGenerateConfig(max_steps=5, eos_threshold=1e30, want_latents=True),
            pkg.RuntimeGenerateConfig(max_steps=7, eos_threshold=1e30, want_latents=True)]
    want = [None] + [gm.generate_batch([prompts[i]], [pkg.RuntimeGenerateConfig(max_steps=cfgs[i].max_steps, eos_threshold=1e30, want_latents=True,
                                                                                lsd_decode_steps=cfgs[i].lsd_decode_steps)])[0] for i in range(1, 6)]
    d = pkg.Dispatcher([gm], max_batch=2, window_us=500, continuous=True, cont_kv_capacity=576, cont_max_steps=512, cont_steps_per_group=2)
    try:
        canceller = threading.Timer(0.03, lambda: flag.__setitem__(0, 1))
        canceller.start()
        got, errs = run_clients(d, prompts, cfgs)
        canceller.join()
        assert isinstance(errs[0], pkg.Cancelled), errs[0]
        assert not any(errs[1:]), errs
        assert seen == [(s, 4) for s in range(1, 5)]
        for i in range(1, 6):
            assert got[i].n_frames == want[i].n_frames
            parity(f"continuous mixed paths latents[{i}]", got[i].latents, want[i].latents, (1e-4, 5e-3))
            parity(f"continuous mixed paths pcm[{i}]", got[i].pcm, want[i].pcm, (1e-4, 5e-2))
    finally:
        d.close()
