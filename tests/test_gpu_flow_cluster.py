"""k_flow_cluster (csrc/flow_cluster.hip): the flow net's residual blocks (flow_net.go:116-172) as ONE launch whose eight workgroups per row
tile hand the rows round through tagged granules.  Held here to
  * the launches it replaces (PTTS_FLOW_CLUSTER=0: 2 x depth k_skinny launches, themselves held to the oracle by test_gpu_step_staged.py /
    test_gpu_fullsize.py): same arithmetic in the same order, so the frames must agree to rounding noise at every row count (1 row, a ragged
    tile, two tiles and a ragged third, the full 64), teacher-forced over several steps;
  * itself: the hand-off has no fence and no barrier, so what would break it is a stale or torn granule -- every frame of a long free-running
    batch must come out bit-identical on a second run, with plain launches and under graph replay, while the decoder runs beside it on
    the second stream (uneven load on the CUs that sweep);
  * the fault path's bookkeeping word stays clear.
Oracle parity of the clustered step itself: the full-size tests (test_gpu_fullsize.py, test_gpu_continuous.py) run it by default."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full(pkg):
    import bench
    cfg = pkg.synth.SynthConfig.full()
    return cfg, bench.checkpoint_path(pkg, "BF16", 0, lambda: None), bench.voice_modules(pkg, cfg)


def _staged_frames(pkg, path, cfg, rows, steps, cluster):
    """`steps` teacher-forced staged steps of a `rows`-slot batch (frames fed: a fixed pseudo-random sequence), with or without the cluster kernel."""
    os.environ["PTTS_FLOW_CLUSTER"] = "1" if cluster else "0"
    try:
        gm = pkg.Model.open(path, device=0, weights=pkg.WEIGHTS_BF16, kv=pkg.KV_BF16, max_batch=64)
        rng = np.random.default_rng(7)
        toks = [rng.integers(1, cfg.n_bins, size=int(rng.integers(3, 12))).astype(np.int64) for _ in range(rows)]
        b = gm.new_batch(rows, 128)
        b.prompt([gm.text_embeddings(t) for t in toks])
        pkg.runtime.launch_counts(True)
        frames = np.full((rows, cfg.ldim), np.nan, np.float32)
        outs = []
        for s in range(steps):
            out, logit, last = b.step(frames)
            outs.append((out.copy(), np.asarray(logit).copy()))
            frames = (0.5 * np.random.default_rng(100 + s).standard_normal((rows, cfg.ldim))).astype(np.float32)
        counts = pkg.runtime.launch_counts(False)
        b.close()
        gm.close()
        return outs, counts
    finally:
        os.environ.pop("PTTS_FLOW_CLUSTER", None)


@pytest.mark.parametrize("rows", [1, 5, 16, 37, 64])
def test_cluster_equals_the_launches_it_replaces(pkg, full, rows):
    cfg, path, _ = full
    steps = 4
    got, c1 = _staged_frames(pkg, path, cfg, rows, steps, True)
    want, c0 = _staged_frames(pkg, path, cfg, rows, steps, False)
    assert c1.get("k_flow_cluster", 0) == steps, c1          # one launch per step ...
    assert c0.get("k_flow_cluster", 0) == 0, c0
    assert c0["k_skinny"] - c1["k_skinny"] == 2 * cfg.flow_depth * steps, (c0, c1)   # ... in place of 2 x depth
    worst = 0.0
    for s, ((f1, l1), (f0, l0)) in enumerate(zip(got, want)):
        assert np.isfinite(f1).all()
        scale = float(np.abs(f0).max())
        worst = max(worst, float(np.abs(f1 - f0).max()) / scale)
        np.testing.assert_array_equal(l1, l0)                 # the EOS logit is computed in front of the flow net: untouched
    # same products, same summation order (only the two MFMA operands trade places): observed 0.0 -- any difference at all would be worth a look,
    # the bound is rounding noise of one frame
    assert worst <= 2e-6, worst
    print(f"rows {rows}: max |cluster - launches| / scale = {worst:.3g}")


def test_cluster_hand_offs_are_reproducible_beside_a_running_decoder(pkg, full):
    """64 x 40 frames free-running, three times (plain, plain, graph replay): bit-identical latents.  A free-running trajectory amplifies one wrong
    bit 1.15x per step, so a single stale granule anywhere in 40 steps x 11 exchanges x 4 tiles x 8 workgroups shows at the end; the decoder of
    the previous call's frames runs on the second stream meanwhile (streamed ranges: PTTS_MIMI_CHUNK)."""
    cfg, path, voice = full
    os.environ["PTTS_MIMI_CHUNK"] = "10"   # decode ranges of 10 frames under the loop: millisecond-long decoder blocks beside the sweeping workgroups
    try:
        gm = pkg.Model.open(path, device=0, weights=pkg.WEIGHTS_BF16, kv=pkg.KV_BF16, max_batch=64)
        v = gm.upload_voice(pkg.VoiceModelState(voice))
        prompts = [p.tolist() for p in pkg.synth.make_prompts(64, 25, 4000, seed=3)]
        c = pkg.RuntimeGenerateConfig(max_steps=40, eos_threshold=float("inf"), frames_after_eos=3, device_voice=v, want_latents=True)
        pkg.runtime.launch_counts(True)
        runs = [gm.generate_batch(prompts, [c] * 64)]
        counts = pkg.runtime.launch_counts(False)
        assert counts.get("k_flow_cluster", 0) == 40, counts
        runs.append(gm.generate_batch(prompts, [c] * 64))
        gm.set_use_graph(True)
        runs.append(gm.generate_batch(prompts, [c] * 64))
        for k in (1, 2):
            for i in range(64):
                np.testing.assert_array_equal(runs[k][i].latents, runs[0][i].latents, err_msg=f"run {k} slot {i}")
                np.testing.assert_array_equal(runs[k][i].pcm, runs[0][i].pcm, err_msg=f"run {k} slot {i} pcm")
        v.close()
        gm.close()
    finally:
        os.environ.pop("PTTS_MIMI_CHUNK", None)


def test_a_withheld_publish_times_out_and_the_call_still_returns_the_stand_alone_audio(pkg, full):
    """Every sweep of the kernel is bounded: with one workgroup withholding what it should publish (test hook, libptts_hooks.so), its peers give up after the
    bound instead of spinning and the launch runs to its end with the fault word raised.  Nobody is failed for it: the chunk is run again on the 2 x depth
    launches -- the same arithmetic in the same order (test above) -- so the caller gets the BITS of an undisturbed call; the event is counted
    (ptts_dispatch_stats.flow_cluster_fallbacks) and the engine keeps the launches from then on.  Only a call that has already spoken to its caller (a step
    callback) cannot be re-run: it fails with the hand-off error, as the reference's GenerateAudio fails a call whose step failed
    (internal/tts/runtime_native_safetensors.go:161-173)."""
    cfg, path, voice = full
    gm = pkg.Model.open(path, device=0, weights=pkg.WEIGHTS_BF16, kv=pkg.KV_BF16, max_batch=16)
    v = gm.upload_voice(pkg.VoiceModelState(voice))
    prompts = [p.tolist() for p in pkg.synth.make_prompts(13, 25, 4000, seed=5)]   # 13 rows: two tiles, the second ragged
    c = pkg.RuntimeGenerateConfig(max_steps=6, eos_threshold=float("inf"), frames_after_eos=3, device_voice=v, want_latents=True)
    good = gm.generate_batch(prompts, [c] * 13)
    pkg.runtime.launch_counts(True)
    gm.generate_batch(prompts, [c] * 13)
    assert pkg.runtime.launch_counts(False).get("k_flow_cluster", 0) == 6
    gm.debug_flow_cluster_inject(3)
    pkg.runtime.launch_counts(True)
    hit = gm.generate_batch(prompts, [c] * 13)          # no exception: the fault is absorbed
    counts = pkg.runtime.launch_counts(False)
    assert counts.get("k_flow_cluster", 0) == 6, counts  # the first pass ran the cluster (one launch of it faulted) ...
    for a, b in zip(hit, good):
        np.testing.assert_array_equal(a.latents, b.latents)
        np.testing.assert_array_equal(a.pcm, b.pcm)
    d = pkg.Dispatcher([gm], max_batch=16)
    assert d.stats()["flow_cluster_fallbacks"] == 1
    d.close()
    pkg.runtime.launch_counts(True)
    again = gm.generate_batch(prompts, [c] * 13)
    assert "k_flow_cluster" not in pkg.runtime.launch_counts(False)   # ... and this engine stays on the launches
    for a, b in zip(again, good):
        np.testing.assert_array_equal(a.latents, b.latents)
        np.testing.assert_array_equal(a.pcm, b.pcm)
    v.close()
    gm.close()
    # a call that reports its steps to the caller cannot be replayed behind the caller's back: it fails, loudly
    gm = pkg.Model.open(path, device=0, weights=pkg.WEIGHTS_BF16, kv=pkg.KV_BF16, max_batch=16)
    v = gm.upload_voice(pkg.VoiceModelState(voice))
    seen = []
    cb = pkg.RuntimeGenerateConfig(max_steps=6, eos_threshold=float("inf"), frames_after_eos=3, device_voice=v, step_callback=lambda s, m: seen.append(s))
    gm.debug_flow_cluster_inject(2)
    with pytest.raises(Exception) as ei:
        gm.generate_batch(prompts[:3], [cb] * 3)
    assert "hand-off timed out" in str(ei.value), str(ei.value)
    v.close()
    gm.close()


def test_a_timed_out_hand_off_in_a_staged_step_redoes_the_frame_from_the_intact_state(pkg, full):
    """Model.SampleNextLatentStateful through the staged API (ptts_batch_step): the transformer part of a step (keys, values, out_norm's rows, the EOS logit)
    does not go through the cluster, so a timed-out hand-off costs only the frame -- re-issued as launches from the saved Euler state; the step returns the
    values of an undisturbed step and the cache offsets advance once."""
    cfg, path, voice = full
    outs = []
    for inject in (0, 4):
        gm = pkg.Model.open(path, device=0, weights=pkg.WEIGHTS_BF16, kv=pkg.KV_BF16, max_batch=16)
        b = gm.new_batch(5, 256)
        vs = pkg.VoiceModelState(voice)
        for sl in range(5):
            b.set_voice_state(sl, vs)
        toks = pkg.synth.make_prompts(5, 25, 4000, seed=8)
        b.prompt([gm.text_embeddings(t) for t in toks])
        frames = np.full((5, 32), np.nan, np.float32)
        f1, l1, _ = b.step(frames)
        if inject:
            gm.debug_flow_cluster_inject(inject)
        f2, l2, _ = b.step(f1)
        f3, l3, _ = b.step(f2)
        assert list(b.offsets()) == [125 + 25 + 3] * 5
        outs.append((f1, f2, f3, l1, l2, l3))
        b.close()
        gm.close()
    for a, c in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, c)


def test_two_euler_steps_per_frame_run_the_cluster_twice_and_match_the_launches(pkg, full):
    """lsd_decode_steps = 2 (flow_lm.go:311-353): the flow net runs twice per frame, the second time on the first's Euler state -- two k_flow_cluster launches per
    AR step whose exchange tags follow on from each other.  Against the 2 x 12 launches, free-running over 5 frames: the same bits."""
    cfg, path, voice = full
    prompts = [p.tolist() for p in pkg.synth.make_prompts(9, 25, 4000, seed=31)]
    outs = []
    for cluster in ("1", "0"):
        os.environ["PTTS_FLOW_CLUSTER"] = cluster
        try:
            gm = pkg.Model.open(path, device=0, weights=pkg.WEIGHTS_BF16, kv=pkg.KV_BF16, max_batch=16)
            v = gm.upload_voice(pkg.VoiceModelState(voice))
            c = pkg.RuntimeGenerateConfig(max_steps=5, lsd_decode_steps=2, eos_threshold=float("inf"), frames_after_eos=3, device_voice=v, want_latents=True)
            pkg.runtime.launch_counts(True)
            outs.append(gm.generate_batch(prompts, [c] * 9))
            counts = pkg.runtime.launch_counts(False)
            assert counts.get("k_flow_cluster", 0) == (10 if cluster == "1" else 0), counts
            v.close()
            gm.close()
        finally:
            os.environ.pop("PTTS_FLOW_CLUSTER", None)
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a.latents, b.latents)
