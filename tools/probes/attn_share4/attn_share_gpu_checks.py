"""The step attention for groups of four slots on one voice (k_attn_step4, attn_step.hip; reference: flow_transformer.go:340-347 +
attention.go:307-484 at one query per utterance): the block shares the voice's keys and values through LDS, every utterance keeps
k_attn_step's assignment of keys to rounds, waves and lanes -- so a batch must get the same BITS from either kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _modules(tensors):
    mods = {}
    for name, t in tensors.items():
        mod, key = name.rsplit("/", 1)
        mods.setdefault(mod, {})[key] = np.asarray(t, np.float32 if key == "cache" else np.int64)
    return mods


@pytest.mark.parametrize("kv_bf16", [False, True])
@pytest.mark.parametrize("graph", [False, True])
def test_shared_voice_groups_give_the_per_utterance_kernels_bits(pkg, tmp_path, kv_bf16, graph):
    """Six requests (a full group of four and a partial one), ragged prompts, 9 steps, a 37-key voice (one whole round of 32 keys in bf16 --
    two of 16 in f32 -- inside the prefix, then a mixed round): all on ONE uploaded voice -> k_attn_step4; the same requests on two uploads
    of the same voice, alternating -> no group shares a pointer -> k_attn_step.  Latents and samples must be identical."""
    synth = pkg.synth
    cfg = synth.SynthConfig.tiny()
    path = str(tmp_path / "tiny.safetensors")
    synth.write_safetensors(path, synth.make_checkpoint(cfg, seed=77))
    gm = pkg.Model.open(path, device=0, max_batch=8, kv=pkg.KV_BF16 if kv_bf16 else pkg.KV_F32)
    gm.set_use_graph(graph)
    voice = pkg.VoiceModelState(_modules(synth.make_voice_state(cfg, offset=37, seed=5)))
    dv1, dv2 = gm.upload_voice(voice), gm.upload_voice(voice)
    rng = np.random.default_rng(3)
    prompts = [rng.integers(1, cfg.n_bins, size=int(rng.integers(3, 9))).astype(np.int64) for _ in range(6)]

    def run(voices):
        cfgs = [pkg.RuntimeGenerateConfig(max_steps=9, eos_threshold=1e30, want_latents=True, device_voice=v) for v in voices]
        pkg.runtime.launch_counts(True)
        out = gm.generate_batch(prompts, cfgs)
        return out, pkg.runtime.launch_counts(False)

    shared, c1 = run([dv1] * 6)
    split, c2 = run([dv1, dv2] * 3)
    if not graph:   # (a replayed graph launches nothing through the census after its capture)
        assert c1.get("k_attn_step4", 0) > 0 and c1.get("k_attn_step", 0) == 0, c1
        assert c2.get("k_attn_step", 0) > 0 and c2.get("k_attn_step4", 0) == 0, c2
    for i in range(6):
        assert shared[i].n_frames == split[i].n_frames == 9
        assert np.array_equal(shared[i].latents, split[i].latents), (i, float(np.abs(shared[i].latents - split[i].latents).max()))
        assert np.array_equal(shared[i].pcm, split[i].pcm), i
    dv1.close(); dv2.close(); gm.close()
