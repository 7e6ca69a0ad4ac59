/* A C host of libptts_hip.so with neither Python nor PyTorch in the process: what the reference's Go service binds through cgo
 * (tts.NewService -> NewNativeSafetensorsRuntime, internal/tts/service.go:39-98, runtime_native_safetensors.go:36-38;
 * one GenerateAudio per chunk, internal/tts/runtime.go:42-45).  Plain C99 against include/ptts.h, linked with -lptts_hip, so the HIP
 * runtime it runs on is the system's /opt/rocm libamdhip64 (the library's own DT_NEEDED), not the copy PyTorch bundles.
 *
 *   smoke <checkpoint.safetensors> <case.bin> <out.bin> [voice.safetensors]
 *
 * case.bin (little endian), written by tests/test_gpu_c_host.py:
 *   int32 n_layers, T, H, D; int64 offsets[n_layers]; float caches[n_layers][2*T*H*D]      -- a voice model state
 *   int32 n_reqs; per request: int32 n_tokens, max_steps, use_voice; int64 tokens[n_tokens]
 *     use_voice 0: none; 1: the arrays above in the request; 2: the voice FILE of argv[4], read and uploaded by the library itself
 *     (ptts_voice_open: safetensors.LoadVoiceModelState + initStateFromVoiceModelState, reader.go:127-140, flow_transformer.go:451-480)
 * Calls: request 0 alone (n_reqs = 1 must reproduce GenerateAudio), then requests 1.. as ONE batched ptts_generate.
 * out.bin: per request int32 status, n_frames, eos_step, ldim; int64 n_samples; float pcm[n_samples]; float latents[n_frames*ldim]. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ptts.h"

static void die(const char* what) {
    fprintf(stderr, "smoke: %s: %s\n", what, ptts_last_error());
    exit(2);
}

static void rd(void* dst, size_t n, FILE* f) {
    if (fread(dst, 1, n, f) != n) { fprintf(stderr, "smoke: short read of case file\n"); exit(3); }
}

int main(int argc, char** argv) {
    ptts_opts opts;
    ptts_model* model = NULL;
    ptts_voice* file_voice = NULL;
    ptts_voice_file* vfile = NULL;
    ptts_info info;
    FILE* f;
    FILE* out;
    int32_t hdr[4], n_reqs, i, l;
    int64_t* offsets;
    int64_t* steps;
    float** caches;
    const float** ccaches;
    ptts_request* reqs;
    ptts_result* res;
    int64_t** toks;
    size_t per;

    if (argc != 4 && argc != 5) { fprintf(stderr, "usage: smoke checkpoint case.bin out.bin [voice.safetensors]\n"); return 1; }
    ptts_default_opts(&opts);
    opts.device = 0;
    if (ptts_model_open(argv[1], &opts, &model) != PTTS_OK) die("ptts_model_open");
    if (ptts_model_info(model, &info) != PTTS_OK) die("ptts_model_info");
    if (argc == 5) {
        /* what tts.loadVoiceConditioning does with a voice path (service.go:216-246): inspect, then load by kind */
        if (ptts_voice_file_open(argv[4], &vfile) != PTTS_OK) die("ptts_voice_file_open");
        if (ptts_voice_file_kind(vfile) != PTTS_VOICE_FILE_MODEL_STATE) { fprintf(stderr, "smoke: voice file kind %d\n", (int)ptts_voice_file_kind(vfile)); return 1; }
        ptts_voice_file_close(vfile);
        if (ptts_voice_open(model, argv[4], &file_voice) != PTTS_OK) die("ptts_voice_open");
    }

    f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 1; }
    rd(hdr, sizeof hdr, f);
    if (hdr[0] != (int32_t)info.n_layers) { fprintf(stderr, "smoke: voice has %d layers, model %d\n", hdr[0], (int)info.n_layers); return 1; }
    offsets = (int64_t*)malloc(sizeof(int64_t) * (size_t)hdr[0]);
    steps = (int64_t*)malloc(sizeof(int64_t) * (size_t)hdr[0]);
    caches = (float**)malloc(sizeof(float*) * (size_t)hdr[0]);
    ccaches = (const float**)malloc(sizeof(float*) * (size_t)hdr[0]);
    rd(offsets, sizeof(int64_t) * (size_t)hdr[0], f);
    per = (size_t)2 * (size_t)hdr[1] * (size_t)hdr[2] * (size_t)hdr[3];
    for (l = 0; l < hdr[0]; l++) {
        caches[l] = (float*)malloc(per * sizeof(float));
        rd(caches[l], per * sizeof(float), f);
        ccaches[l] = caches[l];
        steps[l] = hdr[1];
    }
    rd(&n_reqs, sizeof n_reqs, f);
    reqs = (ptts_request*)calloc((size_t)n_reqs, sizeof(ptts_request));
    res = (ptts_result*)calloc((size_t)n_reqs, sizeof(ptts_result));
    toks = (int64_t**)malloc(sizeof(int64_t*) * (size_t)n_reqs);
    for (i = 0; i < n_reqs; i++) {
        int32_t h3[3];
        rd(h3, sizeof h3, f);
        toks[i] = (int64_t*)malloc(sizeof(int64_t) * (size_t)h3[0]);
        rd(toks[i], sizeof(int64_t) * (size_t)h3[0], f);
        reqs[i].tokens = toks[i];
        reqs[i].n_tokens = h3[0];
        reqs[i].temperature = 0.0f;
        reqs[i].eos_threshold = 1e30f;       /* fixed-length run: never EOS */
        reqs[i].max_steps = h3[1];
        reqs[i].lsd_steps = 1;
        reqs[i].frames_after_eos = 3;
        reqs[i].want_latents = 1;
        if (h3[2] == 2) {
            if (!file_voice) { fprintf(stderr, "smoke: request %d wants the voice file, none given\n", (int)i); return 1; }
            reqs[i].voice = file_voice;
        } else if (h3[2]) {
            reqs[i].voice_caches = ccaches;
            reqs[i].voice_cache_steps = steps;
            reqs[i].voice_offsets = offsets;
        }
    }
    fclose(f);

    if (ptts_generate(model, &reqs[0], 1, &res[0]) != PTTS_OK) die("ptts_generate (single)");
    if (n_reqs > 1 && ptts_generate(model, &reqs[1], n_reqs - 1, &res[1]) != PTTS_OK) die("ptts_generate (batch)");

    out = fopen(argv[3], "wb");
    if (!out) { perror(argv[3]); return 1; }
    for (i = 0; i < n_reqs; i++) {
        int32_t h4[4];
        h4[0] = res[i].status; h4[1] = res[i].n_frames; h4[2] = res[i].eos_step; h4[3] = (int32_t)info.ldim;
        fwrite(h4, sizeof h4, 1, out);
        fwrite(&res[i].n_samples, sizeof(int64_t), 1, out);
        if (res[i].status == PTTS_OK) {
            fwrite(res[i].pcm, sizeof(float), (size_t)res[i].n_samples, out);
            fwrite(res[i].latents, sizeof(float), (size_t)res[i].n_frames * (size_t)info.ldim, out);
        }
        ptts_free_result(&res[i]);
    }
    fclose(out);
    ptts_voice_free(file_voice);
    ptts_model_close(model);
    printf("smoke: %d requests, d_model %d, %d layers\n", (int)n_reqs, (int)info.d_model, (int)info.n_layers);
    return 0;
}
