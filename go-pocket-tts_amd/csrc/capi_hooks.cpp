// capi_hooks.cpp -- extern "C" test and measurement hooks (include/ptts_debug.h).  NOT part of libptts_hip.so: the Makefile links this file into
// libptts_hooks.so, which depends on libptts_hip.so and is loaded by the tests, tools/ and bench.py's measurement passes only.  The library a host of the
// reference links therefore exports no fault injection, no micro-benchmarks and no launch census.
#include "capi_internal.h"
#include "../../include/ptts_debug.h"

using namespace ptts;
using namespace ptts::capi;

extern "C" {

int ptts_decode_stages(ptts_model* h, const float* latents, int32_t n_utt, int32_t frames, float* pcm, float* mimi_latent, float* transformer_out) {
    return decode_stages(h, latents, n_utt, frames, pcm, mimi_latent, transformer_out);
}

const char* ptts_debug_last_attention_kernel(void) { return g_last_attn_kernel; }

int ptts_debug_flow_cluster_inject(ptts_model* h, int32_t block) {
    return guard([&] {
        if (!h || !h->m) throw Error(PTTS_EINVAL, "native-safetensors runtime unavailable");
        if (block < 0 || block > h->m->d.flow_depth) throw Error(PTTS_EINVAL, "ptts-hip: flow-net block out of range");
        std::lock_guard<std::mutex> lock(h->m->mu);
        h->m->fc_inject = block;
    });
}

int64_t ptts_debug_launch_counts(int32_t on, char* out, int64_t cap) {
    static thread_local std::map<std::string, int64_t> census;
    std::string s;
    for (const auto& kv : census) s += kv.first + "=" + std::to_string(kv.second) + ";";
    if (out && cap > 0) {
        const size_t n = std::min<size_t>(s.size(), (size_t)cap - 1);
        std::memcpy(out, s.data(), n);
        out[n] = 0;
    }
    census.clear();
    g_launch_census = on ? &census : nullptr;
    return (int64_t)s.size();
}

// timing aid (tools/microbench.py): `iters` back-to-back launches of the step linear on random operands
int ptts_debug_time_skinny(int32_t M, int32_t N, int32_t K, int32_t w_bf16, int32_t splitk, int32_t fuse_ln, int32_t iters, float* avg_us) {
    return guard([&] {
        require_device();
        Tmp dA((size_t)M * K * 4), dW((size_t)N * K * 4), dC((size_t)M * N * 4 * (splitk > 1 ? splitk : 1)), dlnw((size_t)K * 4);
        PTTS_HIP(hipMemset(dA.p, 0x3c, (size_t)M * K * 4));
        PTTS_HIP(hipMemset(dW.p, 0x3c, (size_t)N * K * (w_bf16 ? 2 : 4)));
        PTTS_HIP(hipMemset(dlnw.p, 0x3c, (size_t)K * 4));
        GemmArgs g;
        g.A = dA.as<float>(); g.amap = RowMap{K, 0, 0};
        g.W = dW.p; g.w_bf16 = w_bf16; g.ldw = K;
        const size_t wt_bytes = (size_t)((N + 15) / 16) * ((K + 127) / 128) * 16 * 128 * (w_bf16 ? 2 : 4);
        Tmp dWt(wt_bytes);
        PTTS_HIP(hipMemset(dWt.p, 0x3c, wt_bytes));
        g.Wt = dWt.p;
        g.C = dC.as<float>(); g.cmap = RowMap{N, 0, 0};
        g.M = M; g.N = N; g.K = K;
        SkinnyFuse fu;
        if (fuse_ln) { fu.ln = 1; fu.ln_w = dlnw.as<float>(); fu.ln_b = dlnw.as<float>(); }
        if (!(fuse_ln ? skinny_fuse_supported(g, fu) : skinny_supported(g, splitk))) throw Error(PTTS_EINVAL, "shape not supported");
        hipEvent_t e0, e1;
        PTTS_HIP(hipEventCreate(&e0)); PTTS_HIP(hipEventCreate(&e1));
        for (int i = 0; i < 5; i++) launch_skinny(g, fu, splitk, dC.as<float>(), nullptr);
        PTTS_HIP(hipDeviceSynchronize());
        PTTS_HIP(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; i++) launch_skinny(g, fu, splitk, dC.as<float>(), nullptr);
        PTTS_HIP(hipEventRecord(e1, nullptr));
        PTTS_HIP(hipEventSynchronize(e1));
        float ms = 0;
        PTTS_HIP(hipEventElapsedTime(&ms, e0, e1));
        *avg_us = ms * 1e3f / (float)iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    });
}

// debug: one stamped launch of the step linear (after warm-up); out receives 8 ticks per block, *n_blocks the block count
int ptts_debug_skinny_stamps(int32_t M, int32_t N, int32_t K, int32_t w_bf16, int32_t splitk, int32_t fuse_ln, uint64_t* out, int32_t max_blocks,
                             int32_t* n_blocks) {
    return guard([&] {
        require_device();
        Tmp dA((size_t)M * K * 4), dW((size_t)N * K * 4), dC((size_t)M * N * 4 * (splitk > 1 ? splitk : 1)), dlnw((size_t)K * 4);
        PTTS_HIP(hipMemset(dA.p, 0x3c, (size_t)M * K * 4));
        PTTS_HIP(hipMemset(dW.p, 0x3c, (size_t)N * K * (w_bf16 ? 2 : 4)));
        PTTS_HIP(hipMemset(dlnw.p, 0x3c, (size_t)K * 4));
        GemmArgs g;
        g.A = dA.as<float>(); g.amap = RowMap{K, 0, 0};
        g.W = dW.p; g.w_bf16 = w_bf16; g.ldw = K;
        const size_t wt_bytes = (size_t)((N + 15) / 16) * ((K + 127) / 128) * 16 * 128 * (w_bf16 ? 2 : 4);
        Tmp dWt(wt_bytes);
        PTTS_HIP(hipMemset(dWt.p, 0x3c, wt_bytes));
        g.Wt = dWt.p;
        g.C = dC.as<float>(); g.cmap = RowMap{N, 0, 0};
        g.M = M; g.N = N; g.K = K;
        SkinnyFuse fu;
        if (fuse_ln) { fu.ln = 1; fu.ln_w = dlnw.as<float>(); fu.ln_b = dlnw.as<float>(); }
        if (!(fuse_ln ? skinny_fuse_supported(g, fu) : skinny_supported(g, splitk))) throw Error(PTTS_EINVAL, "shape not supported");
        const int blocks = ((N + 15) / 16) * ((M + 15) / 16) * (splitk > 1 ? splitk : 1);   // upper bound: the narrow-block variant has N/16 column blocks
        Tmp dS((size_t)blocks * 8 * 8);
        PTTS_HIP(hipMemset(dS.p, 0, (size_t)blocks * 64));
        for (int i = 0; i < 3; i++) launch_skinny(g, fu, splitk, dC.as<float>(), nullptr);
        PTTS_HIP(hipDeviceSynchronize());
        g_skinny_stamps = reinterpret_cast<unsigned long long*>(dS.p);
        launch_skinny(g, fu, splitk, dC.as<float>(), nullptr);
        g_skinny_stamps = nullptr;
        PTTS_HIP(hipDeviceSynchronize());
        *n_blocks = blocks;
        down(out, dS.p, (size_t)std::min(blocks, max_blocks) * 64);
    });
}

// debug: one whole AR step of a prompted batch with every stampable launch of the step linear stamped in place (cold caches, the
// real operands).  out: [cap_blocks][8] ticks; desc: [cap_desc][8] = M, N, K, prologue, NJ, CG, blocks, splitk per launch
int ptts_debug_step_stamps(ptts_batch* hb, int32_t lsd_steps, uint64_t* out, int64_t cap_blocks, int32_t* desc, int32_t cap_desc, int32_t* n_desc) {
    return guard([&] {
        if (!hb || !hb->b || !out || !desc || !n_desc) throw Error(PTTS_EINVAL, "ptts-hip: null argument");
        Model& m = *hb->m;
        Batch& b = *hb->b;
        std::lock_guard<std::mutex> lock(m.mu);
        m.use_device();
        for (int i = 0; i < b.B; i++)
            if (b.kv_len_host[i] + 1 > b.cap) throw Error(PTTS_EINVAL, "ptts-hip: KV capacity exhausted");
        m.tcomb_for(lsd_steps);
        Tmp ds((size_t)cap_blocks * 64);
        PTTS_HIP(hipMemsetAsync(ds.p, 0, (size_t)cap_blocks * 64, m.stream));
        PTTS_HIP(hipMemsetAsync(b.cur.p, 0, (size_t)b.B * m.d.ldim * sizeof(float), m.stream));
        SkinnyStampLog lg;
        lg.base = ds.as<unsigned long long>(); lg.cap_blocks = (size_t)cap_blocks;
        g_skinny_stamp_log = &lg;
        try { step_core(b, lsd_steps); } catch (...) { g_skinny_stamp_log = nullptr; throw; }
        g_skinny_stamp_log = nullptr;
        for (int i = 0; i < b.B; i++) b.kv_len_host[i] += 1;
        PTTS_HIP(hipMemcpyAsync(b.st.kv_len, b.kv_len_host.data(), (size_t)b.B * sizeof(int32_t), hipMemcpyHostToDevice, m.stream));
        PTTS_HIP(hipStreamSynchronize(m.stream));
        if (b.fc_ok) {   // (a stamped step whose cluster hand-off timed out measured nothing)
            unsigned fault = 0;
            down(&fault, b.fc_fault(), sizeof fault);
            if (fault) flow_cluster_fault(b);
        }
        down(out, ds.p, lg.used_blocks * 64);
        const int n = (int)std::min<size_t>(lg.desc.size(), (size_t)cap_desc);
        for (int i = 0; i < n; i++) std::memcpy(desc + 8 * i, &lg.desc[(size_t)i], 32);
        *n_desc = n;
    });
}

// debug: time one many-row GEMM variant (2 = k_gemm2, 3 = k_gemm3) and compare it with the other one on pseudo-random data
int ptts_debug_gemm(int32_t M, int32_t N, int32_t K, int32_t w_bf16, int32_t variant, int32_t epi_flags, int32_t iters, float* avg_us, float* maxdiff) {
    return guard([&] {
        require_device();
        // epi_flags: the epilogue form in the low byte; 0x100: RoPE on the first two thirds of the columns (positions restart every
        // 2000 rows: the decoder's qkv projection); 0x200: the residual is read from the output buffer itself (the decoder's
        // out_proj / linear2), 0x400: activations behind a prologue ELU, 0x800: a per-column scale
        const int epi = epi_flags & 0xff;
        const bool rope = epi_flags & 0x100, inplace = epi_flags & 0x200;
        const size_t na = (size_t)M * K, nw = (size_t)N * K, nc1 = (size_t)M * N;
        const int S = (epi_flags & 0x4000) ? K / 1024 : 1;   // 0x4000: split-K in 1024-deep slices (raw sums, plane z at C + z M N)
        if (S < 1 || (S > 1 && K % 1024)) throw Error(PTTS_EINVAL, "split-K probe needs K % 1024 == 0");
        const size_t nc = nc1 * (size_t)S;       // values compared: every plane
        std::vector<float> ha(na), hw(nw), hb((size_t)N);
        uint32_t st = 12345u;
        auto rnd = [&] { st = st * 1664525u + 1013904223u; return ((float)(st >> 8) / 8388608.0f) - 1.0f; };
        for (auto& x : ha) x = rnd();
        for (auto& x : hw) x = rnd() * 0.05f;
        for (auto& x : hb) x = rnd();
        Tmp dA(na * 4), dW(nw * 4), dB((size_t)N * 4), dC(nc * 4), dC2(nc * 4), dR(nc * 4), dCos(2000 * 32 * 4), dSin(2000 * 32 * 4);
        up(dA.p, ha.data(), na * 4); up(dB.p, hb.data(), (size_t)N * 4);
        if (w_bf16) {
            std::vector<uint16_t> hw16(nw);
            for (size_t i = 0; i < nw; i++) { uint32_t u; memcpy(&u, &hw[i], 4); hw16[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
            up(dW.p, hw16.data(), nw * 2);
        } else up(dW.p, hw.data(), nw * 4);
        if (inplace) {   // a residual with content: the first nc values of the activations' generator, continued
            std::vector<float> hr(nc);
            for (auto& x : hr) x = rnd();
            up(dR.p, hr.data(), nc * 4);
        } else PTTS_HIP(hipMemset(dR.p, 0, nc * 4));
        if (rope) {
            std::vector<float> hc(2000 * 32), hs(2000 * 32);
            for (size_t i = 0; i < hc.size(); i++) { const float ang = rnd() * 3.14159265f; hc[i] = std::cos(ang); hs[i] = std::sin(ang); }
            up(dCos.p, hc.data(), hc.size() * 4); up(dSin.p, hs.data(), hs.size() * 4);
        }
        GemmArgs g;
        g.A = dA.as<float>(); g.amap = RowMap{K, 0, 0};
        g.W = dW.p; g.w_bf16 = w_bf16; g.ldw = K; g.bias = dB.as<float>();
        g.C = dC.as<float>(); g.cmap = RowMap{N, 0, 0};
        g.R = dR.as<float>(); g.epi = epi;
        g.M = M; g.N = N; g.K = K;
        if (epi_flags & 0x400) g.aop = AOP_ELU;
        if (S > 1) { g.kslice = 1024; g.zstride = (int64_t)nc1; g.bias = nullptr; }
        if (epi_flags & 0x800) g.scale = dB.as<float>();   // a per-column scale (the decoder's layer scale): the bias values serve
        if (rope) {
            g.rope_cos = dCos.as<float>(); g.rope_sin = dSin.as<float>(); g.rope_cols = N / 3 * 2; g.rope_hd = 64; g.rope_pos0 = 0; g.rope_rows_per_seg = 2000;
            g.bias = nullptr;
        }
        if (!gemm3_supported(g)) throw Error(PTTS_EINVAL, "shape not supported");
        // variant 2: k_gemm2, 3: k_gemm3 (30 + cfg: a forced shape), 40: whatever launch_gemm dispatches (k_gemm_wres where it applies),
        // 50 + cfg: k_gemm5
        auto run = [&](int v, float* c) {
            GemmArgs h = g; h.C = c;
            if (inplace) { PTTS_HIP(hipMemcpyAsync(c, dR.p, nc * 4, hipMemcpyDeviceToDevice, nullptr)); h.R = c; }
            if (v == 40) { if (rope) { if (!launch_gemm_rope(h, nullptr)) throw Error(PTTS_EINVAL, "no RoPE epilogue for this shape"); } else launch_gemm(h, nullptr); }
            else if (v >= 50) { if (!gemm5_supported(h)) throw Error(PTTS_EINVAL, "shape not supported by k_gemm5"); g_gemm5_cfg = v - 50; launch_gemm5(h, nullptr); g_gemm5_cfg = 0; }
            else if (v >= 3) { g_gemm3_cfg = v >= 30 ? v - 30 : 0; launch_gemm3(h, nullptr); g_gemm3_cfg = 0; }
            else { if (!gemm2_supported(h)) throw Error(PTTS_EINVAL, "shape not supported by k_gemm2"); launch_gemm2(h, nullptr); }
        };
        hipEvent_t e0, e1;
        PTTS_HIP(hipEventCreate(&e0)); PTTS_HIP(hipEventCreate(&e1));
        for (int i = 0; i < 2; i++) run(variant, dC.as<float>());
        PTTS_HIP(hipDeviceSynchronize());
        PTTS_HIP(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; i++) run(variant, dC.as<float>());
        PTTS_HIP(hipEventRecord(e1, nullptr));
        PTTS_HIP(hipEventSynchronize(e1));
        float ms = 0;
        PTTS_HIP(hipEventElapsedTime(&ms, e0, e1));
        *avg_us = ms * 1e3f / (float)iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *maxdiff = -1.0f;
        if (nc <= ((size_t)200 << 20)) {
            // against k_gemm3 (k_gemm2 for k_gemm3 itself): the same k order, so equal bits are expected; and the variant against itself,
            // three more runs (a race shows as a difference between runs)
            run(variant >= 40 ? 3 : (variant >= 3 ? 2 : 3), dC2.as<float>());
            PTTS_HIP(hipDeviceSynchronize());
            std::vector<float> c1(nc), c2(nc);
            down(c1.data(), dC.p, nc * 4); down(c2.data(), dC2.p, nc * 4);
            float md = 0;
            size_t nbad = 0;
            for (size_t i = 0; i < nc; i++) {
                float d = std::fabs(c1[i] - c2[i]);
                if (!(d <= md)) md = d;
                if (d != 0 && nbad++ < 12) fprintf(stderr, "ptts_debug_gemm: variant %d vs reference at plane %zu row %zu column %zu: %.9g vs %.9g\n", variant, i / nc1, (i % nc1) / N, i % N, c1[i], c2[i]);
            }
            if (nbad) fprintf(stderr, "ptts_debug_gemm: %zu of %zu values differ from the reference kernel's\n", nbad, nc);
            for (int rep = 0; rep < 3; rep++) {
                run(variant, dC2.as<float>());
                PTTS_HIP(hipDeviceSynchronize());
                down(c2.data(), dC2.p, nc * 4);
                if (memcmp(c1.data(), c2.data(), nc * 4) != 0) {
                    size_t bad = 0, first = nc;
                    for (size_t i = 0; i < nc; i++) if (memcmp(&c1[i], &c2[i], 4) != 0) { if (first == nc) first = i; bad++; }
                    fprintf(stderr, "ptts_debug_gemm: variant %d differs from itself between runs: %zu of %zu values, first at row %zu column %zu\n", variant, bad, nc, first / N, first % N);
                    md = 1e30f;
                }
            }
            *maxdiff = md;
        }
    });
}

int ptts_debug_tall_linear(int32_t M, int32_t N, int32_t K, int32_t epi, int32_t splitk, const float* x, const float* planes, int32_t psplit, const float* pbias,
                           const float* ln_w, const float* ln_b, float eps, const float* W, const float* bias, const float* R, int32_t out_planes, float* out,
                           float* x_out) {
    return guard([&] {
        require_device();
        if (!x || !W || !out || M <= 0 || N <= 0 || K <= 0 || K % 128) throw Error(PTTS_EINVAL, "ptts_debug_tall_linear: bad arguments");
        const size_t mk = (size_t)M * K, mn = (size_t)M * N;
        const int S = std::max(1, (int)splitk);
        auto bf16 = [](float v) { uint32_t u; memcpy(&u, &v, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); };
        auto f32 = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float v; memcpy(&v, &u, 4); return v; };
        // the weights in the step kernels' fragment order (model.cpp add_tiled): [16-column tile][128-deep super-step][4][64 lanes] x 8 bf16
        const size_t nt = ((size_t)N + 15) / 16, nss = (size_t)K / 128;
        std::vector<uint16_t> wt(nt * nss * 4 * 64 * 8);
        for (size_t t = 0; t < nt; t++)
            for (size_t ss = 0; ss < nss; ss++)
                for (int sidx = 0; sidx < 4; sidx++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++) {
                            const size_t n = t * 16 + (size_t)(lane & 15), k = ss * 128 + (size_t)(lane >> 4) * 32 + (size_t)sidx * 8 + j;
                            wt[((((t * nss + ss) * 4 + sidx) * 64 + lane) * 8 + j)] = n < (size_t)N ? bf16(W[n * K + k]) : 0;
                        }
        Tmp dW(wt.size() * 2), dAh(mk * 2), dAl(mk * 2), dOut((size_t)S * mn * 4), dCh(mn * 2), dCl(mn * 2);
        up(dW.p, wt.data(), wt.size() * 2);
        Tmp dB((size_t)N * 4), dR(mn * 4), dX(mk * 4), dXo(mk * 4), dP(std::max<size_t>(1, (size_t)std::max(0, (int)psplit)) * mk * 4), dPb((size_t)K * 4), dLw((size_t)K * 4), dLb((size_t)K * 4);
        if (bias) up(dB.p, bias, (size_t)N * 4);
        if (R) up(dR.p, R, mn * 4);
        if (ln_w) {   // the rows through k_rowprep: split-K planes + residual -> LayerNorm -> bf16 planes
            up(dX.p, x, mk * 4); up(dLw.p, ln_w, (size_t)K * 4); up(dLb.p, ln_b, (size_t)K * 4);
            PrepArgs pa;
            pa.x = dX.as<float>(); pa.ldx = K; pa.ln_w = dLw.as<float>(); pa.ln_b = dLb.as<float>(); pa.eps = eps;
            if (planes && psplit > 0) {
                up(dP.p, planes, (size_t)psplit * mk * 4);
                pa.partial = dP.as<float>(); pa.psplit = psplit; pa.pstride = (int64_t)mk; pa.x_out = dXo.as<float>();
                if (pbias) { up(dPb.p, pbias, (size_t)K * 4); pa.pbias = dPb.as<float>(); }
            }
            pa.yh = dAh.as<uint16_t>(); pa.yl = dAl.as<uint16_t>(); pa.ldy = K; pa.M = M; pa.D = K;
            if (!rowprep_supported(pa)) throw Error(PTTS_EINVAL, "ptts_debug_tall_linear: rows not taken by k_rowprep");
            launch_rowprep(pa, nullptr);
        } else {      // the rows as they are: split on the host the way the kernels split (hi = bf16(x), lo = bf16(x - hi))
            std::vector<uint16_t> hi(mk), lo(mk);
            for (size_t i = 0; i < mk; i++) { hi[i] = bf16(x[i]); lo[i] = bf16(x[i] - f32(hi[i])); }
            up(dAh.p, hi.data(), mk * 2); up(dAl.p, lo.data(), mk * 2);
        }
        TallArgs t;
        t.ah = dAh.as<uint16_t>(); t.al = dAl.as<uint16_t>(); t.lda = K; t.Wt = dW.p; t.bias = bias ? dB.as<float>() : nullptr;
        t.R = R ? dR.as<float>() : nullptr; t.ldr = N; t.M = M; t.N = N; t.K = K; t.epi = epi;
        if (S > 1) { t.splitk = S; t.partial = dOut.as<float>(); t.zstride = (int64_t)mn; }
        else if (out_planes) { t.ch = dCh.as<uint16_t>(); t.cl = dCl.as<uint16_t>(); t.ldp = N; }
        else { t.C = dOut.as<float>(); t.ldc = N; }
        if (!tall_supported(t)) throw Error(PTTS_EINVAL, "ptts_debug_tall_linear: shape not taken by k_tall");
        PTTS_HIP(hipMemset(dOut.p, 0xff, (size_t)S * mn * 4));   // (NaN where the kernel stores nothing)
        launch_tall(t, nullptr);
        PTTS_HIP(hipDeviceSynchronize());
        if (S == 1 && out_planes) {
            std::vector<uint16_t> hi(mn), lo(mn);
            down(hi.data(), dCh.p, mn * 2); down(lo.data(), dCl.p, mn * 2);
            for (size_t i = 0; i < mn; i++) out[i] = f32(hi[i]) + f32(lo[i]);
        } else down(out, dOut.p, (size_t)S * mn * 4);
        if (x_out && ln_w && planes && psplit > 0) down(x_out, dXo.p, mk * 4);
    });
}

int ptts_mimi_layer_piece(ptts_model* h, int32_t layer, int32_t which, const float* x, int64_t rows, int32_t pos0, int32_t rows_per_seg, float* out) {
    return guard([&] {
        if (!h || !h->m) throw Error(PTTS_EINVAL, "native-safetensors runtime unavailable");
        Model& m = *h->m;
        const Desc& d = m.d;
        if (!x || !out || rows <= 0 || rows > (1 << 24)) throw Error(PTTS_EINVAL, "ptts-hip: bad rows");
        if (layer < 0 || layer >= d.mimi_layers) throw Error(PTTS_EINVAL, strfmt("ptts-hip: mimi layer %d out of range [0,%d)", layer, d.mimi_layers));
        if (which != PTTS_MIMI_PIECE_QKV && which != PTTS_MIMI_PIECE_FFN) throw Error(PTTS_EINVAL, "ptts-hip: unknown layer piece");
        if (pos0 < 0 || rows_per_seg < 0 || (int64_t)pos0 + (rows_per_seg ? rows_per_seg : rows) > ROPE_SEQ)
            throw Error(PTTS_EINVAL, strfmt("ops: rope cos/sin sequence length too small for pos=%d seq=%lld", pos0, (long long)(rows_per_seg ? rows_per_seg : rows)));
        std::lock_guard<std::mutex> lock(m.mu);
        m.use_device();
        const int C = d.mimi_dim, F = d.mimi_ffn, R = (int)rows;
        const int NO = which == PTTS_MIMI_PIECE_QKV ? 3 * C : C;
        Tmp dx((size_t)R * C * 4), dn((size_t)R * C * 4), dy((size_t)R * std::max(NO, F) * 4);
        PTTS_HIP(hipMemcpyAsync(dx.p, x, (size_t)R * C * 4, hipMemcpyHostToDevice, m.stream));
        if (which == PTTS_MIMI_PIECE_QKV) {
            mimi_layer_qkv(m, layer, dx.as<float>(), RowMap{C, 0, 0}, R, dy.as<float>(), RowMap{3 * C, 0, 0}, pos0, rows_per_seg, dn.as<float>(), m.stream);
            PTTS_HIP(hipMemcpyAsync(out, dy.p, (size_t)R * NO * 4, hipMemcpyDeviceToHost, m.stream));
        } else {
            mimi_layer_ffn(m, layer, dx.as<float>(), RowMap{C, 0, 0}, R, dn.as<float>(), dy.as<float>(), m.stream);
            PTTS_HIP(hipMemcpyAsync(out, dx.p, (size_t)R * NO * 4, hipMemcpyDeviceToHost, m.stream));
        }
        PTTS_HIP(hipStreamSynchronize(m.stream));
    });
}

}  // extern "C"
