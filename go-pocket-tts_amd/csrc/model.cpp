// model.cpp -- checkpoint -> device arena.
// Tensor names and optional/required rules follow the reference loaders:
//   flow_lm.go:51-119, flow_transformer.go:110-156,482-511 (no biases loaded for in_proj/out_proj/
//   linear1/linear2), flow_net.go:18-40,92-113,181-203,250-305, conditioner.go:17,
//   mimi.go:44-67,86-114,180-239,546-637, model.go:176-250 (projector fold), var_builder.go.
#include <cmath>
#include <functional>

#include "model.h"

namespace ptts {
namespace {

struct Walker {
    const StFile& f;
    Desc& d;
    bool bf16w;     // matrices are held as bf16 (PTTS_WEIGHTS_BF16 and PTTS_WEIGHTS_INT8: everything the AR step does not stream)
    uint8_t* host;  // null while planning
    bool i8w = false;   // PTTS_WEIGHTS_INT8: the matrices the AR step streams are per-row-scaled int8 in the step kernel's fragment order
    size_t cur = 0;

    size_t reserve(size_t bytes) {
        size_t off = (cur + 255) & ~(size_t)255;
        cur = off + bytes;
        return off;
    }
    bool has(const std::string& n) const { return f.has(n); }
    const std::vector<int64_t>& shape(const std::string& n) const { return f.at(n).shape; }
    std::vector<float> load(const std::string& n) const {
        std::vector<float> v((size_t)f.at(n).count());
        f.decode_f32(n, v.data());
        return v;
    }
    // f32 vector item
    size_t add_f32(size_t count, const std::function<void(float*)>& fill) {
        size_t off = reserve(count * 4);
        if (host) fill(reinterpret_cast<float*>(host + off));
        return off;
    }
    // matrix item, stored f32 or bf16
    size_t add_mat(size_t count, const std::function<void(float*)>& fill, int* is_bf16) {
        *is_bf16 = bf16w ? 1 : 0;
        size_t off = reserve(count * (bf16w ? 2 : 4));
        if (host) {
            if (bf16w) {
                std::vector<float> tmp(count);
                fill(tmp.data());
                uint16_t* dst = reinterpret_cast<uint16_t*>(host + off);
                for (size_t i = 0; i < count; i++) dst[i] = f32_to_bf16_rne(tmp[i]);
            } else fill(reinterpret_cast<float*>(host + off));
        }
        d.n_params += (int64_t)count;
        return off;
    }
    void expect_rank(const std::string& n, size_t rank) const {
        if (shape(n).size() != rank) throw Error(PTTS_EFORMAT, strfmt("native: tensor \"%s\" rank %zu, want %zu", n.c_str(), shape(n).size(), rank));
    }

    // Fragment-ordered copy for the AR-step kernel: per (16-row tile, 128-deep super-step) one block of WV KiB in which
    // MFMA step s of lane l reads 16 contiguous bytes at [s][l]: W[tile*16 + (l & 15)][ss*128 + (l >> 4)*32 + s*E .. +E),
    // E = 8 (bf16) or 4 (f32).  Every wave-level weight load of the step is then one contiguous 1-KiB burst.
    // ---- PTTS_WEIGHTS_INT8: weight-only int8 for the matrices the AR step streams ----
    // Row n of W becomes q[n][k] in [-127, 127] and a scale s[n] = max_k |W[n][k]| / 127 (1 for an all-zero row);
    // q = rint(W / s) (ties to even), the effective weight is W^ = q * s.  The step kernel streams q (1 byte per weight, stored
    // offset-binary q + 128, in its fragment order) and multiplies by s[n] in the epilogue; everything else that touches the
    // matrix (prefill GEMMs, the 32-wide linears of k_step_begin) reads a row-major f32 copy of W^, so that the whole
    // model computes with ONE set of weights -- the ones an f32 oracle is given in the parity tests.
    static void quantize_rows(std::vector<float>& rm, size_t out, size_t in, std::vector<float>& scale, std::vector<int8_t>& q) {
        scale.assign(out, 1.0f);
        q.assign(out * in, 0);
        for (size_t n = 0; n < out; n++) {
            float mx = 0.0f;
            for (size_t k = 0; k < in; k++) mx = std::max(mx, std::fabs(rm[n * in + k]));
            const float s = mx > 0.0f ? mx / 127.0f : 1.0f;
            scale[n] = s;
            for (size_t k = 0; k < in; k++) {
                float v = std::nearbyint(rm[n * in + k] / s);
                v = std::min(127.0f, std::max(-127.0f, v));
                q[n * in + k] = (int8_t)v;
                rm[n * in + k] = v * s;
            }
        }
    }
    // tiled int8 copy + scales of a row-major matrix (rm is replaced by W^).  Layout: per (16-row tile, 128-deep super-step) 2 KiB:
    // load u (0, 1) of lane l is 16 bytes = MFMA steps 2u and 2u + 1, byte j -> k = ss*128 + (l >> 4)*32 + (2u + j/8)*8 + j%8
    void add_tiled_i8(Lin& l, std::vector<float>* rm) {
        const size_t nt = ((size_t)l.out + 15) / 16, nss = ((size_t)l.in + 127) / 128;
        l.wt = reserve(nt * nss * 2048);
        l.wscale = reserve((size_t)l.out * 4);
        l.wt_i8 = 1;
        if (!host) return;
        std::vector<float> scale;
        std::vector<int8_t> q;
        quantize_rows(*rm, (size_t)l.out, (size_t)l.in, scale, q);
        std::memcpy(host + l.wscale, scale.data(), scale.size() * 4);
        uint8_t* dst = host + l.wt;
        for (size_t t = 0; t < nt; t++)
            for (size_t ss = 0; ss < nss; ss++)
                for (int u = 0; u < 2; u++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 16; j++) {
                            const size_t n = t * 16 + (size_t)(lane & 15), k = ss * 128 + (size_t)(lane >> 4) * 32 + (size_t)(2 * u + j / 8) * 8 + (size_t)(j % 8);
                            const int v = (n < (size_t)l.out && k < (size_t)l.in) ? q[n * l.in + k] : 0;
                            dst[((((t * nss + ss) * 2 + u) * 64 + lane) * 16) + j] = (uint8_t)(v + 128);
                        }
    }
    // a matrix the AR step streams, with its row-major copy (own_rowmajor) or only the tiled one (a stack of matrices whose
    // rows already exist row-major elsewhere)
    void add_step_matrix(Lin& l, const std::function<void(float*)>& fill_rowmajor, bool own_rowmajor) {
        const size_t count = (size_t)l.out * l.in;
        if (!i8w) {
            if (own_rowmajor) l.w = add_mat(count, fill_rowmajor, &l.bf16);
            add_tiled(l, fill_rowmajor);
            return;
        }
        std::vector<float> rm;
        if (host) { rm.resize(count); fill_rowmajor(rm.data()); }
        add_tiled_i8(l, &rm);   // rm -> W^
        if (own_rowmajor) {
            l.bf16 = 0;
            l.w = reserve(count * 4);
            if (host) std::memcpy(host + l.w, rm.data(), count * 4);
            d.n_params += (int64_t)count;
        }
    }

    void add_tiled(Lin& l, const std::function<void(float*)>& fill_rowmajor) {
        if (i8w) {
            std::vector<float> rm;
            if (host) { rm.resize((size_t)l.out * l.in); fill_rowmajor(rm.data()); }
            add_tiled_i8(l, &rm);
            return;
        }
        const int E = bf16w ? 8 : 4, WV = bf16w ? 4 : 8;
        const size_t nt = ((size_t)l.out + 15) / 16, nss = ((size_t)l.in + 127) / 128;
        const size_t count = nt * nss * 16 * 128;
        l.wt = reserve(count * (bf16w ? 2 : 4));
        if (!host) return;
        std::vector<float> rm((size_t)l.out * l.in);
        fill_rowmajor(rm.data());
        uint8_t* dst = host + l.wt;
        for (size_t t = 0; t < nt; t++)
            for (size_t ss = 0; ss < nss; ss++)
                for (int sidx = 0; sidx < WV; sidx++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < E; j++) {
                            size_t n = t * 16 + (size_t)(lane & 15), k = ss * 128 + (size_t)(lane >> 4) * 32 + (size_t)sidx * E + j;
                            float v = (n < (size_t)l.out && k < (size_t)l.in) ? rm[n * l.in + k] : 0.0f;
                            size_t e = ((((t * nss + ss) * WV + sidx) * 64 + lane) * E + j);
                            if (bf16w) reinterpret_cast<uint16_t*>(dst)[e] = f32_to_bf16_rne(v);
                            else reinterpret_cast<float*>(dst)[e] = v;
                        }
    }
    // Fragment-ordered bf16 copy for resblock.hip: [16-column tile][32-deep k step][lane][8] with lane l holding
    // W[tile*16 + (l & 15)][step*32 + (l >> 4)*8 .. +8) -- the row operand of v_mfma_f32_16x16x32_bf16, so a wave's fragment
    // load is one contiguous 1-KiB burst.  f32 weights get a second plane with the bf16 residual (w - hi), the same
    // split the kernels apply to activations.
    void add_frag16(Lin& l, const std::function<void(float*)>& fill_rowmajor) {
        if (l.out % 16 != 0 || l.in % 32 != 0) return;
        const size_t nt = (size_t)l.out / 16, ks = (size_t)l.in / 32, count = nt * ks * 64 * 8;
        l.wf = reserve(count * 2);
        if (!bf16w) l.wf_lo = reserve(count * 2);
        if (!host) return;
        std::vector<float> rm((size_t)l.out * l.in);
        fill_rowmajor(rm.data());
        uint16_t* hi = reinterpret_cast<uint16_t*>(host + l.wf);
        uint16_t* lo = bf16w ? nullptr : reinterpret_cast<uint16_t*>(host + l.wf_lo);
        for (size_t t = 0; t < nt; t++)
            for (size_t s = 0; s < ks; s++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const float v = rm[(t * 16 + (size_t)(lane & 15)) * l.in + s * 32 + (size_t)(lane >> 4) * 8 + j];
                        const size_t e = ((t * ks + s) * 64 + lane) * 8 + j;
                        const uint16_t h = f32_to_bf16_rne(v);
                        hi[e] = h;
                        if (lo) lo[e] = f32_to_bf16_rne(v - bf16_to_f32(h));
                    }
    }
    // linear1 [F][512] and linear2 [512][F] of a Mimi transformer layer as the LDS images of k_mimi_ffn (ffn_fused.hip), bf16: per chunk c of 32 hidden
    // units 64 KB that are copied to LDS as they are --
    //   W1 image (32 KB): [k pair j (8)][hidden unit h (32)][128 B]; the 16-byte piece (k-step parity sg, lane group q) of row h sits at position
    //     (4 sg + q) ^ ((h >> 1) & 7) and holds W1[32c + h][k] for k = 32 (2j + sg) + 4q + (0..3), then 32 (2j + sg) + 16 + 4q + (0..3);
    //   W2 image (32 KB): [output o (512)][64 B]; lane group q's piece at position q ^ (3 ((o >> 3) & 1)) holds W2[o][32c + 4q + (0..3)], then
    //     W2[o][32c + 16 + 4q + (0..3)].
    // The positions make every ds_read_b128 fragment read conflict-free (16 lanes of a service group on 16 different 16-byte bank slots); the k order
    // inside a piece is the order in which a pair of 16x16 accumulator tiles hands its values to the next product.
    // rows [32c, 32c + 32) of a row-major [N][512] matrix as one 32-KB W1-format image (above)
    static void w1_image_chunk(const std::vector<float>& w, int c, uint16_t* i1) {
        const int D = 512;
        for (int j = 0; j < 8; j++)
            for (int h = 0; h < 32; h++)
                for (int sg = 0; sg < 2; sg++)
                    for (int q = 0; q < 4; q++) {
                        const int pos = (4 * sg + q) ^ ((h >> 1) & 7), s = 2 * j + sg;
                        uint16_t* p = i1 + ((size_t)j * 32 + h) * 64 + pos * 8;
                        for (int e = 0; e < 8; e++) {
                            const int k = 32 * s + (e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4));
                            p[e] = f32_to_bf16_rne(w[(size_t)(32 * c + h) * D + k]);
                        }
                    }
    }
    // a [N][512] matrix as N / 32 consecutive W1-format images (k_mimi_rowlin)
    size_t add_w1_image(const std::string& wn, int N) {
        const int nch = N / 32;
        const size_t off = reserve((size_t)nch * 32768);
        if (!host) return off;
        const std::vector<float> w = load(wn);
        for (int c = 0; c < nch; c++) w1_image_chunk(w, c, reinterpret_cast<uint16_t*>(host + off) + (size_t)c * 16384);
        return off;
    }
    size_t add_ffn_image(const std::string& w1n, const std::string& w2n, int F) {
        const int D = 512, nch = F / 32;
        const size_t off = reserve((size_t)nch * 65536);
        if (!host) return off;
        const std::vector<float> w1 = load(w1n), w2 = load(w2n);
        uint16_t* dst = reinterpret_cast<uint16_t*>(host + off);
        for (int c = 0; c < nch; c++) {
            uint16_t* i1 = dst + (size_t)c * 32768;          // (uint16 units: 64 KB per chunk)
            uint16_t* i2 = i1 + 16384;
            w1_image_chunk(w1, c, i1);
            for (int o = 0; o < D; o++)
                for (int q = 0; q < 4; q++) {
                    const int pos = q ^ (3 * ((o >> 3) & 1));
                    uint16_t* p = i2 + (size_t)o * 32 + pos * 8;
                    for (int e = 0; e < 8; e++) {
                        const int hcol = 32 * c + (e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4));
                        p[e] = f32_to_bf16_rne(w2[(size_t)o * F + hcol]);
                    }
                }
        }
        return off;
    }
    // the weights of a step linear as the kernels multiply them: the file's values, rounded to bf16 or int8-quantized per the mode
    std::vector<float> effective_weights(const std::string& wn) {
        std::vector<float> rm = load(wn);
        const size_t out = (size_t)shape(wn)[0], in = (size_t)shape(wn)[1];
        if (i8w) {
            std::vector<float> sc;
            std::vector<int8_t> q;
            quantize_rows(rm, out, in, sc, q);
        } else if (bf16w) {
            for (float& v : rm) v = bf16_to_f32(f32_to_bf16_rne(v));
        }
        return rm;
    }
    Lin step_linear(const std::string& name, bool with_bias) {
        if (!i8w) {
            Lin l = linear(name, with_bias);
            const std::string wn = name + ".weight";
            add_tiled(l, [&](float* dst) { f.decode_f32(wn, dst); });
            return l;
        }
        Lin l;
        const std::string wn = name + ".weight";
        expect_rank(wn, 2);
        l.out = (int)shape(wn)[0];
        l.in = (int)shape(wn)[1];
        add_step_matrix(l, [&](float* dst) { f.decode_f32(wn, dst); }, true);
        if (with_bias && has(name + ".bias")) {
            const std::string bn = name + ".bias";
            if (shape(bn).size() != 1 || shape(bn)[0] != l.out) throw Error(PTTS_EFORMAT, strfmt("native: linear \"%s\" bias shape incompatible with weight", name.c_str()));
            l.b = add_f32((size_t)l.out, [&](float* dst) { f.decode_f32(bn, dst); });
        }
        return l;
    }

    Lin linear(const std::string& name, bool with_bias) {  // linear.go:18-45
        Lin l;
        const std::string wn = name + ".weight";
        expect_rank(wn, 2);
        l.out = (int)shape(wn)[0];
        l.in = (int)shape(wn)[1];
        l.w = add_mat((size_t)l.out * l.in, [&](float* dst) { f.decode_f32(wn, dst); }, &l.bf16);
        if (with_bias && has(name + ".bias")) {
            const std::string bn = name + ".bias";
            if (shape(bn).size() != 1 || shape(bn)[0] != l.out) throw Error(PTTS_EFORMAT, strfmt("native: linear \"%s\" bias shape incompatible with weight", name.c_str()));
            l.b = add_f32((size_t)l.out, [&](float* dst) { f.decode_f32(bn, dst); });
        }
        return l;
    }
    Norm norm(const std::string& name, float eps) {  // linear.go:191-207
        Norm n;
        const std::string wn = name + ".weight", bn = name + ".bias";
        expect_rank(wn, 1);
        expect_rank(bn, 1);
        if (shape(wn)[0] != shape(bn)[0]) throw Error(PTTS_EFORMAT, strfmt("native: layernorm \"%s\" invalid shapes", name.c_str()));
        n.d = (int)shape(wn)[0];
        n.eps = eps;
        n.w = add_f32((size_t)n.d, [&](float* dst) { f.decode_f32(wn, dst); });
        n.b = add_f32((size_t)n.d, [&](float* dst) { f.decode_f32(bn, dst); });
        d.n_params += 2 * n.d;
        return n;
    }
    // Conv1d [Cout, Cin, k] -> GEMM operand [Cout][kx*Cin + ic] for channels-last windows (conv1d.go:86-88)
    Lin conv_as_gemm(const std::string& name, int* k_out, int* cin_out, bool frag16 = false) {
        Lin l;
        const std::string wn = name + ".weight";
        expect_rank(wn, 3);
        int oc = (int)shape(wn)[0], ic = (int)shape(wn)[1], k = (int)shape(wn)[2];
        *k_out = k;
        *cin_out = ic;
        l.out = oc;
        l.in = ic * k;
        auto fill = [&](float* dst) {
            std::vector<float> w = load(wn);
            for (int o = 0; o < oc; o++)
                for (int c = 0; c < ic; c++)
                    for (int x = 0; x < k; x++) dst[(size_t)o * ic * k + (size_t)x * ic + c] = w[((size_t)o * ic + c) * k + x];
        };
        l.w = add_mat((size_t)oc * ic * k, fill, &l.bf16);
        if (frag16) add_frag16(l, fill);
        if (has(name + ".bias")) l.b = add_f32((size_t)oc, [&](float* dst) { f.decode_f32(name + ".bias", dst); });
        return l;
    }
    // ConvTranspose1d [Cin, Cout, k=2s], first L*s outputs kept (mimi.go:116-125): out[t*s + r, oc] =
    //   sum_ic x[t-1, ic] * W[ic, oc, r + s] + x[t, ic] * W[ic, oc, r]   -> GEMM operand [(r, oc)][(j, ic)], j = 0: x[t-1], 1: x[t]
    Lin convtr_as_gemm(const std::string& name, int stride, int* cin_out, int* cout_out, bool frag16 = false) {
        Lin l;
        const std::string wn = name + ".weight";
        expect_rank(wn, 3);
        int ic = (int)shape(wn)[0], oc = (int)shape(wn)[1], k = (int)shape(wn)[2];
        if (k != 2 * stride) throw Error(PTTS_EFORMAT, strfmt("native: convtranspose \"%s\" kernel %d, this build needs kernel = 2*stride = %d", name.c_str(), k, 2 * stride));
        *cin_out = ic;
        *cout_out = oc;
        l.out = stride * oc;
        l.in = 2 * ic;
        auto fill = [&](float* dst) {
            std::vector<float> w = load(wn);
            for (int r = 0; r < stride; r++)
                for (int o = 0; o < oc; o++)
                    for (int c = 0; c < ic; c++) {
                        size_t row = (size_t)r * oc + o;
                        dst[row * 2 * ic + c] = w[((size_t)c * oc + o) * k + r + stride];
                        dst[row * 2 * ic + ic + c] = w[((size_t)c * oc + o) * k + r];
                    }
        };
        l.w = add_mat((size_t)l.out * l.in, fill, &l.bf16);
        if (frag16 && bf16w && stride == 4 && oc == 64) {
            // the last transposed convolution inside the fused block (resblock_up.hip): fragment-ordered, with the output columns regrouped so
            // that column q of the 16-column tile T holds phase q >> 2 of channel 8 (T >> 1) + 4 (T & 1) + (q & 3) -- a lane of the product then
            // owns four channels of ONE phase and the 64 lanes of a wave 64 consecutive output rows (conflict-free LDS writes)
            add_frag16(l, [&](float* dst) {
                std::vector<float> rm((size_t)l.out * l.in);
                fill(rm.data());
                for (int T = 0; T < l.out / 16; T++)
                    for (int q = 0; q < 16; q++) {
                        const int n = (q >> 2) * oc + 8 * (T >> 1) + 4 * (T & 1) + (q & 3);
                        std::copy(rm.begin() + (size_t)n * l.in, rm.begin() + (size_t)(n + 1) * l.in, dst + (size_t)(T * 16 + q) * l.in);
                    }
            });
        }
        if (has(name + ".bias"))
            l.b = add_f32((size_t)l.out, [&](float* dst) {
                std::vector<float> b = load(name + ".bias");
                for (int r = 0; r < stride; r++)
                    for (int o = 0; o < oc; o++) dst[(size_t)r * oc + o] = b[o];
            });
        return l;
    }

    void rope_tables(int hd, double max_period, size_t* cos_off, size_t* sin_off) {  // flow_transformer.go:797-832
        int half = hd / 2;
        auto gen = [&](bool want_cos) {
            return [=](float* dst) {
                std::vector<double> inv((size_t)half);
                for (int i = 0; i < half; i++) inv[i] = 1.0 / std::pow(max_period, (double)i / (double)half);
                for (int pos = 0; pos < ROPE_SEQ; pos++)
                    for (int i = 0; i < half; i++) {
                        double ang = (double)pos * inv[i];
                        dst[(size_t)pos * half + i] = (float)(want_cos ? std::cos(ang) : std::sin(ang));
                    }
            };
        };
        *cos_off = add_f32((size_t)ROPE_SEQ * half, gen(true));
        *sin_off = add_f32((size_t)ROPE_SEQ * half, gen(false));
    }

    void run() {
        d.n_params = 0;
        // ---------------- flow_lm ----------------
        const std::string fl = "flow_lm.";
        const std::string en = fl + "conditioner.embed.weight";
        expect_rank(en, 2);
        d.n_bins = (int)shape(en)[0];
        d.d_model = (int)shape(en)[1];
        d.embed = add_f32((size_t)d.n_bins * d.d_model, [&](float* dst) { f.decode_f32(en, dst); });
        d.n_params += (int64_t)d.n_bins * d.d_model;
        d.n_layers = 0;
        for (int i = 0; i < MAX_LAYERS; i++) {
            std::string p = fl + "transformer.layers." + std::to_string(i);
            if (!has(p + ".norm1.weight")) break;
            auto& L = d.layers[i];
            L.n1 = norm(p + ".norm1", 1e-5f);
            L.n2 = norm(p + ".norm2", 1e-5f);
            L.in_proj = step_linear(p + ".self_attn.in_proj", false);
            L.out_proj = step_linear(p + ".self_attn.out_proj", false);
            L.l1 = step_linear(p + ".linear1", false);
            L.l2 = step_linear(p + ".linear2", false);
            if (L.out_proj.out % d.heads) throw Error(PTTS_EFORMAT, strfmt("native: d_model %d not divisible by num_heads %d", L.out_proj.out, d.heads));
            d.n_layers++;
        }
        if (d.n_layers == 0) throw Error(PTTS_EFORMAT, "native: no flow_lm transformer layers found");
        d.hd = d.layers[0].out_proj.out / d.heads;
        d.ffn = d.layers[0].l1.out;
        if (d.hd != 64) throw Error(PTTS_EINVAL, strfmt("ptts-hip: flow head_dim %d unsupported (kernels are built for 64)", d.hd));
        rope_tables(d.hd, 10000.0, &d.rope_cos, &d.rope_sin);
        for (const char* nm : {"emb_std", "emb_mean", "bos_emb"}) {
            expect_rank(fl + nm, 1);
            if (shape(fl + nm)[0] != d.ldim) throw Error(PTTS_EFORMAT, strfmt("native varbuilder: tensor \"flow_lm.%s\" shape does not match expected [%d]", nm, d.ldim));
        }
        d.bos = add_f32((size_t)d.ldim, [&](float* dst) { f.decode_f32(fl + "bos_emb", dst); });
        d.input_linear = step_linear(fl + "input_linear", true);
        d.out_norm = norm(fl + "out_norm", 1e-5f);
        d.out_eos = step_linear(fl + "out_eos", true);
        // ---------------- flow_net ----------------
        const std::string fn = fl + "flow_net.";
        for (int i = 0; i < 2; i++) {
            std::string p = fn + "time_embed." + std::to_string(i);
            auto& te = d.te[i];
            d.nfreq = (int)f.at(p + ".freqs").count();
            te.freqs = add_f32((size_t)d.nfreq, [&](float* dst) { f.decode_f32(p + ".freqs", dst); });
            te.l1 = linear(p + ".mlp.0", true);
            te.l2 = linear(p + ".mlp.2", true);
            te.alpha = add_f32((size_t)f.at(p + ".mlp.3.alpha").count(), [&](float* dst) { f.decode_f32(p + ".mlp.3.alpha", dst); });
        }
        d.cond_embed = step_linear(fn + "cond_embed", true);
        if (d.out_eos.in == d.cond_embed.in && d.out_eos.out == 1) {   // [cond_embed ; out_eos]: N = flow_dim + 1
            const std::string n1 = fn + "cond_embed", n2 = fl + "out_eos";
            const size_t c1 = (size_t)d.cond_embed.out * d.cond_embed.in, c2 = (size_t)d.out_eos.in;
            d.cond_eos.out = d.cond_embed.out + 1;
            d.cond_eos.in = d.cond_embed.in;
            d.cond_eos.bf16 = d.cond_embed.bf16;
            d.cond_eos.w = d.cond_embed.w;   // row-major copy of the first N-1 rows only: the stacked operand exists in tiled form alone
            add_tiled(d.cond_eos, [&](float* dst) { f.decode_f32(n1 + ".weight", dst); f.decode_f32(n2 + ".weight", dst + c1); (void)c2; });
            d.cond_eos.b = add_f32((size_t)d.cond_eos.out, [&](float* dst) {
                std::fill(dst, dst + d.cond_eos.out, 0.0f);
                if (has(n1 + ".bias")) f.decode_f32(n1 + ".bias", dst);
                if (has(n2 + ".bias")) f.decode_f32(n2 + ".bias", dst + d.cond_embed.out);
            });
        }
        d.input_proj = step_linear(fn + "input_proj", true);
        d.flow_dim = d.input_proj.out;
        d.flow_depth = 0;
        for (int i = 0; i < MAX_LAYERS; i++) {
            std::string p = fn + "res_blocks." + std::to_string(i);
            if (!has(p + ".in_ln.weight")) break;
            auto& rb = d.rb[i];
            rb.ln = norm(p + ".in_ln", 1e-6f);
            rb.mlp0 = step_linear(p + ".mlp.0", true);
            rb.mlp2 = step_linear(p + ".mlp.2", true);
            d.flow_depth++;
        }
        if (d.flow_depth == 0) throw Error(PTTS_EFORMAT, "native: no flow_net res blocks found");
        {
            // every adaLN matrix consumes silu(y) only (flow_net.go:117,206), so the depth*3C + 2C output rows are
            // stacked into one [N, C] operand: one weight stream instead of depth+1 small dependent launches
            const int C = d.flow_dim;
            std::vector<std::string> names;
            for (int i = 0; i < d.flow_depth; i++) names.push_back(fn + "res_blocks." + std::to_string(i) + ".adaLN_modulation.1");
            names.push_back(fn + "final_layer.adaLN_modulation.1");
            int rows = 0;
            for (auto& n : names) {
                expect_rank(n + ".weight", 2);
                if (shape(n + ".weight")[1] != C) throw Error(PTTS_EFORMAT, strfmt("native: \"%s\" input width mismatch", n.c_str()));
                rows += (int)shape(n + ".weight")[0];
            }
            d.ada_all.out = rows;
            d.ada_all.in = C;
            add_step_matrix(d.ada_all, [&](float* dst) {
                size_t o = 0;
                for (auto& n : names) { f.decode_f32(n + ".weight", dst + o); o += (size_t)f.at(n + ".weight").count(); }
            }, true);
            d.ada_all.b = add_f32((size_t)rows, [&](float* dst) {
                size_t o = 0;
                for (auto& n : names) {
                    size_t cnt = (size_t)shape(n + ".weight")[0];
                    if (has(n + ".bias")) f.decode_f32(n + ".bias", dst + o);
                    else std::fill(dst + o, dst + o + cnt, 0.0f);
                    o += cnt;
                }
            });
        }
        d.final_linear = step_linear(fn + "final_layer.linear", true);
        // speaker conditioning projection (onnx/voice_encode.go:160-202: either name; [VoiceEmbeddingDim, mimiEncoderLatentDim]), optional
        d.speaker_proj = Lin{};   // (the fill pass starts from the planned Desc)
        for (const char* nm : {"flow_lm.speaker_proj_weight", "condition_provider.conditioners.speaker_wavs.output_proj.weight"}) {
            if (!has(nm) || d.speaker_proj.w != NONE) continue;
            expect_rank(nm, 2);
            const std::string name = nm;
            d.speaker_proj.out = (int)shape(name)[0];
            d.speaker_proj.in = (int)shape(name)[1];
            // held in f32: the reference multiplies the f32 tensor as it is (projectSpeakerConditioning)
            d.speaker_proj.w = add_f32((size_t)d.speaker_proj.out * d.speaker_proj.in, [this, name](float* dst) { f.decode_f32(name, dst); });
            d.speaker_proj.bf16 = 0;
            d.n_params += (int64_t)d.speaker_proj.out * d.speaker_proj.in;
        }
        // ---------------- mimi ----------------
        const std::string mi = "mimi.";
        {
            const std::string qn = mi + "quantizer.output_proj.weight";
            expect_rank(qn, 3);
            int oc = (int)shape(qn)[0], ic = (int)shape(qn)[1], k = (int)shape(qn)[2];
            if (k != 1 || ic != d.ldim) throw Error(PTTS_EFORMAT, "native: quantizer projection must be a 1x1 conv over the latent dim");
            d.mimi_dim = oc;
            // model.go:226-242: W'[oc,ic] = W[oc,ic]*std[ic]; b'[oc] = b[oc] + sum_ic W[oc,ic]*mean[ic] (sequential f32)
            auto fold = [&](std::vector<float>& wf, std::vector<float>& bf) {
                std::vector<float> w = load(qn), sd = load(fl + "emb_std"), mn = load(fl + "emb_mean");
                std::vector<float> braw;
                if (has(mi + "quantizer.output_proj.bias")) braw = load(mi + "quantizer.output_proj.bias");
                wf.resize((size_t)oc * ic);
                bf.resize((size_t)oc);
                for (int o = 0; o < oc; o++) {
                    float bv = braw.empty() ? 0.0f : braw[o];
                    for (int c = 0; c < ic; c++) {
                        float wv = w[(size_t)o * ic + c];
                        wf[(size_t)o * ic + c] = wv * sd[c];
                        float t = wv * mn[c];
                        bv = bv + t;
                    }
                    bf[o] = bv;
                }
            };
            d.proj_w = add_f32((size_t)oc * ic, [&](float* dst) { std::vector<float> wf, bf; fold(wf, bf); std::copy(wf.begin(), wf.end(), dst); });
            d.proj_b = add_f32((size_t)oc, [&](float* dst) { std::vector<float> wf, bf; fold(wf, bf); std::copy(bf.begin(), bf.end(), dst); });
            d.n_params += (int64_t)oc * ic;
        }
        {
            const std::string un = mi + "upsample.convtr.convtr.weight";  // [C, 1, k], groups = C (mimi.go:567)
            expect_rank(un, 3);
            int c = (int)shape(un)[0], opg = (int)shape(un)[1], k = (int)shape(un)[2];
            if (c != d.mimi_dim || opg != 1 || k != 2 * d.up_stride)
                throw Error(PTTS_EFORMAT, strfmt("native: upsample convtr shape [%d,%d,%d], want depthwise [%d,1,%d]", c, opg, k, d.mimi_dim, 2 * d.up_stride));
            d.up_k = k;
            const int s = d.up_stride;
            d.up_w0 = add_f32((size_t)s * c, [&](float* dst) { std::vector<float> w = load(un); for (int r = 0; r < s; r++) for (int ch = 0; ch < c; ch++) dst[(size_t)r * c + ch] = w[(size_t)ch * k + r + s]; });
            d.up_w1 = add_f32((size_t)s * c, [&](float* dst) { std::vector<float> w = load(un); for (int r = 0; r < s; r++) for (int ch = 0; ch < c; ch++) dst[(size_t)r * c + ch] = w[(size_t)ch * k + r]; });
            d.n_params += (int64_t)c * k;
        }
        d.mimi_layers = 0;
        for (int i = 0; i < MAX_LAYERS; i++) {
            std::string p = mi + "decoder_transformer.transformer.layers." + std::to_string(i);
            if (!has(p + ".norm1.weight")) break;
            auto& L = d.ml[i];
            L.n1 = norm(p + ".norm1", 1e-5f);
            L.n2 = norm(p + ".norm2", 1e-5f);
            L.in_proj = linear(p + ".self_attn.in_proj", false);
            L.out_proj = linear(p + ".self_attn.out_proj", false);
            L.l1 = linear(p + ".linear1", false);
            L.l2 = linear(p + ".linear2", false);
            if (has(p + ".layer_scale_1.scale")) L.ls1 = add_f32((size_t)f.at(p + ".layer_scale_1.scale").count(), [&](float* dst) { f.decode_f32(p + ".layer_scale_1.scale", dst); });
            if (has(p + ".layer_scale_2.scale")) L.ls2 = add_f32((size_t)f.at(p + ".layer_scale_2.scale").count(), [&](float* dst) { f.decode_f32(p + ".layer_scale_2.scale", dst); });
            if (L.out_proj.out % d.mimi_heads) throw Error(PTTS_EFORMAT, strfmt("native: mimi d_model %d not divisible by heads %d", L.out_proj.out, d.mimi_heads));
            if (bf16w && L.l1.in == 512 && L.l2.out == 512 && L.l1.out == L.l2.in && L.l1.out % 32 == 0 && L.l1.b == NONE && L.l2.b == NONE)
                L.ffn_img = add_ffn_image(p + ".linear1.weight", p + ".linear2.weight", L.l1.out);
            if (bf16w && L.in_proj.in == 512 && L.in_proj.out % 64 == 0 && L.in_proj.b == NONE)
                L.qkv_img = add_w1_image(p + ".self_attn.in_proj.weight", L.in_proj.out);
            d.mimi_layers++;
        }
        if (d.mimi_layers == 0) throw Error(PTTS_EFORMAT, "native: no mimi decoder transformer layers found");
        d.mimi_hd = d.ml[0].out_proj.out / d.mimi_heads;
        d.mimi_ffn = d.ml[0].l1.out;
        if (d.mimi_hd != 64 || d.ml[0].out_proj.out != d.mimi_dim) throw Error(PTTS_EINVAL, strfmt("ptts-hip: mimi head_dim %d unsupported (kernels are built for 64)", d.mimi_hd));
        int cin = 0, cout = 0;
        d.init_conv = conv_as_gemm(mi + "decoder.model.0.conv", &d.init_k, &cin);
        if (cin != d.mimi_dim) throw Error(PTTS_EFORMAT, "native: decoder initConv input channels mismatch");
        d.sea_ch[0] = d.init_conv.out;
        static const int up_idx[3] = {2, 5, 8}, rb_idx[3] = {3, 6, 9};
        for (int j = 0; j < 3; j++) {
            d.up[j] = convtr_as_gemm(mi + "decoder.model." + std::to_string(up_idx[j]) + ".convtr", d.strides[j], &cin, &cout, j == 2);
            if (cin != d.sea_ch[j]) throw Error(PTTS_EFORMAT, "native: decoder convtr input channels mismatch");
            d.sea_ch[j + 1] = cout;
            int c1 = 0, c2 = 0;
            d.rb1[j] = conv_as_gemm(mi + "decoder.model." + std::to_string(rb_idx[j]) + ".block.1.conv", &d.rb_k1[j], &c1, true);
            d.rb2[j] = conv_as_gemm(mi + "decoder.model." + std::to_string(rb_idx[j]) + ".block.3.conv", &d.rb_k2[j], &c2, true);
            d.sea_hidden[j] = d.rb1[j].out;
            if (c1 != cout || c2 != d.sea_hidden[j] || d.rb2[j].out != cout) throw Error(PTTS_EFORMAT, "native: SEANet residual block channel mismatch");
        }
        {
            const std::string cn = mi + "decoder.model.11.conv";
            expect_rank(cn + ".weight", 3);
            int oc = (int)shape(cn + ".weight")[0], ic = (int)shape(cn + ".weight")[1], k = (int)shape(cn + ".weight")[2];
            if (oc != 1 || ic != d.sea_ch[3]) throw Error(PTTS_EFORMAT, "native: final conv must map the last SEANet width to 1 channel");
            d.final_k = k;
            d.final_w = add_f32((size_t)ic * k, [&](float* dst) { std::vector<float> w = load(cn + ".weight"); for (int c = 0; c < ic; c++) for (int x = 0; x < k; x++) dst[(size_t)x * ic + c] = w[(size_t)c * k + x]; });
            if (has(cn + ".bias")) d.final_b = add_f32(1, [&](float* dst) { f.decode_f32(cn + ".bias", dst); });
            d.n_params += (int64_t)ic * k;
            if ((ic * k) % 32 == 0) {   // one-column matrix in MFMA fragment order for the fused last block
                const size_t ks = (size_t)ic * k / 32, count = ks * 64 * 8;
                d.final_wf = reserve(count * 2);
                d.final_wf_lo = reserve(count * 2);
                if (host) {
                    std::vector<float> w = load(cn + ".weight");
                    uint16_t* hi = reinterpret_cast<uint16_t*>(host + d.final_wf);
                    uint16_t* lo = reinterpret_cast<uint16_t*>(host + d.final_wf_lo);
                    for (size_t s = 0; s < ks; s++)
                        for (int lane = 0; lane < 64; lane++)
                            for (int j = 0; j < 8; j++) {
                                const size_t kk = s * 32 + (size_t)(lane >> 4) * 8 + j;   // = tap * ic + c
                                const float v = (lane & 15) == 0 ? w[(kk % ic) * k + kk / ic] : 0.0f;
                                const uint16_t h = f32_to_bf16_rne(v);
                                hi[(s * 64 + lane) * 8 + j] = h;
                                lo[(s * 64 + lane) * 8 + j] = f32_to_bf16_rne(v - bf16_to_f32(h));
                            }
                }
            }
        }
        d.samples_per_frame = (int64_t)d.up_stride * d.strides[0] * d.strides[1] * d.strides[2];
        d.total_bytes = (cur + 255) & ~(size_t)255;
    }
};

}  // namespace

void plan_build(Plan& p) {
    Walker w{p.file, p.desc, p.opts.weights != PTTS_WEIGHTS_F32, nullptr, p.opts.weights == PTTS_WEIGHTS_INT8};
    w.run();
}

void plan_fill(const Plan& p, uint8_t* host) {
    Desc scratch = p.desc;
    Walker w{p.file, scratch, p.opts.weights != PTTS_WEIGHTS_F32, host, p.opts.weights == PTTS_WEIGHTS_INT8};
    w.run();
    if (scratch.total_bytes != p.desc.total_bytes) throw Error(PTTS_EFORMAT, "ptts-hip: arena layout changed between plan and fill");
}

}  // namespace ptts
