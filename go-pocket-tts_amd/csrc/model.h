// model.h -- device-resident model: one contiguous HBM arena + a table of offsets.
//
// The arena holds every tensor in the layout the kernels stream it in (weights [out, in]
// row-major in f32 or bf16, convolution kernels repacked as GEMM operands for channels-last
// activations, emb_std/emb_mean folded into the latent->mimi projection, RoPE tables, the
// concatenated adaLN matrix).  Because it is one block whose layout depends on the file
// header alone, it can be filled by one rank and broadcast once over RCCL.
#pragma once

#include "common.h"

namespace ptts {

constexpr size_t NONE = (size_t)-1;
constexpr int MAX_LAYERS = 32;
constexpr int ROPE_SEQ = 8192;  // flow_transformer.go:505, mimi.go:498

struct Lin {   // linear.go:11-16 (also a convolution expressed as a GEMM)
    size_t w = NONE, b = NONE;
    size_t wt = NONE;   // second copy in the AR-step kernel's fragment order (skinny.hip), NONE for weights the step never streams
    size_t wscale = NONE; int wt_i8 = 0;   // PTTS_WEIGHTS_INT8: wt holds per-row-scaled int8 (q + 128), wscale the f32 scale of every row
    size_t wf = NONE, wf_lo = NONE;   // copy in 16x16x32 MFMA fragment order for the fused SEANet block (resblock.hip); lo plane: f32 weights only
    int in = 0, out = 0, bf16 = 0;
};
struct Norm {  // linear.go:184-189
    size_t w = NONE, b = NONE;
    int d = 0;
    float eps = 1e-5f;
};

struct Desc {
    // flow_lm (flow_lm.go:13-43)
    int d_model = 0, heads = 16, hd = 0, n_layers = 0, ffn = 0, ldim = 32, n_bins = 0;
    size_t embed = NONE, bos = NONE, rope_cos = NONE, rope_sin = NONE;
    Lin input_linear, out_eos;
    Norm out_norm;
    struct Layer {
        Norm n1, n2; Lin in_proj, out_proj, l1, l2;
        // norm2 folded into linear1's epilogue (kernels.h GemmArgs::stats_in): wg[n] = sum_k g[k] W[n][k], wb[n] = sum_k b[k] W[n][k]
        // over the weights as the step computes with them (f64 sums, stored f32)
    } layers[MAX_LAYERS];
    // flow_net (flow_net.go:242-248)
    int flow_dim = 0, flow_depth = 0, nfreq = 0;
    struct TE { size_t freqs = NONE, alpha = NONE; Lin l1, l2; } te[2];
    Lin cond_embed, input_proj, ada_all, final_linear;
    Lin speaker_proj;   // optional: flow_lm.speaker_proj_weight [d_model, 512] (voice cloning: Mimi-encoder latents -> voice embedding)
    Lin cond_eos;   // cond_embed with out_eos stacked as its last row: both read the out_norm rows, one launch (runtime.cpp step_core)
    struct RB { Norm ln; Lin mlp0, mlp2; } rb[MAX_LAYERS];
    // mimi (mimi.go:16-34,528-544)
    int mimi_dim = 0, mimi_heads = 8, mimi_hd = 0, mimi_layers = 0, mimi_ffn = 0, mimi_ctx = 250;
    int up_stride = 16, up_k = 32;
    size_t proj_w = NONE, proj_b = NONE, up_w0 = NONE, up_w1 = NONE;
    struct ML {
        Norm n1, n2; Lin in_proj, out_proj, l1, l2; size_t ls1 = NONE, ls2 = NONE;
        size_t ffn_img = NONE;   // linear1 / linear2 as the per-chunk LDS images of the fused feed-forward kernel (ffn_fused.hip; bf16 weights, width 512)
        size_t qkv_img = NONE;   // in_proj as W1-format chunk images for k_mimi_rowlin (norm1 + in_proj + RoPE in one launch)
    } ml[MAX_LAYERS];
    int sea_ch[4] = {0, 0, 0, 0};        // channels after initConv, up1, up2, up3
    int sea_hidden[3] = {0, 0, 0};
    int strides[3] = {6, 5, 4};          // mimi.go:582,592,602
    int init_k = 0, rb_k1[3] = {0, 0, 0}, rb_k2[3] = {0, 0, 0}, final_k = 0;
    Lin init_conv, up[3], rb1[3], rb2[3];
    size_t final_w = NONE, final_b = NONE;
    size_t final_wf = NONE, final_wf_lo = NONE;   // final conv as column 0 of a 16-column fragment-ordered matrix, bf16 hi + lo planes (resblock.hip)
    int64_t samples_per_frame = 0;
    int64_t n_params = 0;
    size_t total_bytes = 0;
};

struct Plan {
    StFile file;
    ptts_opts opts;
    Desc desc;
};

void plan_build(Plan& p);                       // header -> Desc + arena layout
void plan_fill(const Plan& p, uint8_t* host);   // decode / convert / derive into a host image of the arena

}  // namespace ptts
