// norm.hip -- LayerNorm (K3; linear.go:265-329 / nn_ops.go:79-149) with the adaLN modulation of the flow net (K11)
// and the split-K reduction + residual update of the AR step fused in front.
#include "kernels.h"
#include "device_util.h"

namespace ptts {

// generic rows (any width / alignment): one wave per row, mean and biased variance accumulated in f64 as the reference does
__global__ __launch_bounds__(256) void k_layernorm(LnArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const float* x = a.x + row_off(a.xmap, row);
    double s = 0.0;
    for (int i = lane; i < a.d; i += WAVE) s += (double)x[i];
    const double mean = wave_sum(s) / (double)a.d;
    double v = 0.0;
    for (int i = lane; i < a.d; i += WAVE) { double dlt = (double)x[i] - mean; v += dlt * dlt; }
    const double var = wave_sum(v) / (double)a.d;
    const float inv_std = (float)(1.0 / sqrt(var + (double)a.eps));
    const float meanf = (float)mean;
    float* y = a.y + (int64_t)row * a.ldy;
    const float* sh = a.shift ? a.shift + (int64_t)row * a.ldmod : nullptr;
    const float* sc = a.scale ? a.scale + (int64_t)row * a.ldmod : nullptr;
    for (int i = lane; i < a.d; i += WAVE) {
        float n = (x[i] - meanf) * inv_std;
        if (a.w) n = n * a.w[i];
        if (a.b) n = n + a.b[i];
        if (sc) n = n * (sc[i] + 1.0f) + sh[i];
        y[i] = n;
    }
}

// d % 4 == 0 and d <= 1024: the row lives in registers (one 16-byte load per 4 values).  Optional prologue
//   x[row] += gate * scale * (sum_z partial[z][row] + pbias)
// folds the split-K reduction and the residual update of the preceding linear into this launch; the partials
// are added in a fixed order, so results are bitwise reproducible.
__global__ __launch_bounds__(256) void k_layernorm_reg(LnArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    float* xw = const_cast<float*>(a.x) + row_off(a.xmap, row);
    const int nv = a.d >> 2;
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int f = lane + j * 64;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < nv) {
            v[j] = reinterpret_cast<const float4*>(xw)[f];
            if (a.partial) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int z0 = 0; z0 < a.splitk; z0 += 8) {   // 8 independent loads in flight, then a fixed-order sum
                    float4 p[8];
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        p[u] = z0 + u < a.splitk ? reinterpret_cast<const float4*>(a.partial + (int64_t)(z0 + u) * a.pstride + (int64_t)row * a.d)[f]
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < 8; u++) { acc.x += p[u].x; acc.y += p[u].y; acc.z += p[u].z; acc.w += p[u].w; }
                }
                if (a.pbias) { float4 b = reinterpret_cast<const float4*>(a.pbias)[f]; acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w; }
                if (a.pgate) { float4 g = reinterpret_cast<const float4*>(a.pgate + (int64_t)row * a.ldpg)[f]; acc.x *= g.x; acc.y *= g.y; acc.z *= g.z; acc.w *= g.w; }
                if (a.pscale) { float4 g = reinterpret_cast<const float4*>(a.pscale)[f]; acc.x *= g.x; acc.y *= g.y; acc.z *= g.z; acc.w *= g.w; }
                v[j].x += acc.x; v[j].y += acc.y; v[j].z += acc.z; v[j].w += acc.w;
                reinterpret_cast<float4*>(xw)[f] = v[j];
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (lane + j * 64 < nv) s += ((double)v[j].x + (double)v[j].y) + ((double)v[j].z + (double)v[j].w);
    const double mean = wave_sum(s) / (double)a.d;
    double q = 0.0;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (lane + j * 64 < nv) {
            double d0 = (double)v[j].x - mean, d1 = (double)v[j].y - mean, d2 = (double)v[j].z - mean, d3 = (double)v[j].w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    const double var = wave_sum(q) / (double)a.d;
    const float inv_std = (float)(1.0 / sqrt(var + (double)a.eps));
    const float meanf = (float)mean;
    if (!a.y) return;   // reduction only
    float* y = a.y + (int64_t)row * a.ldy;
    const float* sh = a.shift ? a.shift + (int64_t)row * a.ldmod : nullptr;
    const float* sc = a.scale ? a.scale + (int64_t)row * a.ldmod : nullptr;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int f = lane + j * 64;
        if (f >= nv) continue;
        float4 o;
        o.x = (v[j].x - meanf) * inv_std; o.y = (v[j].y - meanf) * inv_std; o.z = (v[j].z - meanf) * inv_std; o.w = (v[j].w - meanf) * inv_std;
        if (a.w) { float4 w = reinterpret_cast<const float4*>(a.w)[f]; o.x *= w.x; o.y *= w.y; o.z *= w.z; o.w *= w.w; }
        if (a.b) { float4 b = reinterpret_cast<const float4*>(a.b)[f]; o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w; }
        if (sc) {
            float4 c = reinterpret_cast<const float4*>(sc)[f], h = reinterpret_cast<const float4*>(sh)[f];
            o.x = o.x * (c.x + 1.0f) + h.x; o.y = o.y * (c.y + 1.0f) + h.y; o.z = o.z * (c.z + 1.0f) + h.z; o.w = o.w * (c.w + 1.0f) + h.w;
        }
        reinterpret_cast<float4*>(y)[f] = o;
    }
}

typedef __bf16 nbf16x2 __attribute__((ext_vector_type(2)));
typedef float nf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nsplit2(float a, float b, unsigned& hi, unsigned& lo) {   // as skinny.hip split2
    nf32x2 f = {a, b};
    nbf16x2 h = __builtin_convertvector(f, nbf16x2);
    nf32x2 r = f - __builtin_convertvector(h, nf32x2);
    nbf16x2 l = __builtin_convertvector(r, nbf16x2);
    hi = *reinterpret_cast<unsigned*>(&h);
    lo = *reinterpret_cast<unsigned*>(&l);
}

// one block per row, thread t owns the float4 columns t + 256 j
__global__ __launch_bounds__(256) void k_combine_ln(CombineLnArgs a) {
    __shared__ float red[8];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nv = a.d >> 2;
    const float* xr = a.x + (int64_t)row * a.d;
    float4 v[4];
    float4 lw[4], lb[4];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int f = tid + 256 * j;
        ok[j] = f < nv;
        const int fc = ok[j] ? f : 0;
        v[j] = reinterpret_cast<const float4*>(xr)[fc];
        lw[j] = a.ln_w ? reinterpret_cast<const float4*>(a.ln_w)[fc] : make_float4(1.f, 1.f, 1.f, 1.f);
        lb[j] = a.ln_b ? reinterpret_cast<const float4*>(a.ln_b)[fc] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.partial) {   // the linear's output is summed first (planes in order, then the bias), then added to x: y = W h + b; x += y
            float4 acc = reinterpret_cast<const float4*>(a.partial + (int64_t)row * a.d)[fc];
            for (int z = 1; z < a.splitk; z++) {
                const float4 p = reinterpret_cast<const float4*>(a.partial + (int64_t)z * a.pstride + (int64_t)row * a.d)[fc];
                acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
            }
            if (a.pbias) { const float4 b = reinterpret_cast<const float4*>(a.pbias)[fc]; acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w; }
            v[j].x += acc.x; v[j].y += acc.y; v[j].z += acc.z; v[j].w += acc.w;
            if (ok[j] && a.x_out) reinterpret_cast<float4*>(a.x_out + (int64_t)row * a.d)[f] = v[j];
        }
        if (!ok[j]) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    s = wave_sum_dpp(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float rk = 1.0f / (float)a.d;
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) * rk;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (ok[j]) { const float d0 = v[j].x - mean, d1 = v[j].y - mean, d2 = v[j].z - mean, d3 = v[j].w - mean; q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3); }
    q = wave_sum_dpp(q);
    if (lane == 0) red[4 + wave] = q;
    __syncthreads();
    const float inv_std = 1.0f / sqrtf(((red[4] + red[5]) + (red[6] + red[7])) * rk + a.eps);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int f = tid + 256 * j;
        if (!ok[j]) continue;
        float4 o;
        o.x = (v[j].x - mean) * inv_std * lw[j].x + lb[j].x; o.y = (v[j].y - mean) * inv_std * lw[j].y + lb[j].y;
        o.z = (v[j].z - mean) * inv_std * lw[j].z + lb[j].z; o.w = (v[j].w - mean) * inv_std * lw[j].w + lb[j].w;
        if (a.y_out) reinterpret_cast<float4*>(a.y_out + (int64_t)row * a.d)[f] = o;
        if (a.yh) {
            unsigned h01, l01, h23, l23;
            nsplit2(o.x, o.y, h01, l01);
            nsplit2(o.z, o.w, h23, l23);
            reinterpret_cast<uint2*>(a.yh + (int64_t)row * a.d)[f] = make_uint2(h01, h23);
            reinterpret_cast<uint2*>(a.yl + (int64_t)row * a.d)[f] = make_uint2(l01, l23);
        }
    }
}

void launch_combine_ln(const CombineLnArgs& a, hipStream_t stream) {
    note_launch("k_combine_ln");
    if (a.rows <= 0) return;
    if (a.d % 4 || a.d > 4096 || !aligned16(a.x) || (a.partial && !aligned16(a.partial))) abort();   // step shapes always qualify (checked by the caller)
    hipLaunchKernelGGL(k_combine_ln, dim3(a.rows), dim3(256), 0, stream, a);
}

void launch_layernorm(const LnArgs& a, hipStream_t stream) {
    note_launch("k_layernorm");
    if (a.rows <= 0) return;
    bool reg = a.d % 4 == 0 && a.d <= 1024 && a.xmap.ld % 4 == 0 && a.xmap.batch_stride % 4 == 0 && a.ldy % 4 == 0 && aligned16(a.x) &&
               (!a.y || aligned16(a.y)) && (!a.shift || (a.ldmod % 4 == 0 && aligned16(a.shift) && aligned16(a.scale))) &&
               (!a.w || aligned16(a.w)) && (!a.b || aligned16(a.b));
    if (reg) hipLaunchKernelGGL(k_layernorm_reg, dim3((a.rows + 3) / 4), dim3(256), 0, stream, a);
    else if (a.partial) abort();   // the fused reduction exists in the register kernel only (step shapes always qualify)
    else hipLaunchKernelGGL(k_layernorm, dim3((a.rows + 3) / 4), dim3(256), 0, stream, a);
}

__global__ __launch_bounds__(64) void k_rmsnorm_alpha(float* x, const float* alpha, float eps, int rows, int d) {
    // tensor_util.go:273-326: unbiased variance about the mean, x itself is NOT centred
    const int lane = threadIdx.x, row = blockIdx.x;
    float* r = x + (int64_t)row * d;
    double s = 0.0;
    for (int i = lane; i < d; i += WAVE) s += (double)r[i];
    const double mean = wave_sum(s) / (double)d;
    double v = 0.0;
    for (int i = lane; i < d; i += WAVE) { double dlt = (double)r[i] - mean; v += dlt * dlt; }
    double var = wave_sum(v);
    if (d > 1) var /= (double)(d - 1);
    const float inv = (float)(1.0 / sqrt(var + (double)eps));
    for (int i = lane; i < d; i += WAVE) r[i] = r[i] * inv * alpha[i];
}
void launch_rmsnorm_alpha(float* x, const float* alpha, float eps, int rows, int d, hipStream_t stream) {
    hipLaunchKernelGGL(k_rmsnorm_alpha, dim3(rows), dim3(64), 0, stream, x, alpha, eps, rows, d);
}

}  // namespace ptts
