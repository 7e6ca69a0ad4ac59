// norm.hip -- LayerNorm (K3; linear.go:265-329 / nn_ops.go:79-149) with the adaLN modulation of the flow net (K11)
// and the split-K reduction + residual update of the AR step fused in front.
#include "../../include/ptts.h"
#include "common.h"
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

void launch_layernorm(const LnArgs& a, hipStream_t stream) {
    note_launch("k_layernorm");
    if (a.rows <= 0) return;
    bool reg = a.d % 4 == 0 && a.d <= 1024 && a.xmap.ld % 4 == 0 && a.xmap.batch_stride % 4 == 0 && a.ldy % 4 == 0 && aligned16(a.x) &&
               (!a.y || aligned16(a.y)) && (!a.shift || (a.ldmod % 4 == 0 && aligned16(a.shift) && aligned16(a.scale))) &&
               (!a.w || aligned16(a.w)) && (!a.b || aligned16(a.b));
    if (reg) hipLaunchKernelGGL(k_layernorm_reg, dim3((a.rows + 3) / 4), dim3(256), 0, stream, a);
    else if (a.partial) throw Error(PTTS_EINVAL, "ptts-hip: internal: the fused split-K reduction needs rows of d % 4 == 0, d <= 1024, 16-byte aligned");   // register kernel only
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
