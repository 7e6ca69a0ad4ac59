// Probe: HBM read rate of the many-row GEMMs' activation access pattern on MI355X.
// A row-major f32 matrix [M][K] is read once by 256-row panels: a block owns a panel and walks K; per step every row of the
// panel contributes RB contiguous bytes (RB = 128 is what a 32-k step of k_gemm3 / k_gemm4 asks for).  The whole panel is one
// contiguous region of memory either way -- what changes with RB is how long the contiguous run is that a panel asks of the
// memory system at a time (256 rows x RB bytes, row stride 4 K bytes).  `linear` reads the same bytes front to back.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/panel_read tools/probes/panel_read.hip ; run: /tmp/panel_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// 512 threads; per step the block reads 256 rows x RB bytes = 16 RB 16-byte pieces per ... thread t reads piece p = t + 512 i
template <int RB, int DEPTH>
__global__ void __launch_bounds__(512) k_panel(const uint4* __restrict__ a, int M, int K, float* out) {
    constexpr int PPR = RB / 16;                 // 16-byte pieces per row and step
    constexpr int PIECES = 256 * PPR;            // per step
    constexpr int PPT = PIECES / 512;            // per thread and step (RB >= 32)
    const int npan = M / 256, rowq = K / 4;      // uint4 per row
    const int nsteps = K * 4 / RB;
    float acc = 0.f;
    for (int pan = blockIdx.x; pan < npan; pan += gridDim.x) {
        const uint4* base = a + (size_t)pan * 256 * rowq;
        uint4 v[DEPTH][PPT];
        auto load = [&](int s, uint4 (&d)[PPT]) {
#pragma unroll
            for (int i = 0; i < PPT; i++) {
                const int p = threadIdx.x + 512 * i, row = p / PPR, c = p % PPR;
                d[i] = base[(size_t)row * rowq + s * PPR + c];
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH; d++) load(d, v[d]);
        for (int s = 0; s < nsteps; s += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
#pragma unroll
                for (int i = 0; i < PPT; i++) acc += __uint_as_float(v[d][i].x ^ v[d][i].y ^ v[d][i].z ^ v[d][i].w) * 1e-30f;
                if (s + d + DEPTH < nsteps) load(s + d + DEPTH, v[d]);
            }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int DEPTH>
__global__ void __launch_bounds__(512) k_linear(const uint4* __restrict__ a, size_t n16, float* out) {
    float acc = 0.f;
    const size_t per = n16 / gridDim.x;
    const uint4* base = a + per * blockIdx.x;
    for (size_t i = threadIdx.x; i < per; i += 512 * DEPTH) {
        uint4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) v[d] = i + 512 * d < per ? base[i + 512 * d] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int d = 0; d < DEPTH; d++) acc += __uint_as_float(v[d].x ^ v[d].y ^ v[d].z ^ v[d].w) * 1e-30f;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <typename F>
static void timeit(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-52s %8.1f us  %7.0f GB/s\n", name, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e9);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
    const size_t bytes = (size_t)1 << 30;   // 1 GiB: beyond L2 + Infinity Cache
    uint4* a; float* out;
    CK(hipMalloc((void**)&a, bytes)); CK(hipMemset(a, 0x11, bytes));
    CK(hipMalloc((void**)&out, 64));
    for (int K : {512, 2048}) {
        const int M = (int)(bytes / ((size_t)K * 4));
        printf("matrix %d x %d f32 (row %d bytes), 256 blocks x 512 threads, one 256-row panel at a time\n", M, K, K * 4);
        timeit("  linear, 4 loads in flight per thread", (double)bytes, [&] { hipLaunchKernelGGL((k_linear<4>), dim3(256), dim3(512), 0, 0, a, bytes / 16, out); });
        timeit("  linear, 8 loads in flight per thread", (double)bytes, [&] { hipLaunchKernelGGL((k_linear<8>), dim3(256), dim3(512), 0, 0, a, bytes / 16, out); });
        timeit("  panel, 128 B per row and step, 2 steps in flight", (double)bytes, [&] { hipLaunchKernelGGL((k_panel<128, 2>), dim3(256), dim3(512), 0, 0, a, M, K, out); });
        timeit("  panel, 128 B per row and step, 4 steps in flight", (double)bytes, [&] { hipLaunchKernelGGL((k_panel<128, 4>), dim3(256), dim3(512), 0, 0, a, M, K, out); });
        timeit("  panel, 256 B per row and step, 2 steps in flight", (double)bytes, [&] { hipLaunchKernelGGL((k_panel<256, 2>), dim3(256), dim3(512), 0, 0, a, M, K, out); });
        timeit("  panel, 512 B per row and step, 1 step in flight", (double)bytes, [&] { hipLaunchKernelGGL((k_panel<512, 1>), dim3(256), dim3(512), 0, 0, a, M, K, out); });
        timeit("  panel, 512 B per row and step, 2 steps in flight", (double)bytes, [&] { hipLaunchKernelGGL((k_panel<512, 2>), dim3(256), dim3(512), 0, 0, a, M, K, out); });
        timeit("  panel, 1024 B per row and step, 1 step in flight", (double)bytes, [&] { hipLaunchKernelGGL((k_panel<1024, 1>), dim3(256), dim3(512), 0, 0, a, M, K, out); });
    }
    return 0;
}
