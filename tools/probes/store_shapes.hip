// Probe: HBM write rate on MI355X as a function of the SHAPE of one wave's 16-byte-per-lane store instruction.
// A GEMM epilogue on v_mfma_f32_16x16x32 holds, per lane, four consecutive f32 columns of one of 16 rows: one
// global_store_dwordx4 then covers 16 rows x 64 contiguous bytes (half a 128-byte line per row).  The probe writes a
// row-major f32 matrix [M][N] (256-row x 256-column tiles, 8 waves x 32 rows per block, like k_gemm3 / k_gemm4) with
//   frag   16 rows x  64 B per instruction (the MFMA epilogue's shape; the other half of each line comes from the next instruction)
//   line    8 rows x 128 B per instruction (whole lines)
//   row     1 row  x 1 KB  per instruction (whole 1-KB runs)
// and a linear front-to-back fill for reference.  Values are register-resident: nothing is read.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/store_shapes tools/probes/store_shapes.hip ; run: /tmp/store_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0 frag, 1 line, 2 row.  Tile = 256 rows x 256 columns (256 KB); wave w owns rows 32 w .. 32 w + 31 (32 KB = 32 instructions).
template <int MODE>
__global__ void __launch_bounds__(512) k_tiles(float* c, int M, int N, float seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncol = N / 256, ntiles = (M / 256) * ncol;
    const float4 v = make_float4(seed + lane, seed, seed, seed);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int m0 = (t / ncol) * 256 + wave * 32, n0 = (t % ncol) * 256;
        float* base = c + (size_t)m0 * N + n0;
#pragma unroll
        for (int i = 0; i < 32; i++) {
            int row, col;
            if (MODE == 0) { row = (i >> 4) * 16 + (lane & 15); col = (i & 15) * 16 + (lane >> 4) * 4; }   // acc[t = i >> 4][n = i & 15]
            else if (MODE == 1) { row = (i >> 3) * 8 + (lane >> 3); col = (i & 7) * 32 + (lane & 7) * 4; }
            else { row = i; col = lane * 4; }
            *reinterpret_cast<float4*>(base + (size_t)row * N + col) = v;
        }
    }
}

__global__ void __launch_bounds__(512) k_fill(float4* c, size_t n16, float seed) {
    const size_t per = n16 / gridDim.x;
    float4* base = c + per * blockIdx.x;
    const float4 v = make_float4(seed, seed, seed, seed);
    for (size_t i = threadIdx.x; i < per; i += 512) base[i] = v;
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
    printf("%-64s %8.1f us  %7.0f GB/s\n", name, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e9);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
    const size_t bytes = (size_t)768 << 20;   // 768 MiB (the qkv output of one batch)
    float* c;
    CK(hipMalloc((void**)&c, bytes));
    for (int N : {512, 1536, 2048}) {
        const int M = (int)(bytes / ((size_t)N * 4)) / 256 * 256;
        const double b = (double)M * N * 4;
        printf("matrix %d x %d f32, 256 x 256 tiles\n", M, N);
        for (int grid : {256, 512}) {
            char nm[128];
            snprintf(nm, sizeof nm, "  frag: 16 rows x 64 B per instruction, %d blocks", grid);
            timeit(nm, b, [&] { hipLaunchKernelGGL((k_tiles<0>), dim3(grid), dim3(512), 0, 0, c, M, N, 1.0f); });
            snprintf(nm, sizeof nm, "  line:  8 rows x 128 B per instruction, %d blocks", grid);
            timeit(nm, b, [&] { hipLaunchKernelGGL((k_tiles<1>), dim3(grid), dim3(512), 0, 0, c, M, N, 2.0f); });
            snprintf(nm, sizeof nm, "  row:   1 row x 1 KB per instruction, %d blocks", grid);
            timeit(nm, b, [&] { hipLaunchKernelGGL((k_tiles<2>), dim3(grid), dim3(512), 0, 0, c, M, N, 3.0f); });
        }
        timeit("  linear fill, 256 blocks", (double)bytes, [&] { hipLaunchKernelGGL(k_fill, dim3(256), dim3(512), 0, 0, (float4*)c, bytes / 16, 4.0f); });
        timeit("  linear fill, 1024 blocks", (double)bytes, [&] { hipLaunchKernelGGL(k_fill, dim3(1024), dim3(512), 0, 0, (float4*)c, bytes / 16, 4.0f); });
    }
    return 0;
}
