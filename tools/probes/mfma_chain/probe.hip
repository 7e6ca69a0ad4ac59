// How fast does a wave issue v_mfma_f32_16x16x32_bf16 when consecutive instructions accumulate into the SAME registers (the hi / lo pair of every product in
// this library: acc = mfma(w, x_hi, acc); acc = mfma(w, x_lo, acc)) against 2 / 4 / 8 independent accumulators in rotation -- and what a second / fourth wave on
// the SIMD makes of it.  hipcc --offload-arch=gfx950 -O2 probe.hip -o build/probe; prints cycles per MFMA per wave and per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "hip error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <int NACC, int SHAPE>   // SHAPE 0: 16x16x32, 1: 32x32x16 (acc 16 regs: modelled with four f32x4? no: use the builtin's own type)
__global__ __launch_bounds__(1024) void k_chain(const bf16x8* A, const bf16x8* B, float* out, unsigned long long* ticks, int iters) {
    const bf16x8 a = A[threadIdx.x & 63], b = B[threadIdx.x & 63];
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64 / NACC; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC>
static void run(int waves, const bf16x8* A, const bf16x8* B, float* O, unsigned long long* T) {
    const int iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_chain<NACC, 0>), dim3(1), dim3(64 * waves), 0, 0, A, B, O, T, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_chain<NACC, 0>), dim3(1), dim3(64 * waves), 0, 0, A, B, O, T, iters);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> t(waves);
    CK(hipMemcpy(t.data(), T, waves * 8, hipMemcpyDeviceToHost));
    const double n = 64.0 * iters;   // MFMAs per wave
    const double simd_waves = (waves + 3) / 4;   // waves per SIMD (waves are dealt round the 4 SIMDs)
    printf("accumulators %d  waves/CU %2d (%.0f per SIMD): %7.2f ns per MFMA per wave  -> %6.2f ns per MFMA per SIMD   (s_memtime ticks per MFMA per wave %.2f)\n", NACC, waves, simd_waves,
           1e6 * ms / n, 1e6 * ms / n / simd_waves, (double)t[0] / n);
}

int main() {
    bf16x8 *A, *B; float* O; unsigned long long* T;
    CK(hipMalloc(&A, 64 * 16)); CK(hipMalloc(&B, 64 * 16)); CK(hipMalloc(&O, 1024 * 4)); CK(hipMalloc(&T, 16 * 8));
    CK(hipMemset(A, 0, 64 * 16)); CK(hipMemset(B, 0, 64 * 16));
    for (int waves : {1, 4, 8, 16}) {
        run<1>(waves, A, B, O, T);
        run<2>(waves, A, B, O, T);
        run<4>(waves, A, B, O, T);
        run<8>(waves, A, B, O, T);
    }
    return 0;
}
