// Does v_mfma_f32_16x16x32_bf16 (and 32x32x16) give D = A x B + C when the D tuple PARTIALLY overlaps the C tuple (e.g. D = a[40:43], C = a[42:45])?
// hipcc (ROCm 7.2) emits such instructions for gfx950 when it slides an accumulator down by two registers (first cut of k_mimi_rowlin, round 4:
//   v_mfma_f32_16x16x32_bf16 a[0:3], v[154:157], v[94:97], a[2:5]).
// Each kernel: C registers preset to known per-lane values, one MFMA with D at offset SHIFT from C, long wait, read D; expected = in-place result of the
// same C (computed with D == C in a second statement).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "hip error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

// C lives in a[48:51]; D in a[48+SHIFT : 51+SHIFT]  (SHIFT in -3..4; 0 = in place)
#define KERNEL(NAME, DLO, DHI, R0, R1, R2, R3)                                                                                                     \
__global__ __launch_bounds__(64) void NAME(const bf16x8* A, const bf16x8* B, const float* C, float* out) {                                        \
    const bf16x8 a = A[threadIdx.x], b = B[threadIdx.x];                                                                                          \
    const float c0 = C[threadIdx.x * 4 + 0], c1 = C[threadIdx.x * 4 + 1], c2 = C[threadIdx.x * 4 + 2], c3 = C[threadIdx.x * 4 + 3];               \
    float o0, o1, o2, o3;                                                                                                                         \
    asm volatile("v_accvgpr_write_b32 a48, %[c0]\n\tv_accvgpr_write_b32 a49, %[c1]\n\tv_accvgpr_write_b32 a50, %[c2]\n\tv_accvgpr_write_b32 a51, %[c3]\n\t" \
                 "s_nop 7\n\ts_nop 7\n\t"                                                                                                       \
                 "v_mfma_f32_16x16x32_bf16 a[" #DLO ":" #DHI "], %[a], %[b], a[48:51]\n\t"                                                      \
                 "s_nop 15\n\ts_nop 15\n\t"                                                                                                     \
                 "v_accvgpr_read_b32 %[o0], a" #R0 "\n\tv_accvgpr_read_b32 %[o1], a" #R1 "\n\tv_accvgpr_read_b32 %[o2], a" #R2 "\n\tv_accvgpr_read_b32 %[o3], a" #R3 "\n\t" \
                 "s_nop 7"                                                                                                                      \
                 : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3)                                                                \
                 : [a] "v"(a), [b] "v"(b), [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3)                                                \
                 : "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55");                                          \
    float* o = out + threadIdx.x * 4; o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;                                                                 \
}
KERNEL(k_s0, 48, 51, 48, 49, 50, 51)
KERNEL(k_sm2, 46, 49, 46, 47, 48, 49)
KERNEL(k_sm4, 44, 47, 44, 45, 46, 47)
KERNEL(k_sp2, 50, 53, 50, 51, 52, 53)
KERNEL(k_sp4, 52, 55, 52, 53, 54, 55)

int main() {
    std::vector<unsigned short> ha(64 * 8), hb(64 * 8);
    std::vector<float> hc(256);
    unsigned s = 777;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) & 0xffffu; };
    for (auto& x : ha) { float f = ((int)(rnd() % 2001) - 1000) / 500.0f; unsigned u; memcpy(&u, &f, 4); x = (unsigned short)(u >> 16); }
    for (auto& x : hb) { float f = ((int)(rnd() % 2001) - 1000) / 500.0f; unsigned u; memcpy(&u, &f, 4); x = (unsigned short)(u >> 16); }
    for (int i = 0; i < 256; i++) hc[i] = 100.0f * (i % 4 + 1) + (i / 4);   // element e of lane l: 100 (e + 1) + l
    bf16x8 *A, *B; float *C, *O;
    CK(hipMalloc(&A, 64 * 16)); CK(hipMalloc(&B, 64 * 16)); CK(hipMalloc(&C, 1024)); CK(hipMalloc(&O, 1024));
    CK(hipMemcpy(A, ha.data(), 64 * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hb.data(), 64 * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(C, hc.data(), 1024, hipMemcpyHostToDevice));
    struct V { const char* name; int shift; void (*fn)(const bf16x8*, const bf16x8*, const float*, float*); };
    const V vs[] = {{"D = C (in place)", 0, k_s0}, {"D = C - 2", -2, k_sm2}, {"D = C - 4 (disjoint)", -4, k_sm4},
                    {"D = C + 2", 2, k_sp2}, {"D = C + 4 (disjoint)", 4, k_sp4}};
    std::vector<float> ref(256), got(256);
    hipLaunchKernelGGL(k_s0, dim3(1), dim3(64), 0, 0, A, B, C, O); CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref.data(), O, 1024, hipMemcpyDeviceToHost));
    printf("# v_mfma_f32_16x16x32_bf16 with the destination tuple offset from the C tuple; C[e] of lane l = 100 (e + 1) + l; reference = the in-place form\n");
    for (const V& v : vs) {
        int bad = 0, lo = 64, hi = -1; unsigned em = 0; int fl = -1, fe = -1;
        for (int rep = 0; rep < 10; rep++) {
            CK(hipMemset(O, 0, 1024));
            hipLaunchKernelGGL(v.fn, dim3(1), dim3(64), 0, 0, A, B, C, O); CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), O, 1024, hipMemcpyDeviceToHost));
            for (int l = 0; l < 64; l++) for (int e = 0; e < 4; e++)
                if (memcmp(&got[l * 4 + e], &ref[l * 4 + e], 4) != 0) { bad++; em |= 1u << e; if (l < lo) lo = l; if (l > hi) hi = l; if (fl < 0) { fl = l; fe = e; } }
        }
        printf("%-22s: bad=%d", v.name, bad);
        if (bad) printf("  lanes %d..%d elements mask 0x%x  (lane %d element %d: got %g want %g, difference %g)", lo, hi, em, fl, fe, got[fl * 4 + fe], ref[fl * 4 + fe], got[fl * 4 + fe] - ref[fl * 4 + fe]);
        printf("\n");
    }
    return 0;
}
