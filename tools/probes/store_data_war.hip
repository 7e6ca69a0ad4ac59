// Probe: can a VALU write to a data register of an ISSUED global_store_dwordx4 still change what is stored, on MI355X, when the
// other workgroup of the CU keeps the vector-memory path busy with LDS-DMA?  (gfx940+ documents 2 wait states between a store of
// more than 8 bytes and a VALU write of its data registers; hipcc inserts them.  The question is whether more are needed when
// the store has to queue behind another wave's traffic -- the signature of k_gemm4's run-to-run differences was lanes 48-63 of
// the THIRD data register of an epilogue store whose registers the RoPE code reuses.)
// Workgroups 0..255 ("hammer", one per CU): an endless-looking K loop of LDS-DMA pieces + ds_reads, like k_gemm4's.
// Workgroups 256..511 ("store", the second workgroup of each CU): every wave stores a tagged float4 per lane, waits DELAY wait
// states, then overwrites the store's third data register with 0xDEADBEEF; the host counts poisoned dwords in memory.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/store_data_war tools/probes/store_data_war.hip ; run: /tmp/store_data_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void* g, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}

// DELAY: 0 -> no wait state, 1 -> s_nop 1 (2 wait states: the documented requirement), 2 -> s_nop 7, 3 -> 8 x s_nop 7, 4 -> 32 x s_nop 7
template <int DELAY>
__global__ void __launch_bounds__(256, 2) k_probe(const char* __restrict__ src, unsigned* __restrict__ out, int T, int hammer_on, float* sink) {
    __shared__ __attribute__((aligned(1024))) char lds[65536];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (blockIdx.x < 256) {   // hammer: DMA 32 KB per step into alternating stages, read 20 KB of the other stage per wave
        if (!hammer_on) return;
        const unsigned lds0 = (unsigned)(uintptr_t)lds;
        float acc = 0.f;
        for (int t = 0; t < T * 4; t++) {
            const char* s = src + (size_t)((blockIdx.x * 7 + t) % 2048) * 32768 + wave * 8192 + lane * 16;
#pragma unroll
            for (int p = 0; p < 8; p++) glds16(s + p * 1024, lds0 + (t & 1) * 32768 + wave * 8192 + p * 1024);
            const char* st = lds + ((t + 1) & 1) * 32768;
#pragma unroll
            for (int i = 0; i < 20; i++) {
                const uint4 v = *reinterpret_cast<const uint4*>(st + ((wave * 5 + i) % 32) * 1024 + lane * 16);
                acc += __uint_as_float(v.x ^ v.w) * 1e-30f;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (acc == 123.456f) sink[0] = acc;
        return;
    }
    const int b = blockIdx.x - 256;
    unsigned* base = out + ((size_t)(b * 4 + wave) * T) * 256 + lane * 4;   // 1 KB per wave and iteration
    for (int it = 0; it < T; it++) {
        unsigned* p = base + (size_t)it * 256;
        const unsigned tag = 0x40000000u + (unsigned)it;
        if (DELAY == 0)
            asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %1\n\tv_mov_b32 v23, %1\n\ts_nop 4\n\t"
                         "global_store_dwordx4 %0, v[20:23], off\n\tv_mov_b32 v22, 0xdeadbeef" ::"v"(p), "v"(tag) : "v20", "v21", "v22", "v23", "memory");
        else if (DELAY == 1)
            asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %1\n\tv_mov_b32 v23, %1\n\ts_nop 4\n\t"
                         "global_store_dwordx4 %0, v[20:23], off\n\ts_nop 1\n\tv_mov_b32 v22, 0xdeadbeef" ::"v"(p), "v"(tag) : "v20", "v21", "v22", "v23", "memory");
        else if (DELAY == 2)
            asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %1\n\tv_mov_b32 v23, %1\n\ts_nop 4\n\t"
                         "global_store_dwordx4 %0, v[20:23], off\n\ts_nop 7\n\tv_mov_b32 v22, 0xdeadbeef" ::"v"(p), "v"(tag) : "v20", "v21", "v22", "v23", "memory");
        else if (DELAY == 3)
            asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %1\n\tv_mov_b32 v23, %1\n\ts_nop 4\n\t"
                         "global_store_dwordx4 %0, v[20:23], off\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                         "v_mov_b32 v22, 0xdeadbeef" ::"v"(p), "v"(tag) : "v20", "v21", "v22", "v23", "memory");
        else {
            asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %1\n\tv_mov_b32 v23, %1\n\ts_nop 4\n\t"
                         "global_store_dwordx4 %0, v[20:23], off\n\t"
                         "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                         "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                         "v_mov_b32 v22, 0xdeadbeef" ::"v"(p), "v"(tag) : "v20", "v21", "v22", "v23", "memory");
        }
    }
}

template <int DELAY>
static void run(const char* name, const char* src, unsigned* out, int T, int hammer, float* sink, std::vector<unsigned>& h) {
    const size_t n = (size_t)256 * 4 * T * 256;
    CK(hipMemset(out, 0, n * 4));
    hipLaunchKernelGGL((k_probe<DELAY>), dim3(512), dim3(256), 0, 0, src, out, T, hammer, sink);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost));
    size_t poisoned = 0, other = 0, by_dword[4] = {0, 0, 0, 0}, by_quarter[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        const unsigned want = 0x40000000u + (unsigned)((i / 256) % T);
        if (h[i] == want) continue;
        if (h[i] == 0xdeadbeefu) { poisoned++; by_dword[i & 3]++; by_quarter[((i & 255) >> 2) >> 4]++; }
        else other++;
    }
    printf("%-34s hammer %d: %zu poisoned dwords of %zu (by dword %zu %zu %zu %zu; by 16-lane quarter %zu %zu %zu %zu), %zu other mismatches\n", name, hammer, poisoned, n,
           by_dword[0], by_dword[1], by_dword[2], by_dword[3], by_quarter[0], by_quarter[1], by_quarter[2], by_quarter[3], other);
}

int main() {
    const int T = 400;
    char* src; unsigned* out; float* sink;
    CK(hipMalloc((void**)&src, (size_t)2048 * 32768)); CK(hipMemset(src, 0x11, (size_t)2048 * 32768));
    const size_t n = (size_t)256 * 4 * T * 256;
    CK(hipMalloc((void**)&out, n * 4)); CK(hipMalloc((void**)&sink, 64));
    std::vector<unsigned> h(n);
    for (int hammer = 0; hammer < 2; hammer++) {
        run<0>("no wait state", src, out, T, hammer, sink, h);
        run<1>("s_nop 1 (2 wait states, documented)", src, out, T, hammer, sink, h);
        run<2>("s_nop 7", src, out, T, hammer, sink, h);
        run<3>("8 x s_nop 7", src, out, T, hammer, sink, h);
        run<4>("32 x s_nop 7", src, out, T, hammer, sink, h);
    }
    return 0;
}
