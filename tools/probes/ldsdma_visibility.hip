// Probe: is "issuing wave's s_waitcnt vmcnt(0), then s_barrier" enough for OTHER waves' ds_reads to see LDS-DMA data on MI355X
// when two workgroups share a CU?  Every workgroup (4 waves, 64 KB of LDS = two 32-KB stages: two workgroups per CU) loops:
//   DMA tile t+1 (32 KB, eight 1-KB global_load_lds_dwordx4 pieces per wave, every dword of tile t holds the value t) into stage (t+1)&1
//   read 20 KB of stage t&1 per wave with ds_read_b128 -- 4 KB of its own pieces, 16 KB of the other waves' -- and count dwords != t
//   s_waitcnt vmcnt(0); s_barrier
// ORDER 0: the DMA is issued before the reads of the step (what k_gemm4's two-stage form does; only the last step reads right after
// a barrier); ORDER 1: the reads come FIRST, straight after the barrier (every step has the narrow window); ORDER 2: as 1 with a
// second barrier.  The co-resident workgroup runs the same loop out of phase: its ds_read stream competes for the LDS while the
// DMA data of this one arrives.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/ldsdma_visibility tools/probes/ldsdma_visibility.hip ; run: /tmp/ldsdma_visibility
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

template <int ORDER>
__global__ void __launch_bounds__(256, 2) k_probe(const unsigned* __restrict__ tiles, int T, int delay, unsigned long long* bad, unsigned* first_bad) {
    __shared__ __attribute__((aligned(1024))) char lds[65536];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    auto issue = [&](int t) {
        const char* src = reinterpret_cast<const char*>(tiles) + (size_t)(t % 64) * 32768 + wave * 8192 + lane * 16;
#pragma unroll
        for (int p = 0; p < 8; p++) glds16(src + p * 1024, lds0 + (t & 1) * 32768 + wave * 8192 + p * 1024);
    };
    unsigned long long nbad = 0;
    auto verify = [&](int t) {
        const unsigned want = (unsigned)(t % 64);
        const char* st = lds + (t & 1) * 32768;
#pragma unroll
        for (int i = 0; i < 20; i++) {   // 4 pieces of its own, 16 of the other three waves
            const int piece = i < 4 ? wave * 8 + i : ((wave + 1 + (i - 4) / 6) % 4) * 8 + (i - 4) % 6;
            const uint4 v = *reinterpret_cast<const uint4*>(st + piece * 1024 + lane * 16);
            const int b = (v.x != want) + (v.y != want) + (v.z != want) + (v.w != want);
            if (b) { nbad += b; atomicMin(first_bad, (unsigned)t); }
        }
    };
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < T; t++) {
        if (ORDER == 0 && t + 1 < T) issue(t + 1);
        for (int d = 0; d < delay; d++) __builtin_amdgcn_s_sleep(1);
        verify(t);
        if (ORDER != 0 && t + 1 < T) issue(t + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ORDER == 2) __syncthreads();
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int ORDER>
static void run(const char* name, const unsigned* tiles, int blocks, int T, int delay, unsigned long long* bad, unsigned* first) {
    CK(hipMemset(bad, 0, 8)); CK(hipMemset(first, 0xff, 4));
    hipLaunchKernelGGL((k_probe<ORDER>), dim3(blocks), dim3(256), 0, 0, tiles, T, delay, bad, first);
    CK(hipDeviceSynchronize());
    unsigned long long hb; unsigned hf;
    CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost));
    printf("%-58s blocks %4d steps %5d delay %2d : %llu stale dwords of %.3g read (first at step %d)\n", name, blocks, T, delay, hb,
           (double)blocks * T * 4 * 20 * 256.0, hb ? (int)hf : -1);
}

int main() {
    unsigned* tiles;   // 64 tiles of 32 KB: every dword of tile t is t
    CK(hipMalloc((void**)&tiles, 64 * 32768));
    std::vector<unsigned> h(64 * 8192);
    for (int t = 0; t < 64; t++) for (int i = 0; i < 8192; i++) h[(size_t)t * 8192 + i] = (unsigned)t;
    CK(hipMemcpy(tiles, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    unsigned long long* bad; unsigned* first;
    CK(hipMalloc((void**)&bad, 8)); CK(hipMalloc((void**)&first, 4));
    for (int rep = 0; rep < 2; rep++) {
        run<0>("DMA before the reads (two-stage GEMM order)", tiles, 512, 2000, 0, bad, first);
        run<1>("reads straight after the barrier", tiles, 512, 2000, 0, bad, first);
        run<1>("reads straight after the barrier, one workgroup per CU", tiles, 256, 2000, 0, bad, first);
        run<1>("reads after the barrier + s_sleep 1", tiles, 512, 2000, 1, bad, first);
        run<1>("reads after the barrier + 4 x s_sleep 1", tiles, 512, 2000, 4, bad, first);
        run<2>("reads after TWO barriers", tiles, 512, 2000, 0, bad, first);
        run<1>("reads straight after the barrier, 2048 workgroups", tiles, 2048, 500, 0, bad, first);
    }
    return 0;
}
