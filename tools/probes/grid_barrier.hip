// Probe: what does a grid-wide barrier cost on MI355X (256 CUs, 8 XCDs with non-coherent L2s)?
// One block per CU, R rounds of {touch some memory, barrier}.  Variants:
//   0  one counter in device memory, agent-scope atomic add + relaxed spin (monotonic target, no reset)
//   1  two-level: per-XCD counter (blockIdx % 8 = XCD), the last arriver of each XCD bumps the global one
//   2  as 0 but the spin uses s_sleep between polls
//   3  barrier among the blocks of ONE XCD only (blockIdx % 8), workgroup-scope atomic add (executes in that XCD's L2) +
//      agent-scope poll (skips the CU's vector cache)
//   4  as 3, polling with a workgroup-scope atomic add of 0
// The probe also prints the XCC_ID hardware register of the first 16 blocks: variants 3/4 rest on blockIdx % 8 = XCD.
// Every spin is bounded (bail out after SPIN_MAX polls and flag it) so a lost block cannot hang the box.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/grid_barrier tools/probes/grid_barrier.hip ; run: /tmp/grid_barrier [blocks] [threads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define SPIN_MAX 20000   // polls; a failed round aborts the whole kernel (every block sees *fail at its next round)

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int VARIANT>
__global__ void __launch_bounds__(1024) k_probe(unsigned* ctr, unsigned* xcd_ctr, int rounds, unsigned* fail, float* scratch) {
    const unsigned nb = gridDim.x;
    float acc = 0.f;
    __shared__ unsigned s_abort;
    if (threadIdx.x == 0) s_abort = 0;
    for (int r = 0; r < rounds; r++) {
        if (threadIdx.x == 0 && ld_relaxed(fail)) s_abort = 1;
        acc += scratch[(blockIdx.x * 64 + (threadIdx.x & 63)) & 16383];
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);   // agent-scope release of this block's writes
            const unsigned target = (unsigned)(r + 1) * nb;
            if (VARIANT >= 3) {
                unsigned* c = &xcd_ctr[(blockIdx.x & 7) * 32];
                const unsigned tgt = (unsigned)(r + 1) * (nb / 8);
                __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                int spins = 0;
                for (;;) {
                    const unsigned v = VARIANT == 3 ? ld_relaxed(c) : __hip_atomic_fetch_add(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (v >= tgt) break;
                    if (++spins > SPIN_MAX || ld_relaxed(fail)) { __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                goto synced;
            }
            if (VARIANT == 1) {
                const unsigned x = blockIdx.x & 7, per = nb / 8;
                const unsigned old = __hip_atomic_fetch_add(&xcd_ctr[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == (unsigned)(r + 1) * per) __hip_atomic_fetch_add(ctr, per, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int spins = 0;
            while (ld_relaxed(ctr) < target) {
                if (VARIANT == 2) __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_MAX || ld_relaxed(fail)) { __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
    synced:
        __syncthreads();
        if (s_abort || ld_relaxed(fail)) break;
    }
    if (acc == 12345.f) scratch[0] = acc;
}

__global__ void k_xcc(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (3 << 11));   // HW_REG_XCC_ID[3:0]
}

template <int V>
static void run(const char* name, int blocks, int threads, int rounds) {
    unsigned *ctr, *xc, *fail; float* scratch;
    hipMalloc(&ctr, 4); hipMalloc(&xc, 8 * 32 * 4); hipMalloc(&fail, 4); hipMalloc(&scratch, 16384 * 4);
    hipMemset(scratch, 0, 16384 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipMemset(ctr, 0, 4); hipMemset(xc, 0, 8 * 32 * 4); hipMemset(fail, 0, 4);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_probe<V>, dim3(blocks), dim3(threads), 0, 0, ctr, xc, rounds, fail, scratch);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    unsigned f; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
    printf("%-28s blocks %4d threads %4d  %7.3f us per round%s\n", name, blocks, threads, best * 1e3f / rounds, f ? "  (SPIN BAILED OUT)" : "");
    fflush(stdout);
    hipFree(ctr); hipFree(xc); hipFree(fail); hipFree(scratch);
}

int main(int argc, char** argv) {
    const int rounds = 2000;
    int cfgs[][2] = {{256, 256}, {256, 1024}, {128, 1024}, {64, 1024}, {512, 512}};
    {
        unsigned* d; hipMalloc(&d, 64 * 4);
        hipLaunchKernelGGL(k_xcc, dim3(64), dim3(64), 0, 0, d);
        unsigned h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("XCC_ID of blocks 0..63:");
        for (int i = 0; i < 64; i++) printf(" %u", h[i]);
        printf("\n");
        fflush(stdout);
    }
    for (auto& c : cfgs) {
        run<3>("in-XCD, L2 atomics, ld poll", c[0], c[1], rounds);
        run<4>("in-XCD, L2 atomics, rmw poll", c[0], c[1], rounds);
        run<0>("flat counter", c[0], c[1], rounds);
        run<1>("two-level (per-XCD)", c[0], c[1], rounds);
        run<2>("flat counter + s_sleep", c[0], c[1], rounds);
    }
    return 0;
}
