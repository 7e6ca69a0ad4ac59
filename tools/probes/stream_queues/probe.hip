// Which hardware queue does the HIP runtime give each user stream once one of them carries a CU mask?  (round 4: a fourth user stream beside the continuous engine's
// CU-masked decoder stream halved the engine's throughput -- "every step ran as if confined to the decoder's CUs and serialised with its kernels".)
// Run under `rocprofv3 --kernel-trace`: the trace's Queue_Id column per kernel name k<stream index> shows the mapping; each kernel also reports how many distinct
// CUs (XCC_ID, CU id from HW_ID) its 1024 workgroups ran on, and the host times a long kernel on the masked stream beside short ones on the others.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "hip error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <int ID>
__global__ void k(unsigned* cu_seen, int spin) {
    // HW_REG_HW_ID (id 4): bits 11:8 CU id, 15:13 SE id ...; HW_REG_XCC_ID (id 20)
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
    unsigned cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | ((xcc & 0xf) << 7);   // (se, cu) within the XCD + the XCD
    if (threadIdx.x == 0) atomicOr(&cu_seen[ID * 64 + (cu >> 5)], 1u << (cu & 31));
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(64);
}

int main(int argc, char** argv) {
    const int n_extra = argc > 1 ? atoi(argv[1]) : 1;      // plain streams created AFTER the masked one
    const int masked_first = argc > 2 ? atoi(argv[2]) : 0; // create the masked stream before the priority streams
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t s[8]; const char* kind[8]; int ns = 0;
    uint32_t mask[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0};   // CUs 0..127
    auto mk_masked = [&] { CK(hipExtStreamCreateWithCUMask(&s[ns], 8, mask)); kind[ns++] = "cu-masked(128)"; };
    if (masked_first) mk_masked();
    CK(hipStreamCreateWithPriority(&s[ns], hipStreamNonBlocking, hi)); kind[ns++] = "high priority";
    CK(hipStreamCreateWithPriority(&s[ns], hipStreamNonBlocking, lo)); kind[ns++] = "low priority";
    if (!masked_first) mk_masked();
    for (int i = 0; i < n_extra; i++) { CK(hipStreamCreateWithFlags(&s[ns], hipStreamNonBlocking)); kind[ns++] = "plain"; }
    unsigned* seen; CK(hipMalloc(&seen, 8 * 64 * 4)); CK(hipMemset(seen, 0, 8 * 64 * 4));
    auto launch = [&](int i, int spin) {
        switch (i) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(2048), dim3(64), 0, s[0], seen, spin); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(2048), dim3(64), 0, s[1], seen, spin); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(2048), dim3(64), 0, s[2], seen, spin); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(2048), dim3(64), 0, s[3], seen, spin); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(2048), dim3(64), 0, s[4], seen, spin); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(2048), dim3(64), 0, s[5], seen, spin); break;
            default: break;
        }
    };
    for (int rep = 0; rep < 3; rep++) for (int i = 0; i < ns; i++) launch(i, 8);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(8 * 64); CK(hipMemcpy(h.data(), seen, h.size() * 4, hipMemcpyDeviceToHost));
    printf("# streams in creation order; distinct (XCD, SE, CU) slots the stream's workgroups ran on\n");
    for (int i = 0; i < ns; i++) { int c = 0; for (int w = 0; w < 64; w++) c += __builtin_popcount(h[i * 64 + w]); printf("stream %d (%s): kernel k<%d> ran on %d distinct CU slots\n", i, kind[i], i, c); }
    // interference: a long kernel on the masked stream, then time 200 short launches on each other stream while it runs
    int mi = masked_first ? 0 : 2;
    for (int i = 0; i < ns; i++) {
        if (i == mi) continue;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        launch(mi, 20000);                      // ~ tens of ms on its 128 CUs
        CK(hipEventRecord(e0, s[i]));
        for (int r = 0; r < 200; r++) launch(i, 0);
        CK(hipEventRecord(e1, s[i]));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipDeviceSynchronize());
        printf("200 short launches on stream %d (%s) beside a long kernel on the masked stream: %.3f ms (%.1f us each)\n", i, kind[i], ms, 5.0f * ms);
    }
    return 0;
}
