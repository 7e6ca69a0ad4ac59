// Probe: can a SEPARATE persistent kernel, running beside a chain of dependent weight-streaming launches, pull the NEXT launch's
// weights into each XCD's L2 so that the chain's launches find them there?  (round 2 measured that a launch which prefetches for
// its successor is lengthened by what the successor saves -- its waves cannot retire before the prefetch returns.  A second
// kernel on a second stream has no such coupling; what it costs is whatever it takes from the chain's CUs and memory pipes.)
//   chain     L launches on stream A, launch i reads buffer i (rotating over nbuf buffers, working set > L2 + Infinity Cache),
//             all loads issued up front, one value per block written that the next launch reads first (a true dependency);
//             block 0 publishes "launch i has started" in a progress word
//   feeder    one launch on stream B per chain: 256 blocks (block p runs on XCD p % 8, like block p of a chain launch), for
//             i = 0 .. L-1: wait until launch i - AHEAD has started, then touch one dword of every 128-byte line that chain
//             block p will read in launch i
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/prefetch_stream tools/probes/prefetch_stream.hip ; run: /tmp/prefetch_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// NL 16-byte loads per thread, all issued before the first use; block b reads pieces b + G i (i < NL) of T x 16 bytes
template <int NL>
__global__ void __launch_bounds__(1024) k_read(const uint4* __restrict__ w, const float* __restrict__ dep, float* __restrict__ out, int idx, int* progress) {
    if (blockIdx.x == 0 && threadIdx.x == 0)   // "launch idx has started", and the XCD its block 0 landed on (block b of the launch runs on XCD (b + that) % 8)
        __hip_atomic_store(progress, ((idx + 1) << 4) | (int)(__builtin_amdgcn_s_getreg(6164) & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (idx == 6 && threadIdx.x == 0) progress[16 + blockIdx.x] = (int)(__builtin_amdgcn_s_getreg(6164) & 15);   // HW_REG_XCC_ID of this block (launch 5)
    const float d = dep[blockIdx.x & 63];
    float acc = d;
    const int T = blockDim.x, G = gridDim.x;
    const uint4* p = w + (size_t)blockIdx.x * T + threadIdx.x;
    uint4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) v[i] = p[(size_t)i * G * T];
#pragma unroll
    for (int i = 0; i < NL; i++) acc += __uint_as_float(v[i].x ^ v[i].y ^ v[i].z ^ v[i].w) * 1e-30f;
    __shared__ float red[16];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += red[i];
        out[blockIdx.x & 63] = s * 0.f + 1.0f;
    }
}

struct Sched { const char* buf[64]; };

// feeder block p touches, for launch i, the lines of chain block p: NL pieces of T16 bytes at p * T16 + j * G * T16
__global__ void __launch_bounds__(256) k_feed(Sched s, int nbuf, int L, int NL, int T16, int G, int ahead, const int* progress, float* sink, int first_off) {
    unsigned acc = 0;
    const int lines_per_piece = T16 / 128, lines = NL * lines_per_piece;
    const int my_xcc = (int)(__builtin_amdgcn_s_getreg(6164) & 7);
    __shared__ int sh_off;
    if (threadIdx.x == 0) sh_off = first_off;
    __syncthreads();
    for (int i = 0; i < L; i++) {
        const int need = i - ahead + 1;   // launches that must have started before launch i may be fed
        if (need > 0) {
            if (threadIdx.x == 0) {
                int spins = 0, p;
                while (((p = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 4) < need && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(8);
                sh_off = p & 7;   // every grid of the chain is a multiple of 8 blocks: the next launch starts on the same XCD
            }
            __syncthreads();
        }
        // the chain block of the same rank on THIS XCD: chain block b runs on XCD (b + off) % 8
        const int b = 8 * (blockIdx.x >> 3) + ((my_xcc - sh_off) & 7);
        if (i == 6 && threadIdx.x == 0) const_cast<int*>(progress)[16 + 512 + b] = my_xcc;   // (check) the XCD that fed chain block b of launch 6
        const char* base = s.buf[i % nbuf] + (size_t)b * T16;
        for (int li = threadIdx.x; li < lines; li += blockDim.x) {
            const int j = li / lines_per_piece, within = li % lines_per_piece;
            acc |= *reinterpret_cast<const unsigned*>(base + (size_t)j * G * T16 + (size_t)within * 128);
        }
    }
    if (acc == 0x7fc01234u) sink[0] = 1.0f;
}

template <int NL>
static void chain(const std::vector<uint4*>& bufs, float* dep, int blocks, int threads, int L, int* progress, hipStream_t s) {
    for (int l = 0; l < L; l++)
        hipLaunchKernelGGL((k_read<NL>), dim3(blocks), dim3(threads), 0, s, bufs[l % bufs.size()], dep + (l & 1) * 64, dep + ((l + 1) & 1) * 64, l, progress);
}

template <int NL>
static void run(const char* name, const std::vector<uint4*>& bufs, float* dep, int blocks, int threads, int ahead, int* progress, float* sink, hipStream_t sa, hipStream_t sb) {
    const int L = 46, reps = 30;
    Sched sc;
    for (int i = 0; i < 64; i++) sc.buf[i] = reinterpret_cast<const char*>(bufs[i % bufs.size()]);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
    chain<NL>(bufs, dep, blocks, threads, L, progress, sa);
    CK(hipStreamEndCapture(sa, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float total = 0;
    int first_off = 0;
    for (int r = 0; r < reps + 3; r++) {
        CK(hipMemsetAsync(progress, 0, 4, sa));
        CK(hipStreamSynchronize(sa));
        if (ahead >= 0) hipLaunchKernelGGL(k_feed, dim3(blocks), dim3(256), 0, sb, sc, (int)bufs.size(), L, NL, threads * 16, blocks, ahead, progress, sink, first_off);
        CK(hipEventRecord(e0, sa));
        CK(hipGraphLaunch(ge, sa));
        CK(hipEventRecord(e1, sa));
        CK(hipStreamSynchronize(sa));
        CK(hipStreamSynchronize(sb));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) total += ms;
        int hp = 0;
        CK(hipMemcpy(&hp, progress, 4, hipMemcpyDeviceToHost));
        first_off = hp & 7;   // where the chain's block 0 landed in this replay (all grids are multiples of 8: the same in the next)
    }
    if (ahead >= 0) {
        std::vector<int> h(2048);
        CK(hipMemcpy(h.data(), progress, 8192, hipMemcpyDeviceToHost));
        int same = 0;
        for (int b = 0; b < blocks; b++) same += h[16 + b] == h[16 + 512 + b];
        printf("    (chain block p and feeder block p on the same XCD: %d of %d; chain block 0..9 XCC ids:", same, blocks);
        for (int b = 0; b < 10; b++) printf(" %d", h[16 + b]);
        printf(")\n");
    }
    const double bytes = (double)blocks * threads * NL * 16;
    printf("%-52s %6.2f MB/launch, %s : %6.2f us/launch (graph)  -> %7.1f GB/s\n", name, bytes / 1e6,
           ahead < 0 ? "no feeder       " : (ahead == 1 ? "feeder 1 ahead  " : (ahead == 2 ? "feeder 2 ahead  " : "feeder unpaced  ")), total * 1e3 / (L * reps),
           bytes / (total * 1e-3 / (L * reps)) / 1e9);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const size_t buf_bytes = (size_t)32 << 20;
    const int nbuf = 40;   // 40 x (6..25 MB touched): beyond L2 and the 256 MB Infinity Cache for the larger sizes, ~250 MB for 6.3 MB
    std::vector<uint4*> bufs(nbuf);
    for (auto& b : bufs) { CK(hipMalloc((void**)&b, buf_bytes)); CK(hipMemset(b, 0x11, buf_bytes)); }
    float* dep; int* progress; float* sink;
    CK(hipMalloc((void**)&dep, 128 * sizeof(float))); CK(hipMemset(dep, 0, 128 * sizeof(float)));
    CK(hipMalloc((void**)&progress, 8192)); CK(hipMalloc((void**)&sink, 64));
    printf("chain of 46 dependent launches replayed from a graph; feeder = a second kernel on a second stream touching the next launch's lines\n");
    for (int ahead : {-1, 1, 2, 64}) {
        run<6>("6.3 MB: 256 x 256 x 6 (24 KB/block)", bufs, dep, 256, 256, ahead, progress, sink, sa, sb);
        run<2>("6.3 MB: 192 x 1024 x 2 (32 KB/block)", bufs, dep, 192, 1024, ahead, progress, sink, sa, sb);
        run<8>("25 MB: 192 x 1024 x 8 (128 KB/block)", bufs, dep, 192, 1024, ahead, progress, sink, sa, sb);
        run<8>("8.4 MB: 256 x 256 x 8", bufs, dep, 256, 256, ahead, progress, sink, sa, sb);
    }
    return 0;
}
