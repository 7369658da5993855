// Round 3: do GEMM workgroups of two launch chains (two HIP streams on two hardware queues) really share CUs, and what does a pair cost?
// Links the engine's own kernels_gemm.o.  W1 shape (M = 896, N = 4096, K = 1024, SiLU -> bf16) and W2 shape (N = 1024, K = 4096, 2 splits,
// f32 partials), deep-ring kernels (coresident = 0) against the shallow-ring ones (1): one chain of 60 launches alone, then two
// chains of 60 side by side (different outputs, same weights).  Perfect overlap = the time of one chain; none = twice that.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../nemotron-asr.cpp_amd/csrc -o cores_probe cores_probe.hip ../../nemotron-asr.cpp_amd/csrc/kernels_gemm.o
#include "nasr_internal.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
using namespace nasr;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_spin(unsigned long long ticks) {          // 100 MHz s_memrealtime: 1 tick = 10 ns
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_empty() {}
int main(int argc, char **argv) {
    const int offset_us = argc > 1 ? atoi(argv[1]) : 0;
    const int K1 = argc > 2 ? atoi(argv[2]) : 1024;         // K of the W1-shaped GEMM (sweep: fixed part and per-chunk part of a launch)      // chain 1 starts this much later (a one-workgroup spin kernel at the head of its graph)
    const int fence_mode = argc > 4 ? atoi(argv[4]) : 0;    // 1: chain 1 = empty launches (8 per GEMM launch of chain 0): what do its kernel boundaries cost chain 0?
    const int mask_mode = argc > 5 ? atoi(argv[5]) : 0;     // CU masks of the two streams: 1 = low / high half of the 256 mask bits, 2 = even / odd bits, 3 = bits with (i / 8) even / odd
    const int prio_mode = argc > 6 ? atoi(argv[6]) : 0;     // GemmParams::prio of chain 0 (chain 1 keeps 0): 1 = s_setprio 3 throughout, 3 = only in the K loop; +4: chain 1 gets it instead
    const int K2 = argc > 3 ? atoi(argv[3]) : 4096;         // K of the W2-shaped GEMM (N = 1024, 2 splits)
    if (K2 < 256 || K2 > 4096 || K2 % 128) { fprintf(stderr, "K2: a multiple of 128 up to 4096\n"); return 1; }
    if (K1 < 64 || K1 > 1024 || K1 % 64) { fprintf(stderr, "K1: a multiple of 64 up to 1024 (the weight buffer holds 4096 x 1024)\n"); return 1; }
    init_gemm_kernel_attributes();
    const int M = 896, L = 60;
    bf16_t *A, *W, *act[2]; float *part[2];
    CHK(hipMalloc(&A, (size_t)M * 4096 * 2)); CHK(hipMalloc(&W, (size_t)4096 * 1024 * 2));
    std::vector<bf16_t> h((size_t)4096 * 1024);
    for (size_t i = 0; i < h.size(); i++) h[i] = (bf16_t)(0x3c00 + (i * 2654435761u >> 24 & 0x7f) + ((i & 8) ? 0x8000 : 0));
    CHK(hipMemcpy(W, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CHK(hipMemcpy(A, h.data(), (size_t)M * 4096 * 2 < h.size() * 2 ? (size_t)M * 4096 * 2 : h.size() * 2, hipMemcpyHostToDevice));
    for (int c = 0; c < 2; c++) { CHK(hipMalloc(&act[c], (size_t)M * 4096 * 2)); CHK(hipMalloc(&part[c], (size_t)2 * M * 1024 * 4)); }
    hipStream_t st[2];
    if (mask_mode == 0) { for (auto &s : st) CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    else {
        for (int c = 0; c < 2; c++) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < 256; i++) {
                const int side = mask_mode == 1 ? (i >= 128) : mask_mode == 2 ? (i & 1) : ((i >> 3) & 1);
                if (side == c) mask[i >> 5] |= 1u << (i & 31);
            }
            CHK(hipExtStreamCreateWithCUMask(&st[c], 8, mask));
        }
    }
    for (int shape = 0; shape < 3; shape++)          // 2: chain 0 runs W1 launches, chain 1 W2 launches (different lengths: the phases drift)
        for (int cores = 0; cores < 2; cores++) {
            hipGraphExec_t ex[2];
            for (int c = 0; c < 2; c++) {
                GemmParams g;
                memset(&g, 0, sizeof(g));
                g.A = A; g.W = W; g.M = M; g.coresident = cores; g.prio = (c == ((prio_mode >> 2) & 1)) ? (prio_mode & 3) : 0;
                if (shape == 0 || (shape == 2 && c == 0)) { g.N = 4096; g.K = K1; g.lda = K1; g.splits = 1; g.epi = EPI_SILU_ACT; g.out_act = act[c]; g.ldo_act = 4096; }
                else { g.N = 1024; g.K = K2; g.lda = K2; g.splits = 2; g.epi = EPI_PART_F32; g.out_f32 = part[c]; g.ldo = 1024; }
                hipGraph_t gr;
                CHK(hipStreamBeginCapture(st[c], hipStreamCaptureModeThreadLocal));
                if (c == 1 && offset_us > 0) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[c], (unsigned long long)offset_us * 100);
                if (fence_mode && c == 1) { for (int i = 0; i < 8 * L; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[c]); }
                else for (int i = 0; i < L; i++) launch_gemm_bf16(g, st[c]);
                CHK(hipStreamEndCapture(st[c], &gr)); CHK(hipGraphInstantiate(&ex[c], gr, nullptr, nullptr, 0)); CHK(hipGraphDestroy(gr));
            }
            double t[2] = {1e18, 1e18};
            for (int n = 1; n <= 2; n++)
                for (int rep = 0; rep < 8; rep++) {
                    for (auto &s : st) CHK(hipStreamSynchronize(s));
                    auto t0 = std::chrono::steady_clock::now();
                    for (int c = 0; c < n; c++) CHK(hipGraphLaunch(ex[c], st[c]));
                    for (int c = 0; c < n; c++) CHK(hipStreamSynchronize(st[c]));
                    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                    if (us < t[n - 1]) t[n - 1] = us;
                }
            printf("[offset %d us, K1 %d, K2 %d, masks %d, prio %d] %s, %s rings: one chain %.2f us per launch; two chains side by side %.2f us per launch pair (%.2fx one chain)\n",
                   offset_us, K1, K2, mask_mode, prio_mode, shape == 0 ? "W1 (4096 x 1024, SiLU bf16 out)" : shape == 1 ? "W2 (1024 x 4096, 2 splits, f32 partials)" : "chain 0 W1 / chain 1 W2 (one chain = W1 alone)", cores ? "shallow (two workgroups per CU)" : "deep (one workgroup per CU)",
                   t[0] / L, t[1] / L, t[1] / t[0]);
            for (auto &e : ex) CHK(hipGraphExecDestroy(e));
        }
    return 0;
}
