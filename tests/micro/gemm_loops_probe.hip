// Round 5: the large-M GEMM loops alone on the chip with COLD operands, per launch: rounds 1-4's (k_gemm_wide, k_gemm_tiled2_k32: GemmParams::prio = 20) against round 5's
// (k_gemm_wide2, k_gemm_tiled3).  Earlier in the round this probe also timed s_setprio around the MFMA cluster and two re-orderings of the old loops: profiles/r5_gemm_loops_probe_*.txt.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../nemotron-asr.cpp_amd/csrc -c gemm_loops_probe.hip -o /tmp/prp.o && hipcc --offload-arch=gfx950 -o gemm_loops_probe /tmp/prp.o ../../nemotron-asr.cpp_amd/csrc/kernels_gemm.o
#include "nasr_internal.h"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace nasr;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char **argv) {
    init_gemm_kernel_attributes();
    std::vector<int> Ms;
    for (int i = 1; i < argc; i++) Ms.push_back(atoi(argv[i]));
    if (Ms.empty()) Ms = {896, 3584, 7168};
    const int Mmax = 7168;
    bf16_t *A, *W, *act; float *outf, *bias;
    const int NW = 12, NA = 3;
    CHK(hipMalloc(&A, (size_t)NA * Mmax * 4096 * 2)); CHK(hipMalloc(&W, (size_t)NW * 4096 * 4096 * 2));
    CHK(hipMalloc(&act, (size_t)Mmax * 4096 * 2)); CHK(hipMalloc(&outf, (size_t)Mmax * 4096 * 4)); CHK(hipMalloc(&bias, 4096 * 4));
    std::vector<bf16_t> h((size_t)4096 * 4096);
    for (size_t i = 0; i < h.size(); i++) h[i] = (bf16_t)(0x3c00 + (i * 2654435761u >> 24 & 0x7f) + ((i & 8) ? 0x8000 : 0));
    for (int i = 0; i < NW; i++) CHK(hipMemcpy(W + (size_t)i * h.size(), h.data(), h.size() * 2, hipMemcpyHostToDevice));
    for (size_t off = 0; off < (size_t)NA * Mmax * 4096; off += h.size()) CHK(hipMemcpy(A + off, h.data(), std::min(h.size(), (size_t)NA * Mmax * 4096 - off) * 2, hipMemcpyHostToDevice));
    CHK(hipMemset(bias, 0, 4096 * 4));
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct Shape { const char *name; int N, K, epi; } shapes[] = {{"W1  N=4096 K=1024 SiLU->bf16", 4096, 1024, EPI_SILU_ACT}, {"W2  N=1024 K=4096 f32", 1024, 4096, EPI_PART_F32},
        {"Wo  N=1024 K=1024 f32", 1024, 1024, EPI_PART_F32}, {"pw1 N=2048 K=1024 GLU", 2048, 1024, EPI_GLU}, {"QKV-shaped N=3072 K=1024 f32", 3072, 1024, EPI_PART_F32}};
    for (int M : Ms)
        for (const Shape &s : shapes) {
            double us[2][2];
            for (int rep = 0; rep < 2; rep++)
                for (int pv = 0; pv < 2; pv++) {
                    GemmParams g;
                    memset(&g, 0, sizeof(g));
                    g.A = A; g.W = W; g.M = M; g.N = s.N; g.K = s.K; g.lda = s.K; g.splits = 1; g.epi = s.epi;
                    g.out_f32 = outf; g.ldo = s.epi == EPI_GLU ? s.N / 2 : s.N; g.out_act = act; g.ldo_act = s.N; g.bias = bias;
                    g.no_persist = 1; g.coresident = 1; g.prio = (pv == 0 ? 5 : 0) << 2;          // the engine's pipelined configuration
                    for (int i = 0; i < 3; i++) launch_gemm_bf16(g, st);
                    hipEvent_t a, b;
                    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
                    const int reps = 40;
                    CHK(hipEventRecord(a, st));
                    for (int i = 0; i < reps; i++) {
                        g.W = W + (size_t)(i % (4 * NW)) * ((size_t)1024 * 4096); g.A = A + (size_t)(i % NA) * Mmax * 4096;
                        launch_gemm_bf16(g, st);
                    }
                    CHK(hipEventRecord(b, st));
                    CHK(hipEventSynchronize(b));
                    float ms = 0;
                    CHK(hipEventElapsedTime(&ms, a, b));
                    us[pv][rep] = 1e3 * ms / reps;
                }
            const double fl = 2.0 * M * s.N * s.K * 1e-6;
            printf("[cold] M = %5d  %-30s", M, s.name);
            const char *nm[2] = {"rounds 1-4 loops", "round 5 loops"};
            for (int pv = 0; pv < 2; pv++) printf(" | %s %7.2f / %7.2f us (%5.0f TF)", nm[pv], us[pv][0], us[pv][1], fl / std::min(us[pv][0], us[pv][1]));
            printf("\n");
        }
    return 0;
}
