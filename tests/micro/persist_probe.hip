// Round 4: the engine's large-M GEMM paths alone on the chip, per launch: the persistent tile loop (k_gemm_persist) against the
// per-tile kernels (two co-resident workgroups per CU from 1 792 rows: k_gemm_tiled2_k32<4>).  Links the engine's kernels_gemm.o.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../nemotron-asr.cpp_amd/csrc -c persist_probe.hip -o /tmp/pp.o && hipcc --offload-arch=gfx950 -o persist_probe /tmp/pp.o ../../nemotron-asr.cpp_amd/csrc/kernels_gemm.o
//   ./persist_probe [M ...]
#include "nasr_internal.h"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace nasr;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char **argv) {
    init_gemm_kernel_attributes();
    std::vector<int> Ms;
    // "cold": every launch reads another weight matrix (48 x 8 MiB, cycled: more than the 256 MiB of MALL) and another activation
    // buffer (4 x), as the engine's GEMMs do -- a layer's weights come from HBM once per step and the activations were just written
    bool cold = false;
    for (int i = 1; i < argc; i++) { if (!strcmp(argv[i], "cold")) cold = true; else Ms.push_back(atoi(argv[i])); }
    if (Ms.empty()) Ms = {1792, 3584, 7168, 15360};
    const int Mmax = 15360;
    bf16_t *A, *W, *act; float *outf, *bias;
    const int NW = cold ? 12 : 1, NA = cold ? 3 : 1;          // 12 x 32 MiB of weights (4 matrices of 8 MiB each), 3 activation buffers
    CHK(hipMalloc(&A, (size_t)NA * Mmax * 4096 * 2)); CHK(hipMalloc(&W, (size_t)NW * 4096 * 4096 * 2));
    CHK(hipMalloc(&act, (size_t)Mmax * 4096 * 2)); CHK(hipMalloc(&outf, (size_t)Mmax * 4096 * 4)); CHK(hipMalloc(&bias, 4096 * 4));
    std::vector<bf16_t> h((size_t)4096 * 4096);
    for (size_t i = 0; i < h.size(); i++) h[i] = (bf16_t)(0x3c00 + (i * 2654435761u >> 24 & 0x7f) + ((i & 8) ? 0x8000 : 0));
    for (int i = 0; i < NW; i++) CHK(hipMemcpy(W + (size_t)i * h.size(), h.data(), h.size() * 2, hipMemcpyHostToDevice));
    for (size_t off = 0; off < (size_t)NA * Mmax * 4096; off += h.size()) CHK(hipMemcpy(A + off, h.data(), std::min(h.size(), (size_t)NA * Mmax * 4096 - off) * 2, hipMemcpyHostToDevice));
    CHK(hipMemset(bias, 0, 4096 * 4));
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct Shape { const char *name; int N, K, epi; } shapes[] = {{"W1  N=4096 K=1024 SiLU->bf16", 4096, 1024, EPI_SILU_ACT}, {"W2  N=1024 K=4096 f32", 1024, 4096, EPI_PART_F32},
        {"Wo  N=1024 K=1024 f32", 1024, 1024, EPI_PART_F32}, {"pw1 N=2048 K=1024 GLU", 2048, 1024, EPI_GLU}, {"spk N=1024 K=1024 bias+relu f32", 1024, 1024, EPI_BIAS_RELU_F32},
        {"spk N=3072 K=1024 bias+relu f32", 3072, 1024, EPI_BIAS_RELU_F32}, {"QKV-shaped N=3072 K=1024 f32", 3072, 1024, EPI_PART_F32}};
    for (int M : Ms)
        for (const Shape &s : shapes) {
            double us[5];
            for (int mode = 0; mode < 5; mode++) {          // 0: persistent allowed, 1: per-tile kernels (co-resident rule), 2: per-tile, deep rings forced, 3: wide tiles (k_gemm_wide: 256 or 224 rows, the launcher's rule), 4: wide tiles, 256 rows only
                GemmParams g;
                memset(&g, 0, sizeof(g));
                g.A = A; g.W = W; g.M = M; g.N = s.N; g.K = s.K; g.lda = s.K; g.splits = 1; g.epi = s.epi;
                g.out_f32 = outf; g.ldo = s.epi == EPI_GLU ? s.N / 2 : s.N; g.out_act = act; g.ldo_act = s.N; g.bias = bias;
                g.no_persist = mode >= 1; g.coresident = mode == 2 ? 3 : 0; g.no_wide = mode != 3; g.wide_rows = 0; g.tile_bands = mode == 4 ? 2 : 0;
                for (int i = 0; i < 3; i++) launch_gemm_bf16(g, st);
                hipEvent_t a, b;
                CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
                const int reps = 40;
                CHK(hipEventRecord(a, st));
                for (int i = 0; i < reps; i++) {
                    if (cold) { g.W = W + (size_t)(i % (4 * NW)) * ((size_t)1024 * 4096); g.A = A + (size_t)(i % NA) * Mmax * 4096; }
                    launch_gemm_bf16(g, st);
                }
                CHK(hipEventRecord(b, st));
                CHK(hipEventSynchronize(b));
                float ms = 0;
                CHK(hipEventElapsedTime(&ms, a, b));
                us[mode] = 1e3 * ms / reps;
            }
            const double fl = 2.0 * M * s.N * s.K * 1e-6;
            printf("%sM = %5d  %-34s persistent %7.2f us (%5.0f TFLOP/s) | per-tile, two per CU %7.2f us (%5.0f) | per-tile, deep rings %7.2f us (%5.0f) | wide tiles %7.2f us (%5.0f) | per-tile, two per CU, row-fastest order %7.2f us (%5.0f)\n", cold ? "[cold] " : "", M, s.name,
                   us[0], fl / us[0], us[1], fl / us[1], us[2], fl / us[2], us[3], fl / us[3], us[4], fl / us[4]);
        }
    return 0;
}
