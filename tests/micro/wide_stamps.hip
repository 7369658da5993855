// Round 5: where a 224 x 256 tile of k_gemm_wide2 spends its cycles (in-kernel stamps, a diagnostic compile of kernels_gemm.hip with -DNASR_GEMM_STAMPS: no
// stamp executes in the product build).  Per workgroup: s_memtime at entry / chunk 0 landed / K loop done / stores drained, s_memrealtime at entry and end.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNASR_GEMM_STAMPS -I../../include -I../../nemotron-asr.cpp_amd/csrc -c ../../nemotron-asr.cpp_amd/csrc/kernels_gemm.hip -o /tmp/kgs.o
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNASR_GEMM_STAMPS -I../../include -I../../nemotron-asr.cpp_amd/csrc -c wide_stamps.hip -o /tmp/ws.o && hipcc --offload-arch=gfx950 -o wide_stamps /tmp/ws.o /tmp/kgs.o
#include "nasr_internal.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
using namespace nasr;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
static double pct(std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; }
int main(int argc, char **argv) {
    init_gemm_kernel_attributes();
    const int M = argc > 1 ? atoi(argv[1]) : 7168;
    const int Mmax = 7168;
    bf16_t *A, *W, *act; float *outf, *bias; unsigned long long *stamps;
    const int NW = 12, NA = 3;
    CHK(hipMalloc(&A, (size_t)NA * Mmax * 4096 * 2)); CHK(hipMalloc(&W, (size_t)NW * 4096 * 4096 * 2));
    CHK(hipMalloc(&act, (size_t)Mmax * 4096 * 2)); CHK(hipMalloc(&outf, (size_t)Mmax * 4096 * 4)); CHK(hipMalloc(&bias, 4096 * 4));
    CHK(hipMalloc(&stamps, 1024 * 8 * 8));
    std::vector<bf16_t> h((size_t)4096 * 4096);
    for (size_t i = 0; i < h.size(); i++) h[i] = (bf16_t)(0x3c00 + (i * 2654435761u >> 24 & 0x7f) + ((i & 8) ? 0x8000 : 0));
    for (int i = 0; i < NW; i++) CHK(hipMemcpy(W + (size_t)i * h.size(), h.data(), h.size() * 2, hipMemcpyHostToDevice));
    for (size_t off = 0; off < (size_t)NA * Mmax * 4096; off += h.size()) CHK(hipMemcpy(A + off, h.data(), std::min(h.size(), (size_t)NA * Mmax * 4096 - off) * 2, hipMemcpyHostToDevice));
    CHK(hipMemset(bias, 0, 4096 * 4));
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct Shape { const char *name; int N, K, epi; } shapes[] = {{"W1  N=4096 K=1024 SiLU->bf16", 4096, 1024, EPI_SILU_ACT}, {"W2  N=1024 K=4096 f32", 1024, 4096, EPI_PART_F32},
        {"pw1 N=2048 K=1024 GLU", 2048, 1024, EPI_GLU}, {"QKV-shaped N=3072 K=1024 f32", 3072, 1024, EPI_PART_F32}};
    for (const Shape &s : shapes) {
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = A; g.W = W; g.M = M; g.N = s.N; g.K = s.K; g.lda = s.K; g.splits = 1; g.epi = s.epi;
        g.out_f32 = outf; g.ldo = s.epi == EPI_GLU ? s.N / 2 : s.N; g.out_act = act; g.ldo_act = s.N; g.bias = bias;
        g.no_persist = 1; g.coresident = 1; g.prio = 0; g.stamps = nullptr;
        for (int i = 0; i < 200; i++) {          // ~15 ms of back-to-back launches first: the clock the chip holds under this load
            g.W = W + (size_t)(i % (4 * NW)) * ((size_t)1024 * 4096); g.A = A + (size_t)(i % NA) * Mmax * 4096;
            if (i == 199) { CHK(hipMemsetAsync(stamps, 0, 1024 * 64, st)); g.stamps = stamps; }
            launch_gemm_bf16(g, st);
        }
        CHK(hipStreamSynchronize(st));
        std::vector<unsigned long long> hs(1024 * 8);
        CHK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        const bool small = M < 1344;          // below k_gemm_wide2's range: the 128 x 128 tiles of k_gemm_tiled3
        const int tiles = small ? (s.N / 128) * ((M + 127) / 128) : (s.N == 3072 ? (s.N / 192) : (s.N / 256)) * ((M + 223) / 224), KT = s.K / 32;
        std::vector<double> pro, loop, epi, clk, start, end;
        unsigned long long r0 = ~0ull;
        for (int b = 0; b < tiles; b++) if (hs[b * 8 + 4]) r0 = std::min(r0, hs[b * 8 + 4]);
        for (int b = 0; b < tiles; b++) {
            const unsigned long long *t = &hs[b * 8];
            if (!t[4] || !t[7]) continue;
            pro.push_back((double)(t[1] - t[0])); loop.push_back((double)(t[2] - t[1]) / KT); epi.push_back((double)(t[3] - t[2]));
            clk.push_back((double)(t[3] - t[0]) / (double)(t[7] - t[4]) * 0.1);          // GHz: shader cycles per 10 ns tick
            start.push_back((t[4] - r0) * 0.01); end.push_back((t[7] - r0) * 0.01);      // us
        }
        if (pro.empty()) { printf("M = %d %s: no stamps (another kernel took this shape)\n", M, s.name); continue; }
        int late = 0;
        for (double v : start) late += v > 2.0;
        printf("M = %d %-30s %d tiles: in-kernel clock %.2f GHz | prologue %6.0f cyc (p90 %6.0f) | K loop %6.0f cyc per chunk (p10 %6.0f p90 %6.0f; MFMA alone: 896 / 256 on 224 x 256 / 128 x 128 tiles) | epilogue %6.0f cyc (p90 %6.0f) | "
               "workgroups starting > 2 us after the first: %d; last end %.1f us, median end %.1f us\n",
               M, s.name, tiles, med(clk), med(pro), pct(pro, 0.9), med(loop), pct(loop, 0.1), pct(loop, 0.9), med(epi), pct(epi, 0.9), late, *std::max_element(end.begin(), end.end()), med(end));
    }
    return 0;
}
