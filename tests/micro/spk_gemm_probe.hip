// spk_gemm_probe.hip -- k_spk_gemm (csrc/kernels_spk.hip) alone at TitaNet-L's shapes, with cold operands and in-kernel stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSG_STAMPS -I../../nemotron-asr.cpp_amd/csrc -I../../include -o spk_gemm_probe spk_gemm_probe.hip
//   ./spk_gemm_probe [S=96]
// Per mode and shape: average launch duration (HIP events, operands rotated through 6 buffer sets so that a launch finds none of them in L2 / MALL),
// TFLOP/s, and from the stamps of every workgroup (s_memrealtime, 100 MHz): prologue (entry -> first chunk landed), K loop, epilogue, and how the
// workgroups' lifetimes fill the launch (first start -> last end).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kernels_spk.hip"

using namespace nasr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_fill_bf16(bf16_t *p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = f32_to_bf16(((int)(h & 0xffff) - 32768) * (1.0f / 65536.0f));
    }
}
__global__ void k_fill_f32(float *p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((int)(h & 0xffff) - 32768) * (1.0f / 65536.0f);
    }
}

int main(int argc, char **argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 96;
    const int M = S * SPK_T, NSETS = argc > 2 ? atoi(argv[2]) : 6, REPS = 18;
    init_spk_kernel_attributes();
    struct Shape { int mode, N, K, dwk; const char *name; };
    const Shape shapes[] = {{SG_Y, 1024, 1024, 0, "SG_Y   1024x1024"}, {SG_DW, 1024, 1024, 11, "SG_DW  1024x1024 k=11"}, {SG_RES, 1024, 1024, 15, "SG_RES 1024x1024 k=15"},
                            {SG_Y, 3072, 1024, 0, "SG_Y   3072x1024"}, {SG_ASP, 3072, 128, 0, "SG_ASP 3072x128"}, {SG_Y, 1024, 128, 0, "SG_Y   1024x128"}};
    const size_t maxN = 3072;
    std::vector<bf16_t *> A(NSETS), W(NSETS), AO(NSETS), XO(NSETS), XI(NSETS);
    std::vector<float *> Y(NSETS);
    for (int i = 0; i < NSETS; i++) {
        CK(hipMalloc((void **)&A[i], (size_t)M * 1024 * 2)); CK(hipMalloc((void **)&W[i], maxN * 1024 * 2)); CK(hipMalloc((void **)&AO[i], (size_t)M * 1024 * 2));
        CK(hipMalloc((void **)&XO[i], (size_t)M * 1024 * 2)); CK(hipMalloc((void **)&XI[i], (size_t)M * maxN * 2)); CK(hipMalloc((void **)&Y[i], (size_t)M * maxN * 4));
        hipLaunchKernelGGL(k_fill_bf16, dim3(1024), dim3(256), 0, 0, A[i], (size_t)M * 1024, 11u + i);
        hipLaunchKernelGGL(k_fill_bf16, dim3(1024), dim3(256), 0, 0, W[i], maxN * 1024, 101u + i);
        hipLaunchKernelGGL(k_fill_bf16, dim3(1024), dim3(256), 0, 0, XI[i], (size_t)M * maxN, 201u + i);
        hipLaunchKernelGGL(k_fill_f32, dim3(1024), dim3(256), 0, 0, Y[i], (size_t)M * maxN, 301u + i);
    }
    float *bias, *dw, *z, *colmean, *pool, *bn;
    int *lens;
    unsigned long long *stamps;
    CK(hipMalloc((void **)&bias, maxN * 4)); CK(hipMalloc((void **)&dw, 15 * 1024 * 4)); CK(hipMalloc((void **)&z, (size_t)S * maxN * 4)); CK(hipMalloc((void **)&colmean, (size_t)S * maxN * 4));
    CK(hipMalloc((void **)&pool, (size_t)S * 2 * maxN * 4)); CK(hipMalloc((void **)&bn, 2 * maxN * 4)); CK(hipMalloc((void **)&lens, S * 4));
    const size_t max_wg = (size_t)S * (maxN / 128);
    CK(hipMalloc((void **)&stamps, max_wg * 8 * 8));
    hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, 0, bias, maxN, 1u); hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, 0, dw, (size_t)15 * 1024, 2u);
    hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, 0, z, (size_t)S * maxN, 3u); hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, 0, bn, 2 * maxN, 4u);
    std::vector<int> hl(S);
    for (int s = 0; s < S; s++) hl[s] = s % 7 == 3 ? 75 : SPK_TVALID;
    CK(hipMemcpy(lens, hl.data(), S * 4, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape &sh : shapes) {
        SpkGemmParams g;
        memset(&g, 0, sizeof(g));
        g.S = S; g.N = sh.N; g.K = sh.K; g.lda = sh.K; g.mode = sh.mode; g.bias = bias; g.lens = lens; g.stamps = nullptr;
        g.dw_w = dw; g.dw_k = sh.dwk; g.lda_out = 1024; g.colmean = colmean; g.z = z; g.bn_s = bn; g.bn_b = bn; g.pool = pool;
        auto set = [&](int i) { g.A = A[i]; g.W = W[i]; g.a_out = AO[i]; g.y_out = Y[i]; g.y_in = Y[i]; g.x_out = XO[i]; g.x_in = XI[i]; };
        set(0);
        if (const char *why = spk_gemm_check(g)) { printf("%s: %s\n", sh.name, why); continue; }
        for (int i = 0; i < NSETS; i++) { set(i); launch_spk_gemm(g, 0); }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < REPS; r++) { set(r % NSETS); launch_spk_gemm(g, 0); }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / REPS, tf = 2.0 * M * sh.N * sh.K / (us * 1e-6) / 1e12;
        // one stamped launch
        const size_t nwg = (size_t)S * (sh.N / 128);
        CK(hipMemset(stamps, 0, nwg * 64));
        set(NSETS > 3 ? 3 : 0); g.stamps = stamps;
        launch_spk_gemm(g, 0);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(nwg * 8);
        CK(hipMemcpy(h.data(), stamps, nwg * 64, hipMemcpyDeviceToHost));
        unsigned long long t_min = ~0ull, t_max = 0;
        double pro = 0, loop = 0, epi = 0, life = 0, clk_num = 0, clk_den = 0;
        for (size_t w = 0; w < nwg; w++) {
            const unsigned long long *t = &h[w * 8];
            clk_num += (double)(t[4 + 2] - t[4 + 1]); clk_den += (double)(t[2] - t[1]);          // core-clock ticks / 100 MHz ticks over the K loop
            t_min = std::min(t_min, t[0]); t_max = std::max(t_max, t[3]);
            pro += (double)(t[1] - t[0]); loop += (double)(t[2] - t[1]); epi += (double)(t[3] - t[2]); life += (double)(t[3] - t[0]);
        }
        const double c = 0.01 / nwg;     // 100 MHz ticks -> us, mean over workgroups
        printf("%-24s %7.1f us  %6.0f TFLOP/s | stamped launch %6.1f us; per workgroup: prologue %5.2f  K loop %6.2f (%5.3f us per chunk)  epilogue %6.2f  lifetime %6.2f us; "
               "%zu workgroups, lifetime sum / (launch x 512 slots) = %.2f; core clock during the K loops %.2f GHz\n",
               sh.name, us, tf, (t_max - t_min) * 0.01, pro * c, loop * c, loop * c / (sh.K / 32 - 1), epi * c, life * c, nwg, life * 0.01 / ((t_max - t_min) * 0.01 * 512), clk_den > 0 ? clk_num / clk_den * 0.1 : 0.0);
    }
    return 0;
}
