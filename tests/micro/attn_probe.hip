// attn_probe.hip -- k_attention_mfma (csrc/kernels_layer.hip) alone at the large-batch shape, in-kernel stamps per phase.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DATTN_STAMPS -I../../nemotron-asr.cpp_amd/csrc -I../../include -o attn_probe attn_probe.hip
//   ./attn_probe [B=512] [T=14]
// K/V rings of 24 "layers" are rotated so that a launch finds none of its rows in L2 / MALL (as in a step: a layer's rings are read once).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kernels_layer.hip"

using namespace nasr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_fill16(bf16_t *p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = f32_to_bf16(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));
    }
}
__global__ void k_fill32(float *p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f);
    }
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 512, T = argc > 2 ? atoi(argv[2]) : 14, NL = 6, REPS = 24;
    const size_t slot_elems = (size_t)2 * KVC * D;
    std::vector<bf16_t *> kv(NL);
    for (int l = 0; l < NL; l++) { CK(hipMalloc((void **)&kv[l], (size_t)B * slot_elems * 2)); hipLaunchKernelGGL(k_fill16, dim3(2048), dim3(256), 0, 0, kv[l], (size_t)B * slot_elems, 7u + l); }
    float *q, *bu, *bv;
    bf16_t *pos, *ctx;
    RowDesc *rows;
    const size_t M = (size_t)B * T;
    CK(hipMalloc((void **)&q, M * D * 4)); CK(hipMalloc((void **)&bu, D * 4)); CK(hipMalloc((void **)&bv, D * 4));
    CK(hipMalloc((void **)&pos, (size_t)(LCTX + 2 * T) * D * 2)); CK(hipMalloc((void **)&ctx, M * D * 2)); CK(hipMalloc((void **)&rows, B * sizeof(RowDesc)));
    hipLaunchKernelGGL(k_fill32, dim3(1024), dim3(256), 0, 0, q, M * D, 1u); hipLaunchKernelGGL(k_fill32, dim3(4), dim3(256), 0, 0, bu, (size_t)D, 2u);
    hipLaunchKernelGGL(k_fill32, dim3(4), dim3(256), 0, 0, bv, (size_t)D, 3u); hipLaunchKernelGGL(k_fill16, dim3(64), dim3(256), 0, 0, pos, (size_t)(LCTX + 2 * T) * D, 4u);
    std::vector<RowDesc> hr(B);
    for (int b = 0; b < B; b++) { memset(&hr[b], 0, sizeof(RowDesc)); hr[b].slot = b; hr[b].valid_len = 70; hr[b].kv_head = (b * 37) % KVC; }
    CK(hipMemcpy(rows, hr.data(), B * sizeof(RowDesc), hipMemcpyHostToDevice));
    AttnParams p;
    memset(&p, 0, sizeof(p));
    p.q = q; p.kv_slot_stride = (int64_t)slot_elems; p.act_bf16 = 1; p.posproj = pos; p.bias_u = bu; p.bias_v = bv; p.rows = rows; p.B = B; p.T = T; p.TS = 0; p.ctx_out = ctx;
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)B * NH * ((2.0 * (LCTX + T) * DH * 2) + T * DH * 4 + T * DH * 2);
    std::vector<bf16_t> out[2];
    for (int variant = 0; variant < 1; variant++) {          // (round 6 also timed a four-streams-per-workgroup form here: profiles/r6_attention.md)
        p.ablate = variant ? 0 : 4;
        for (int l = 0; l < NL; l++) { p.kv_pool = kv[l]; launch_attention(p, 0); }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < REPS; r++) { p.kv_pool = kv[r % NL]; launch_attention(p, 0); }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / REPS;
        printf("%s B=%d T=%d: %.1f us per launch, %.2f TB/s of K/V + q + ctx (%.0f MB)\n", variant ? "k_attention_mfma_ps" : "k_attention_mfma   ", B, T, us, bytes / (us * 1e-6) / 1e12, bytes / 1e6);
        CK(hipMemset(ctx, 0, M * D * 2));
        p.kv_pool = kv[1]; launch_attention(p, 0);
        CK(hipDeviceSynchronize());
        out[variant].resize(M * D);
        CK(hipMemcpy(out[variant].data(), ctx, M * D * 2, hipMemcpyDeviceToHost));
    }
    p.ablate = 4;
    const int QB = T <= 2 ? 16 : T, nz = (T + QB - 1) / QB;
    const size_t nwg = (size_t)NH * B * nz;
    unsigned long long *st;
    CK(hipMalloc((void **)&st, nwg * 64)); CK(hipMemset(st, 0, nwg * 64));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &st, sizeof(st)));
    p.kv_pool = kv[2]; launch_attention(p, 0);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nwg * 8);
    CK(hipMemcpy(h.data(), st, nwg * 64, hipMemcpyDeviceToHost));
    double ph[5] = {0, 0, 0, 0, 0}, life = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (size_t w = 0; w < nwg; w++) {
        const unsigned long long *t = &h[w * 8];
        for (int k = 0; k < 5; k++) ph[k] += (double)(t[k + 1] - t[k]);
        life += (double)(t[5] - t[0]); tmin = std::min(tmin, t[0]); tmax = std::max(tmax, t[5]);
    }
    const double c = 0.01 / nwg;
    printf("  stamped launch %.1f us, %zu workgroups; per workgroup (us): loads + q images %.2f | score tiles %.2f | softmax %.2f | V^T image %.2f | P.V + store %.2f | lifetime %.2f; "
           "workgroups resident on average %.1f per CU\n", (tmax - tmin) * 0.01, nwg, ph[0] * c, ph[1] * c, ph[2] * c, ph[3] * c, ph[4] * c, life * c, life * 0.01 / ((tmax - tmin) * 0.01) / 256);
    return 0;
}
